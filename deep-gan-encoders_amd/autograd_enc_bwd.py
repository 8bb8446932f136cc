"""Hand-written backward of the encoder (reference model/E/E.py:50-85,122-136 differentiated).

Gradient enters only through the latent codes w (E_align_s2.py:203-221: neither phase uses the
encoder's const output in a loss that is back-propagated).  Weights are re-packed from the
*current* parameter values at backward time, so the second backward of a step sees the weights
already updated by LREQAdam together with the activations saved before the update - the
reference's behaviour.
"""
import torch

from . import ops
from .autograd_enc import _packed, heads_layout


def _linear_backward(lin, g_w, musig, grads, name):
    """w = musig @ W^T + b  ->  g_musig [B,2C]; parameter gradients into `grads`."""
    B = g_w.shape[0]
    W = lin.weight.detach()
    gms = torch.empty((B, W.shape[1]), dtype=torch.float32, device=g_w.device)
    ops.linear_t(g_w, W, gms, ldx=g_w.stride(0), B=B)
    gw = torch.empty_like(W)
    gb = torch.empty_like(lin.bias)
    ops.dense_wgrad(g_w, musig, gw, gb)
    grads[name + ".weight"], grads[name + ".bias"] = gw, gb
    return gms


import os
FUSE_IN_BWD = not os.environ.get("DGE_NO_FUSED_IN_BWD")


def encoder_backward(E, saved, g_w):
    """Returns gradients for E.parameters() in registration order (None where the reference
    produces none, e.g. the last block's noise_weight_2 / bias_2)."""
    if saved is None:
        raise RuntimeError("encoder forward ran without saved activations")
    cache = E.__dict__.setdefault("_pack_cache", {})
    dev = g_w.device
    L = E.layer_count
    B = g_w.shape[0]
    grads = {}
    g_out = fr = None
    later = ops.DeferredSums()          # per-channel parameter-gradient reductions: one grouped launch (two with the DDP hook)
    post = []                           # what reads a deferred sum runs after the flush

    def flush_sums():
        later.flush()
        for f in post:
            f()
        post.clear()
    R = saved["img"].shape[2]
    dt = ops.dtype_of(saved["x0"])
    # every inver_mod head at once (their gradient g_w is complete before the backward starts): two launches instead of 4 per block
    heads = None
    if saved.get("musig_all") is not None:
        lay = heads_layout(E, B, dev)
        if g_w.stride(2) != 1 or g_w.stride(1) != lay["O"]:
            g_w = g_w.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        gms_all, gw_all = torch.empty(lay["total_m"], **f32), torch.empty(lay["total_w"], **f32)
        gb_all = torch.empty(lay["n"] * lay["O"], **f32)
        ops.check(ops.lib().dge_heads_bwd(ops._p(lay["tab"]), lay["n"], lay["max_I"], ops._f32(g_w), g_w.stride(0),
                                          ops._f32(saved["musig_all"]), ops._p(gms_all), ops._p(gw_all), ops._p(gb_all), B, lay["O"],
                                          ops._stream()), "dge_heads_bwd")

        def heads(i, name):
            moff, woff, boff, I = lay["items"][i]
            grads[name + ".weight"] = gw_all[woff:woff + lay["O"] * I].view(lay["O"], I)
            grads[name + ".bias"] = gb_all[boff:boff + lay["O"]]
            return gms_all[moff:moff + B * I].view(B, I)
    for j in range(L - 1, -1, -1):
        blk = E.decode_block[j]
        rec = saved["blocks"][j]
        pre = f"decode_block.{j}."
        Cc, C2 = blk.inputs, blk.outputs
        H = R >> j
        N = H * H
        last = not blk.has_last_conv
        has3 = Cc != C2
        # w index map (E.py:130-134): w[:, 2(L-1-j)] = w2_j, w[:, 2(L-1-j)+1] = w1_j
        g_w2, g_w1 = g_w[:, 2 * (L - 1 - j)], g_w[:, 2 * (L - 1 - j) + 1]
        if heads is not None:
            gms2, gms1 = heads(2 * j + 1, pre + "inver_mod2"), heads(2 * j, pre + "inver_mod1")
        else:
            gms2 = _linear_backward(blk.inver_mod2, g_w2, rec["musig2"], grads, pre + "inver_mod2")
            gms1 = _linear_backward(blk.inver_mod1, g_w1, rec["musig1"], grads, pre + "inver_mod1")
        x, x1 = rec["x"], rec["x1"]
        extra, extra_pool, extra_scale = None, False, 1.0
        fuse2 = False
        if not last:
            if g_out is None:
                raise RuntimeError("non-final encoder block without an output gradient")
            # planar reductions ([k, C]): every parameter gradient below is a contiguous view, no strided copies
            red2 = ops.zeros((3 if has3 else 2, C2), dev)     # third row: sum of g_out = conv_3.bias gradient / 0.889
            g_pre2 = ops.act_bwd_mask(g_out, rec["m2"], rec["n2"], scale=0.111 * 0.25, red=red2, planar=True, defer=later)
            grads[pre + "bias_2"] = red2[0].reshape(1, C2, 1, 1)
            grads[pre + "noise_weight_2"] = red2[1].reshape(1, C2, 1, 1)
            gW2 = ops.zeros(tuple(blk.conv_2.weight.shape), dev)
            dots2 = ops.SlotStats(B, Cc, dev)                 # slot copies are added by in_bwd_coef
            # High-resolution blocks: the two sums the instance-norm backward needs come out of the weight-gradient launch, so the data
            # gradient can apply that backward (and the activation backward of conv_1's tail) in its epilogue - the in_bwd pass over
            # g_y2 and x1 below disappears (measured at batch 8: 410 -> 239 us at 1024^2, 212 -> 121 us at 512^2)
            fuse2 = FUSE_IN_BWD and ops.conv_in_bwd_supported(B, H, H, C2, Cc, dt) and \
                ops.conv_wgrad_dots(g_pre2, x1, gW2, rec["sc2"], rec["sh2"], blk.conv_2.weight, dots2)
            if not fuse2:
                ops.conv_wgrad(g_pre2, x1, gW2, rec["sc2"], rec["sh2"])
            grads[pre + "conv_2.weight"] = gW2
            if not fuse2:
                g_y2 = ops.conv2d(g_pre2, _packed(cache, blk.conv_2, dt, ops.PACK_DGRAD, H), Cc, 3, stats=dots2, dot_src=x1)
            if has3:
                post.append(lambda n=pre + "conv_3.bias", t=red2[2]: grads.__setitem__(n, t * 0.889))
                gW3 = ops.zeros(tuple(blk.conv_3.weight.shape), dev)
                ops.conv_wgrad(g_out, rec["xp"], gW3)
                grads[pre + "conv_3.weight"] = ops.scale_(gW3, 0.889)
                extra = ops.conv2d(g_out, _packed(cache, blk.conv_3, dt, ops.PACK_DGRAD), Cc, 1, gain=0.889)
                extra_pool, extra_scale = True, 0.25
            else:
                extra, extra_pool, extra_scale = g_out, True, 0.889 * 0.25
        else:
            if g_out is not None:
                raise RuntimeError("the final encoder block's activation output carries no gradient in E_align")
            g_y2, dots2 = None, None
        coef2 = (dots2, gms2, rec["musig2"], rec["sc2"], rec["sh2"], N)          # computed inside in_bwd (dge_in_bwd_fused)
        red1 = ops.zeros((2, Cc), dev)
        if fuse2:
            redp = ops.SlotStats(B, Cc, dev)
            g_pre1 = ops.conv2d(g_pre2, _packed(cache, blk.conv_2, dt, ops.PACK_DGRAD, H), Cc, 3, dot_src=x1,
                                in_bwd=dict(coef=ops.in_bwd_coef(*coef2), noise=rec["n1"].reshape(B, H, H), red=redp))
            ops._sum_planar(redp.buf.view(-1, Cc, 2), red1, later)
        else:
            g_pre1 = ops.in_bwd(g_y2, x1, coef2, noise=rec["n1"], act=True, red=red1, planar=True, defer=later)
        grads[pre + "bias_1"] = red1[0].reshape(1, Cc, 1, 1)
        grads[pre + "noise_weight_1"] = red1[1].reshape(1, Cc, 1, 1)
        gW1 = ops.zeros(tuple(blk.conv_1.weight.shape), dev)
        dots1 = ops.SlotStats(B, Cc, dev)
        # Block 0: the gradient w.r.t. the FromRGB output has one reader, the FromRGB parameter gradients.  With the instance-norm sums
        # out of the weight-gradient launch, the data gradient of conv_1 reduces them in its epilogue and stores nothing (the
        # in_bwd_fromrgb pass over g_y1, x0 and the image disappears, and so does the store of g_y1)
        fuse_fr = FUSE_IN_BWD and j == 0 and saved.get("img4") is not None and ops.conv_in_bwd_fromrgb_supported(B, H, H, Cc, Cc, dt) and \
            ops.conv_wgrad_dots(g_pre1, x, gW1, rec["sc1"], rec["sh1"], blk.conv_1.weight, dots1)
        # Blocks 1, 2: the same for the block input (instance-norm backward + pooled skip gradient, no activation) - the data gradient
        # stores the block's input gradient itself
        fuse_x = FUSE_IN_BWD and not fuse_fr and j > 0 and (extra is None or extra_pool) and ops.conv_in_bwd_x_supported(B, H, H, Cc, Cc, dt) and \
            ops.conv_wgrad_dots(g_pre1, x, gW1, rec["sc1"], rec["sh1"], blk.conv_1.weight, dots1)
        if not (fuse_fr or fuse_x):
            ops.conv_wgrad(g_pre1, x, gW1, rec["sc1"], rec["sh1"])
        grads[pre + "conv_1.weight"] = gW1
        coef1 = (dots1, gms1, rec["musig1"], rec["sc1"], rec["sh1"], N)
        if fuse_fr:
            frh = ops.SlotStats(B, Cc, dev)
            ops.conv2d(g_pre1, _packed(cache, blk.conv_1, dt, ops.PACK_DGRAD, H), Cc, 3, dot_src=x, out=x.new_empty((1, 1, 1, 1)),
                       in_bwd=dict(coef=ops.in_bwd_coef(*coef1), fr=frh, img4=saved["img4"], extra=extra if extra_pool else None,
                                   extra_scale=extra_scale))
            fr = ops._sum_planar(frh.buf.view(-1, Cc, 4), torch.empty((4, Cc), dtype=torch.float32, device=dev), later)
        elif fuse_x:
            g_out = ops.conv2d(g_pre1, _packed(cache, blk.conv_1, dt, ops.PACK_DGRAD, H), Cc, 3, dot_src=x,
                               in_bwd=dict(coef=ops.in_bwd_coef(*coef1), extra=extra, extra_scale=extra_scale))
        else:
            g_y1 = ops.conv2d(g_pre1, _packed(cache, blk.conv_1, dt, ops.PACK_DGRAD, H), Cc, 3, stats=dots1, dot_src=x)
        if fuse_fr or fuse_x:
            pass
        elif j == 0 and Cc <= 512:
            # x is the FromRGB output: its gradient has one reader, the FromRGB parameter gradients - reduced in the same launch
            fr = ops.in_bwd_fromrgb(g_y1, x, coef1, saved["img"].float(), extra=extra, extra_pool=extra_pool, extra_scale=extra_scale,
                                    defer=later)
        else:
            g_out = ops.in_bwd(g_y1, x, coef1, extra=extra, extra_pool=extra_pool, extra_scale=extra_scale)
        if j == L // 2:
            # data-parallel runs: the gradients of blocks L-1 .. L/2 (the 512-channel blocks: > 90 % of the parameter bytes) are
            # complete here, while the high-resolution blocks still to come take most of the backward's time
            hook = E.__dict__.get("_early_grad_hook")
            if hook is not None:
                flush_sums()
                hook(dict(grads))
    if fr is None:
        fr = ops.fromrgb_bwd(g_out, saved["x0"], saved["img"].float(), planar=True, defer=later)
    flush_sums()
    C0 = E.startf
    grads["FromRGB.from_rgb.weight"] = fr[:3].t().reshape(C0, 3, 1, 1)
    grads["FromRGB.from_rgb.bias"] = fr[3]
    out = []
    for name, p in E.named_parameters():
        g = grads.get(name)
        out.append(g.contiguous() if g is not None else None)
    return out
