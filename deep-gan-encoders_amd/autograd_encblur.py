"""E_Blur encoder (reference model/E/E_Blur.py:16-134) forward / backward pipelines over the HIP ops.

Unlike E.BE inside E_align, the inversion loop (embedding_img.py:86-127) back-propagates through BOTH encoder
outputs (latents w and the 4x4 `const`) and through the encoder's INPUT (the second call encodes a generated image
that carries a gradient), so this backward returns parameter gradients and the image gradient.

Block forward (not last): IN1 -> conv_1 +noise+bias -> lrelu (x1) -> IN2 -> blur (y2) ->
  fused_scale (block resolution label >= 128): conv_2 stride 2 with transform_kernel == avg_pool(conv3x3) (lreq.py:145-147),
      then +noise+bias -> lrelu at half resolution;
  else: conv3x3 +noise+bias -> lrelu -> avg_pool.
out = 0.111*x2 + 0.889*(conv_3)(avg_pool(x)).  Last block: out = 0.111*IN2(x1) + 0.889*x.
"""
import torch

from . import ops
from .autograd_enc import _packed, draw_noises
from .autograd_enc_bwd import _linear_backward
from .stylegan2_generator import _dt


def blur_noises(E, B, R, dev):
    """Noise tensors in the reference's draw order; fused-scale blocks draw the second one at half resolution."""
    noises = draw_noises(E, B, R, dev)
    ni = 0
    for j, blk in enumerate(E.decode_block):
        ni += 1
        if blk.has_last_conv:
            if blk.fused_scale:
                r = (R >> j) // 2
                noises[ni] = ops.randn((B, 1, r, r), dev)
            ni += 1
    return noises


def blur_encoder_forward(E, img, noises=None, save=False):
    dt = _dt(E.compute_dtype)
    dev = img.device
    B, _, R, _ = img.shape
    if noises is None:
        noises = blur_noises(E, B, R, dev)
    cache = E.__dict__.setdefault("_pack_cache", {})
    zeros = lambda c: ops.zeros((B, c, 2), dev)
    fr = E.FromRGB.from_rgb
    stats = zeros(E.startf)
    x = ops.fromrgb(img.float(), fr.weight.detach(), fr.bias.detach(), dt, stats)
    saved = {"img": img, "x0": x, "blocks": []} if save else None
    ws, ni = [], 0
    for j, blk in enumerate(E.decode_block):
        Cc, C2, H = blk.inputs, blk.outputs, R >> j
        last = not blk.has_last_conv
        has3 = Cc != C2
        musig1, sc1, sh1 = ops.stats_finalize(stats, H * H)
        w1 = ops.linear(musig1, blk.inver_mod1.weight.detach(), blk.inver_mod1.bias.detach())
        n1 = noises[ni].reshape(B, H, H).contiguous(); ni += 1
        st1 = zeros(Cc)
        x1 = ops.conv2d(x, _packed(cache, blk.conv_1, dt, ops.PACK_FWD, H), Cc, 3, in_scale=sc1, in_shift=sh1, noise=n1,
                        noise_w=blk.noise_weight_1.detach().reshape(-1), bias=blk.bias_1.detach().reshape(-1),
                        act=ops.ACT_LRELU, stats=st1)
        musig2, sc2, sh2 = ops.stats_finalize(st1, H * H)
        w2 = ops.linear(musig2, blk.inver_mod2.weight.detach(), blk.inver_mod2.bias.detach())
        rec = dict(x=x, musig1=musig1, sc1=sc1, sh1=sh1, n1=n1, x1=x1, musig2=musig2, sc2=sc2, sh2=sh2) if save else None
        nstats = zeros(C2) if not last else None
        if not last:
            y2 = ops.blur_noise_act(ops.blend(x1, sc=sc2, sh=sh2), None, None, None, blur=True, act=False)   # blur(IN2(x1))
            wpk = _packed(cache, blk.conv_2, dt, ops.PACK_FWD, H)
            n2 = noises[ni]; ni += 1
            nw2, b2 = blk.noise_weight_2.detach().reshape(-1), blk.bias_2.detach().reshape(-1)
            if blk.fused_scale:        # conv(s2, transform_kernel) == pool(conv); noise/bias/lrelu at half resolution
                n2 = n2.reshape(B, H // 2, H // 2).contiguous()
                t = ops.blend(ops.conv2d(y2, wpk, C2, 3), pool=True)
                a2 = x2 = ops.blur_noise_act(t, n2, nw2, b2, blur=False)
            else:
                n2 = n2.reshape(B, H, H).contiguous()
                a2 = ops.conv2d(y2, wpk, C2, 3, noise=n2, noise_w=nw2, bias=b2, act=ops.ACT_LRELU)
                x2 = ops.blend(a2, pool=True)
            xp = ops.blend(x, pool=True)
            if has3:
                out = ops.conv2d(xp, _packed(cache, blk.conv_3, dt, ops.PACK_FWD), C2, 1, bias=blk.conv_3.bias.detach(),
                                 gain=0.889, addend=x2, add_scale=0.111, stats=nstats)
            else:
                out = ops.blend(x2, z=xp, alpha=0.111, beta=0.889, stats=nstats)
            if save:
                rec.update(y2=y2, n2=n2, a2=a2, xp=xp if has3 else None)
        else:
            if has3:
                raise NotImplementedError("E_Blur: last block with a channel change is not reachable with maxf-clamped widths")
            out = ops.blend(x1, z=x, sc=sc2, sh=sh2, alpha=0.111, beta=0.889)
        if save:
            saved["blocks"].append(rec)
        ws = [w2, w1] + ws
        x, stats = out, nstats
    return ops.nhwc_to_nchw(x), torch.stack(ws, dim=1), saved


def blur_encoder_backward(E, saved, g_w, g_const=None, need_img=False):
    """-> (gradients for E.parameters() in registration order, image gradient [B,3,R,R] or None)."""
    cache = E.__dict__.setdefault("_pack_cache", {})
    dev = g_w.device
    L = E.layer_count
    B = g_w.shape[0]
    grads = {}
    R = saved["img"].shape[2]
    dt = ops.dtype_of(saved["x0"])
    g_out = None
    if g_const is not None:
        g_out = ops.nchw_to_nhwc(g_const.float().contiguous(), B, dt) if g_const.shape[0] == B else None
    for j in range(L - 1, -1, -1):
        blk = E.decode_block[j]
        rec = saved["blocks"][j]
        pre = f"decode_block.{j}."
        Cc, C2 = blk.inputs, blk.outputs
        H = R >> j
        N = H * H
        last = not blk.has_last_conv
        has3 = Cc != C2
        g_w2, g_w1 = g_w[:, 2 * (L - 1 - j)], g_w[:, 2 * (L - 1 - j) + 1]
        gms2 = _linear_backward(blk.inver_mod2, g_w2, rec["musig2"], grads, pre + "inver_mod2")
        gms1 = _linear_backward(blk.inver_mod1, g_w1, rec["musig1"], grads, pre + "inver_mod1")
        x, x1 = rec["x"], rec["x1"]
        extra, extra_pool, extra_scale = None, False, 1.0
        if not last:
            if g_out is None:
                raise RuntimeError("non-final encoder block without an output gradient")
            red2 = ops.zeros((C2, 3 if has3 else 2), dev)      # {bias_2, noise_weight_2 [, sum g_out -> conv_3.bias]}
            if blk.fused_scale:
                g_t = ops.act_bwd(g_out, rec["a2"], rec["n2"], pool=False, scale=0.111, red=red2)      # lrelu' at half resolution
                g_c2 = ops.nearest_up2(g_t, 0.25)                                                  # adjoint of the 2x2 average
            else:
                g_c2 = ops.act_bwd(g_out, rec["a2"], rec["n2"], pool=True, scale=0.111 * 0.25, red=red2)
            grads[pre + "bias_2"] = red2[:, 0].reshape(1, C2, 1, 1)
            grads[pre + "noise_weight_2"] = red2[:, 1].reshape(1, C2, 1, 1)
            gW2 = ops.zeros(tuple(blk.conv_2.weight.shape), dev)
            ops.conv_wgrad(g_c2, rec["y2"], gW2)
            grads[pre + "conv_2.weight"] = gW2
            g_y2b = ops.conv2d(g_c2, _packed(cache, blk.conv_2, dt, ops.PACK_DGRAD, H), Cc, 3)
            g_y2 = ops.blur_noise_act(g_y2b, None, None, None, blur=True, act=False)                 # Blur is self-adjoint
            dots2 = ops.dot_stats(g_y2, x1)
            if has3:
                grads[pre + "conv_3.bias"] = red2[:, 2] * 0.889
                gW3 = ops.zeros(tuple(blk.conv_3.weight.shape), dev)
                ops.conv_wgrad(g_out, rec["xp"], gW3)
                grads[pre + "conv_3.weight"] = ops.scale_(gW3, 0.889)
                extra = ops.conv2d(g_out, _packed(cache, blk.conv_3, dt, ops.PACK_DGRAD), Cc, 1, gain=0.889)
                extra_pool, extra_scale = True, 0.25
            else:
                extra, extra_pool, extra_scale = g_out, True, 0.889 * 0.25
        else:
            if g_out is not None:            # out = 0.111*IN2(x1) + 0.889*x
                g_y2 = ops.blend(g_out, alpha=0.111)
                dots2 = ops.dot_stats(g_y2, x1)
                extra, extra_pool, extra_scale = g_out, False, 0.889
            else:
                g_y2, dots2 = None, None
        coef2 = ops.in_bwd_coef(dots2, gms2, rec["musig2"], rec["sc2"], rec["sh2"], N)
        red1 = ops.zeros((Cc, 2), dev)
        g_pre1 = ops.in_bwd(g_y2, x1, coef2, noise=rec["n1"], act=True, red=red1)
        grads[pre + "bias_1"] = red1[:, 0].reshape(1, Cc, 1, 1)
        grads[pre + "noise_weight_1"] = red1[:, 1].reshape(1, Cc, 1, 1)
        gW1 = ops.zeros(tuple(blk.conv_1.weight.shape), dev)
        ops.conv_wgrad(g_pre1, x, gW1, rec["sc1"], rec["sh1"])
        grads[pre + "conv_1.weight"] = gW1
        dots1 = ops.zeros((B, Cc, 2), dev)
        g_y1 = ops.conv2d(g_pre1, _packed(cache, blk.conv_1, dt, ops.PACK_DGRAD, H), Cc, 3, stats=dots1, dot_src=x)
        coef1 = ops.in_bwd_coef(dots1, gms1, rec["musig1"], rec["sc1"], rec["sh1"], N)
        g_out = ops.in_bwd(g_y1, x, coef1, extra=extra, extra_pool=extra_pool, extra_scale=extra_scale)
    fr = ops.fromrgb_bwd(g_out, saved["x0"], saved["img"].float())
    C0 = E.startf
    grads["FromRGB.from_rgb.weight"] = fr[:, :3].reshape(C0, 3, 1, 1)
    grads["FromRGB.from_rgb.bias"] = fr[:, 3]
    g_img = ops.fromrgb_dgrad(g_out, saved["x0"], E.FromRGB.from_rgb.weight.detach()) if need_img else None
    out = []
    for name, p in E.named_parameters():
        g = grads.get(name)
        out.append(g.contiguous() if g is not None else None)
    return out, g_img


class BlurEncoderFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E, img, noises, *params):
        need = any(ctx.needs_input_grad[1:])
        xo, w, saved = blur_encoder_forward(E, img.detach(), noises, save=need)
        ctx.E, ctx.saved_acts = E, saved
        ctx.need_img = ctx.needs_input_grad[1]
        return xo, w

    @staticmethod
    def backward(ctx, g_x, g_w):
        if ctx.saved_acts is None:
            raise RuntimeError("E_Blur forward ran without saved activations")
        B = ctx.saved_acts["img"].shape[0]
        if g_w is None:
            g_w = torch.zeros((B, 2 * ctx.E.layer_count, ctx.E.latent_size), dtype=torch.float32, device=ctx.saved_acts["img"].device)
        grads, g_img = blur_encoder_backward(ctx.E, ctx.saved_acts, g_w.float().contiguous(), g_x, need_img=ctx.need_img)
        if _DIRECT_ACCUMULATE:
            accumulate_param_grads(ctx.E, grads)
            return (None, g_img, None) + (None,) * len(grads)
        return (None, g_img, None) + tuple(grads)


import os as _os
_DIRECT_ACCUMULATE = _os.environ.get("DGE_AUTOGRAD_ACCUMULATE") != "1"


def accumulate_param_grads(E, grads):
    """Adds the parameter gradients of ONE encoder call to `.grad` from inside the backward instead of handing them to autograd.
    The inversion loop (embedding_img.py:86-88) calls the encoder twice per iteration, so autograd's AccumulateGrad node clones every
    gradient of the first call (they are views of the step's zero-filled arena) and adds every gradient of the second: 2 x 105 launches
    of ~4 us per backward at batch 1, where the loop is paced by launches (profiles/r05_embed_kernel_stats.txt).  Here: one
    multi-tensor copy into a persistent flat buffer for the parameters without a gradient, one multi-tensor add for those with one.
    Same sums in the same order (first call, then second)."""
    import torch
    params = list(E.parameters())
    lay = E.__dict__.get("_grad_flat")
    total = sum(p.numel() for p in params)
    if lay is None or lay["total"] != total or lay["buf"].device != params[0].device:
        buf = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        views, o = [], 0
        for p in params:
            views.append(buf[o:o + p.numel()].view_as(p)); o += p.numel()
        lay = E.__dict__["_grad_flat"] = dict(total=total, buf=buf, views=views)
    first_dst, first_src, add_dst, add_src = [], [], [], []
    for p, v, g in zip(params, lay["views"], grads):
        if g is None:
            continue
        if p.grad is None:
            first_dst.append(v); first_src.append(g.reshape(p.shape))
            p.grad = v
        else:
            add_dst.append(p.grad); add_src.append(g.reshape(p.shape))
    if first_dst:
        torch._foreach_copy_(first_dst, first_src)
    if add_dst:
        torch._foreach_add_(add_dst, add_src)
