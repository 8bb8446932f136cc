"""Encoder forward / backward pipelines over the HIP ops, exposed as one autograd.Function
(PyTorch only routes the gradient tensors; every kernel is in libdge_hip.so)."""
import torch

from . import ops
from .stylegan2_generator import _dt


def _ver(w):
    return (w._version, w.data_ptr(), getattr(w, "_dge_gen", 0))


def _is_bwd_mode(mode):
    return (mode & 0xff) in (ops.PACK_DGRAD, ops.PACK_UPFOLD_DGRAD, ops.PACK_SG1_UP_DGRAD, ops.PACK_UPT2D_DGRAD)


def _stale(cache, only_bwd=False):
    return [(k, e) for k, e in cache.items() if isinstance(k, tuple) and len(k) == 3 and isinstance(e, list) and _ver(e[2]) != e[0]
            and (not only_bwd or _is_bwd_mode(k[1]))]


def _refresh(cache, stale, which):
    """One launch for all of `stale`; a descriptor table (device scratch) per kind of refresh, so that the two alternating sets of
    a step (everything / data-gradient copies only) each find their table already uploaded (a hipGraph capture cannot upload)."""
    key = ("_pack_scratch", which)
    cache[key] = ops.pack_conv_weights_multi([(e[2].detach(), k[1], k[2], 1.0, e[1]) for k, e in stale], cache.get(key))
    for k, e in stale:
        e[0] = _ver(e[2])


def refresh_packs(module):
    """Refreshes every stale packed copy of the module's conv weights now (on the current stream): EAlignStep runs this beside the
    generator's first pass at the start of an iteration instead of in front of the encoder's first conv."""
    cache = module.__dict__.get("_pack_cache")
    if cache:
        stale = _stale(cache)
        if stale:
            _refresh(cache, stale, "all")


def prime_pack_tables(module):
    """EAlignStep.capture, between the eager warm-up and the capture: uploads the descriptor table of the all-copies refresh (a
    capture cannot upload; a single warm-up iteration has only used the data-gradient table) and leaves every copy marked stale,
    so that the captured iteration re-packs exactly as a steady-state iteration does."""
    cache = module.__dict__.get("_pack_cache")
    if cache:
        stale = _stale(cache)
        if stale:
            _refresh(cache, stale, "all")
            for _, e in stale:
                e[0] = None


def _packed(cache, conv, dtype, mode, hw=None):
    """Packed copy of a conv weight, rebuilt when the parameter was updated in place.  An optimizer step makes EVERY copy of
    the module stale at once: the first stale hit of a FORWARD copy refreshes all of them in one launch
    (ops.pack_conv_weights_multi) - in place, the consumers of the old values are earlier on the same stream; the first stale
    hit of a data-gradient copy (the second backward of an E_align step, after the first optimizer step) refreshes the
    data-gradient copies only - the forward copies would be stale again before their next use.  `hw`: resolution the conv runs
    at (the low-resolution blocks keep their copies in fragment order for csrc/conv_small.hip)."""
    w = conv.weight
    if hw is not None:
        mode = ops.pack_mode_for(w, mode, hw, hw, dtype)
    key = (id(w), mode, dtype)
    ver = _ver(w)
    hit = cache.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1]
    if hit is None:
        cache[key] = [ver, ops.pack_conv_weight(w, mode, dtype, 1.0), w]
        return cache[key][1]
    bwd = _is_bwd_mode(mode)
    _refresh(cache, _stale(cache, only_bwd=bwd), "bwd" if bwd else "all")
    return cache[key][1]


def draw_noises(E, B, R, device):
    """The encoder's per-layer noise tensors ([B,1,r,r], two per block, one for the last): one generator launch for all of
    them, handed out as contiguous slices (the reference draws 17 separate CPU tensors, model/E/E.py:60,73 - quirk Q6).  Under
    data parallelism each tensor is the rank's slice of the draw a single process would make for the global batch."""
    shapes = []
    for j in range(E.layer_count):
        r = R >> j
        shapes.append((B, 1, r, r))
        if j != E.layer_count - 1:
            shapes.append((B, 1, r, r))
    if torch.device(device).type == "cpu":        # reference_noise mode: the reference's own sequence of CPU draws
        return [torch.randn(*s) for s in shapes]
    return ops.randn_rows(shapes, device)       # counter-based: the rank's rows of the global-batch draw (csrc/rng_kernels.hip)


def heads_layout(E, B, dev):
    """Static layout of the encoder's `inver_mod` heads for the grouped backward (dge_heads_bwd): entry order = (inver_mod1,
    inver_mod2) per block; statistics / their gradients in one flat buffer, parameter gradients in another; the device-side
    table holds the weight pointers (parameter storage does not move) and the column of each head in w (E.py:130-134)."""
    import numpy as np
    lins = []
    L = E.layer_count
    for j, blk in enumerate(E.decode_block):
        lins.append((blk.inver_mod1, 2 * (L - 1 - j) + 1))
        lins.append((blk.inver_mod2, 2 * (L - 1 - j)))
    key = (B, str(dev), tuple((l.weight.data_ptr(), l.bias.data_ptr()) for l, _ in lins))
    lay = E.__dict__.get("_heads_layout")
    if lay is not None and lay["key"] == key:
        return lay
    O = lins[0][0].weight.shape[0]
    rec = np.dtype([("W", "u8"), ("moff", "i8"), ("woff", "i8"), ("I", "i4"), ("gcol", "i4"), ("boff", "i4"), ("pad", "i4"), ("bias", "u8")])
    assert rec.itemsize == ops.lib().dge_head_entry_size()
    tab = np.zeros(len(lins), dtype=rec)
    moff = woff = 0
    items = []
    for i, (lin, col) in enumerate(lins):
        I = lin.weight.shape[1]
        assert lin.weight.shape[0] == O and lin.weight.is_contiguous()
        tab[i] = (lin.weight.data_ptr(), moff, woff, I, col * O, i * O, 0, lin.bias.data_ptr())
        items.append((moff, woff, i * O, I))
        moff += B * I
        woff += O * I
    lay = dict(key=key, tab=torch.from_numpy(tab.view(np.uint8).copy()).to(dev), items=items, n=len(lins), O=O, total_m=moff,
               total_w=woff, max_I=max(it[3] for it in items))
    E.__dict__["_heads_layout"] = lay
    return lay


def encoder_forward(E, img, noises=None, save=False):
    """BE.forward (reference model/E/E.py:122-136) + BEBlock.forward (:50-85)."""
    dt = _dt(E.compute_dtype)
    dev = img.device
    B, _, R, _ = img.shape
    if noises is None:
        noises = draw_noises(E, B, R, dev)
    cache = E.__dict__.setdefault("_pack_cache", {})
    zeros = lambda c: ops.SlotStats(B, c, dev)        # statistics slots are added by stats_finalize itself
    fr = E.FromRGB.from_rgb
    stats = zeros(E.startf)
    # (training: the image also leaves in pixel-major form - the last data gradient of the backward reduces the FromRGB parameter
    #  gradients against it, autograd_enc_bwd.py)
    want4 = save and E.startf == 16 and not ops.is_deterministic() and ops.conv_in_bwd_fromrgb_supported(B, R, R, 16, 16, dt)
    x = ops.fromrgb(img.float(), fr.weight.detach(), fr.bias.detach(), dt, stats.plain(), img4=want4)
    img4 = None
    if want4:
        x, img4 = x
    saved = {"img": img, "x0": x, "blocks": [], "img4": img4} if save else None
    ws, ni = [], 0
    L = E.layer_count
    lay = heads_layout(E, B, dev)
    musig_all = torch.empty(lay["total_m"], dtype=torch.float32, device=dev)     # all (mean, std) vectors, flat: grouped backward
    if save:
        saved["musig_all"] = musig_all

    def ms_slot(i):
        moff, _, _, I = lay["items"][i]
        return musig_all[moff:moff + B * I].view(B, I)
    for j, blk in enumerate(E.decode_block):
        Cc, C2 = blk.inputs, blk.outputs
        H = R >> j
        last = not blk.has_last_conv
        musig1, sc1, sh1 = ops.stats_finalize(stats, H * H, musig_out=ms_slot(2 * j))
        n1 = noises[ni].reshape(B, H, H).contiguous(); ni += 1
        st1 = zeros(Cc)
        x1 = ops.conv2d(x, _packed(cache, blk.conv_1, dt, ops.PACK_FWD, H), Cc, 3, in_scale=sc1, in_shift=sh1, noise=n1,
                        noise_w=blk.noise_weight_1.detach().reshape(-1), bias=blk.bias_1.detach().reshape(-1),
                        act=ops.ACT_LRELU, stats=st1)
        musig2, sc2, sh2 = ops.stats_finalize(st1, H * H, musig_out=ms_slot(2 * j + 1))
        rec = dict(x=x, musig1=musig1, sc1=sc1, sh1=sh1, n1=n1, x1=x1, musig2=musig2, sc2=sc2, sh2=sh2) if save else None
        has3 = Cc != C2
        nstats = zeros(C2) if not last else None
        if not last:
            n2 = noises[ni].reshape(B, H, H).contiguous(); ni += 1
            c2args = dict(in_scale=sc2, in_shift=sh2, noise=n2, noise_w=blk.noise_weight_2.detach().reshape(-1),
                          bias=blk.bias_2.detach().reshape(-1), act=ops.ACT_LRELU)
            m2 = None
            # the first blocks: conv_2 stores the 2x2 average pool of its result (and the signs for the backward) itself - the
            # full-resolution activation (537 MB at block 0) is neither written nor read back by a pooling pass
            pooled = has3 and ops.conv_pool_supported(B, H, H, Cc, C2, 3, dt)
            if pooled:
                r2 = ops.conv2d(x1, _packed(cache, blk.conv_2, dt, ops.PACK_FWD, H), C2, 3, pool_out=True, pool_mask=save, **c2args)
                x2, m2 = r2 if save else (r2, None)
            else:
                a2 = ops.conv2d(x1, _packed(cache, blk.conv_2, dt, ops.PACK_FWD, H), C2, 3, **c2args)
            if has3:
                if not pooled:
                    x2, m2 = ops.blend(a2, pool=True, mask=True) if save else (ops.blend(a2, pool=True), None)
                xp = ops.blend(x, pool=True)
                out = ops.conv2d(xp, _packed(cache, blk.conv_3, dt, ops.PACK_FWD), C2, 1, bias=blk.conv_3.bias.detach(),
                                 gain=0.889, addend=x2, add_scale=0.111, stats=nstats)
            else:
                xp = ops.blend(x, pool=True, alpha=0.889)
                if save:
                    out, m2 = ops.blend(a2, z=xp, pool=True, alpha=0.111, beta=1.0, stats=nstats.plain(), mask=True)
                else:
                    out = ops.blend(a2, z=xp, pool=True, alpha=0.111, beta=1.0, stats=nstats.plain())
            if save:
                # a2 itself is not kept: its backward (lrelu derivative + pool adjoint) needs only the signs (1 bit per element)
                rec.update(n2=n2, m2=m2, xp=xp if has3 else None)
        else:
            if has3:
                y2 = ops.blend(x1, sc=sc2, sh=sh2)
                out = ops.conv2d(x, _packed(cache, blk.conv_3, dt, ops.PACK_FWD), C2, 1, bias=blk.conv_3.bias.detach(),
                                 gain=0.889, addend=y2, add_scale=0.111)
            else:
                out = ops.blend(x1, z=x, sc=sc2, sh=sh2, alpha=0.111, beta=0.889)
        if save:
            saved["blocks"].append(rec)
        x, stats = out, nstats
    # every inver_mod head (w_l = musig_l @ W_l^T + b_l, E.py:51-53,64-66) in one launch: none of them feeds the trunk; column order
    # of w per E.py:130-134 (later / deeper blocks first) comes from the table
    w = torch.empty((B, 2 * L, lay["O"]), dtype=torch.float32, device=dev)
    ops.check(ops.lib().dge_heads_fwd(ops._p(lay["tab"]), lay["n"], ops._f32(musig_all), ops._p(w), w.stride(0), B, lay["O"],
                                      ops._stream()), "dge_heads_fwd")
    return ops.nhwc_to_nchw(x), w, saved


class EncoderFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E, img, noises, *params):
        if ctx.needs_input_grad[1]:
            # embedding_v2_styleGAN2.py back-propagates through E(imgs2) into G: only E_Blur has the image gradient here
            raise ops.DgeError("E.BE: the gradient w.r.t. the input image is not implemented on the HIP path (E_Blur provides it); "
                               "detach the image or call under torch.no_grad()")
        need = any(ctx.needs_input_grad[3:])
        xo, w, saved = encoder_forward(E, img.detach(), noises, save=need)
        ctx.E, ctx.saved_acts = E, saved
        ctx.set_materialize_grads(False)      # an output that no loss uses arrives as None in backward
        return xo, w

    @staticmethod
    def backward(ctx, g_x, g_w):
        from .autograd_enc_bwd import encoder_backward
        if g_x is not None:
            # a loss on the const output (space_loss(const2, const3) in embedding_v2_styleGAN2.py): refuse instead of dropping it
            raise ops.DgeError("E.BE: a gradient arrived through the encoder's activation output, which the hand-written backward "
                               "does not propagate (the E_align losses use w only, E_align_s2.py:203-221); detach it")
        if g_w is None:
            return (None, None, None) + (None,) * len(ctx.needs_input_grad[3:])
        grads = encoder_backward(ctx.E, ctx.saved_acts, g_w.contiguous())
        return (None, None, None) + tuple(grads)
