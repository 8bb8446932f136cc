"""E_BIG.BE (reference model/E/E_BIG.py:93-227, conditional-BN encoder for BigGAN) forward / hand-written backward over the
HIP ops.

Per block (:129-169): x1 = lrelu(conv_1(bn_1(x)) + noise + bias); x2 = lrelu(conv_2(bn_2(x1)) + noise + bias);
residual = conv_3(bn_3(x)) and a SECOND lrelu on x2 when the channel count changes; out = avg_pool2d(x2 + residual).
Head: c_v = new_final_1(flat), z = new_final_2(c_v).  The conditional batch norm is a per-(b,c) affine in the conv
prologue (no activation in between), so its backward is linear: the data-gradient conv's epilogue yields
(sum g*x, sum g) = (d/da, d/db) per (b,c), from which the spectral-norm `scale` / `offset` weight gradients follow
(dense_wgrad + sn_weight_grad), and g_x = a * g is one streaming pass (`in_bwd` with coefficients (a, 0, 0)).
conv_3 / bn_3 commute with the average pool (1x1 conv, per-channel affine), so that branch is differentiated at the pooled
resolution.  Gradient enters through z only (E_align_s2.py:207-221; loss_c is commented out).
"""
import os

import torch

from . import ops
from .autograd_enc import _packed, draw_noises
from .biggan_generator import sn_weight_grad, sn_prepare, sn_cbn_linears
from .stylegan2_generator import _dt


def big_encoder_forward(E, img, cond_vector, noises=None, save=False, truncation=0.4):
    dt = _dt(E.compute_dtype)
    dev = img.device
    B, _, R, _ = img.shape
    training = E.training
    cond = cond_vector.detach().float().contiguous()
    if noises is None:
        noises = draw_noises(E, B, R, dev)
    cache = E.__dict__.setdefault("_pack_cache", {})
    if "_sn_skip" not in E.__dict__:
        # the last block stops after conv_1 (E_BIG.py:146-152): its batch_norm_2 / batch_norm_3 are never called by the reference
        last = len(E.decode_block) - 1
        E.__dict__["_sn_skip"] = (f"decode_block.{last}.batch_norm_2.", f"decode_block.{last}.batch_norm_3.")
    sn_prepare(E, training); sn_cbn_linears(E, cond)          # every conditional-BN linear's spectral norm (power iteration in train mode) at once
    fr = E.FromRGB.from_rgb
    x = ops.fromrgb(img.float(), fr.weight.detach(), fr.bias.detach(), dt, None)
    saved = {"img": img, "x0": x, "cond": cond, "blocks": []} if save else None
    ni = 0
    for j, blk in enumerate(E.decode_block):
        Cc, C2, H = blk.inputs, blk.outputs, R >> j
        c1 = {} if save else None
        a1, b1 = blk.batch_norm_1.affine(truncation, cond, training, c1)
        n1 = noises[ni].reshape(B, H, H).contiguous(); ni += 1
        x1 = ops.conv2d(x, _packed(cache, blk.conv_1, dt, ops.PACK_FWD, H), Cc, 3, in_scale=a1, in_shift=b1, noise=n1,
                        noise_w=blk.noise_weight_1.detach().reshape(-1), bias=blk.bias_1.detach().reshape(-1), act=ops.ACT_LRELU)
        rec = dict(x=x, a1=a1, b1=b1, c1=c1, n1=n1, x1=x1) if save else None
        if not blk.has_second_conv:
            if save:
                saved["blocks"].append(rec)
            x = x1
            break
        c2 = {} if save else None
        a2, b2 = blk.batch_norm_2.affine(truncation, cond, training, c2)
        n2 = noises[ni].reshape(B, H, H).contiguous(); ni += 1
        x2 = ops.conv2d(x1, _packed(cache, blk.conv_2, dt, ops.PACK_FWD, H), C2, 3, in_scale=a2, in_shift=b2, noise=n2,
                        noise_w=blk.noise_weight_2.detach().reshape(-1), bias=blk.bias_2.detach().reshape(-1), act=ops.ACT_LRELU)
        xp = ops.blend(x, pool=True)                                  # avg_pool2d of the block input (residual branch)
        if Cc != C2:
            c3 = {} if save else None
            a3, b3 = blk.batch_norm_3.affine(truncation, cond, training, c3)
            resp = ops.conv2d(xp, _packed(cache, blk.conv_3, dt, ops.PACK_FWD), C2, 1, in_scale=a3, in_shift=b3, bias=blk.conv_3.bias.detach())
            x2 = ops.blur_noise_act(x2, None, None, None, blur=False)          # the second leaky_relu of E_BIG.py:163
        else:
            a3 = b3 = c3 = None
            resp = xp
        xn = ops.blend(x2, z=resp, pool=True, alpha=1.0, beta=1.0)     # avg_pool2d(x2) + pooled residual
        if save:
            rec.update(a2=a2, b2=b2, c2=c2, n2=n2, x2=x2, xp=xp, a3=a3, b3=b3, c3=c3)
            saved["blocks"].append(rec)
        x = xn
    xo = ops.nhwc_to_nchw(x)
    c_v = z = None
    if E.biggan:
        flat = xo.reshape(B, -1)
        c_v = ops.linear(flat, E.new_final_1.weight.detach(), E.new_final_1.bias.detach())
        z = ops.linear(c_v, E.new_final_2.weight.detach(), E.new_final_2.bias.detach())
        if save:
            saved.update(flat=flat, c_v=c_v)
    return xo, c_v, z, saved


_NO_CBN_GROUP = bool(os.environ.get("DGE_NO_CBN_GROUP"))
_CBN_TAB = {}          # device -> (host bytes of the last entry table, its device copy, the pinned staging buffer)


def _cbn_param_grads(bn, bn_ctx, dots, cond, grads, name, pend=None):
    """dots [B,C,2] = (dL/da, dL/db) -> gradients of `scale.weight_orig` / `offset.weight_orig` (biggan BigGANBatchNorm :141-144).
    `pend` (a list): the work is only recorded; _cbn_param_grads_flush runs all norms of the backward in two launches."""
    if pend is not None and not _NO_CBN_GROUP:
        pend.append((bn, bn_ctx, dots, name))
        return
    g_a, g_b = dots[:, :, 0], dots[:, :, 1]
    g_scale = ((g_a - g_b * bn_ctx["mean"]) * bn_ctx["rstd"]).contiguous()
    g_off = g_b.contiguous()
    for key, gy, sn in (("scale", g_scale, bn_ctx["sn_sc"]), ("offset", g_off, bn_ctx["sn_of"])):
        w_live = getattr(bn, key).weight_orig
        gw = torch.empty_like(w_live)
        ops.dense_wgrad(gy, cond, gw)
        grads[f"{name}.{key}.weight_orig"] = sn_weight_grad(gw, w_live, sn)


def _cbn_param_grads_flush(pend, cond, grads):
    """All conditional-batch-norm parameter gradients of a backward: dge_cbn_sn_wgrad_group (two launches; the per-norm form above
    is ~20 torch launches per norm: dense_wgrad + the spectral-norm backward as tensor glue)."""
    if not pend:
        return
    import numpy as np
    dev = cond.device
    B, K = cond.shape
    rec = np.dtype([("dots", "u8"), ("mean", "u8"), ("rstd", "u8"), ("W", "u8"), ("u", "u8"), ("v", "u8"), ("sigma", "u8"), ("out", "u8"),
                    ("C", "i4"), ("kind", "i4"), ("row0", "i8")])
    assert rec.itemsize == ops.lib().dge_cbn_sn_wgrad_entry_size()
    n = 2 * len(pend)
    tab = np.zeros(n, dtype=rec)
    Cs = [int(d.shape[1]) for (_, _, d, _) in pend for _k in range(2)]
    rows = int(sum(Cs))
    out_all = torch.empty(rows * K, dtype=torch.float32, device=dev)
    keep = []
    r0 = 0
    for i, (bn, ctx, dots, name) in enumerate(pend):
        Cc = int(dots.shape[1])
        mean, rstd = ctx["mean"].float().contiguous(), ctx["rstd"].float().contiguous()
        keep += [mean, rstd]
        for kind, key, sn in ((0, "scale", ctx["sn_sc"]), (1, "offset", ctx["sn_of"])):
            w_live = getattr(bn, key).weight_orig
            if tuple(w_live.shape) != (Cc, K) or not w_live.is_contiguous():
                raise ops.DgeError(f"{name}.{key}: weight_orig {tuple(w_live.shape)} does not match [C={Cc}, K={K}]")
            u, v, sig = sn["u"].contiguous(), sn["v"].contiguous(), sn["sigma"].reshape(1)
            keep += [u, v, sig]
            e = tab[2 * i + kind]
            e["dots"], e["mean"], e["rstd"] = dots.data_ptr(), mean.data_ptr(), rstd.data_ptr()
            e["W"], e["u"], e["v"], e["sigma"] = w_live.data_ptr(), u.data_ptr(), v.data_ptr(), sig.data_ptr()
            e["out"] = out_all.data_ptr() + 4 * r0 * K
            e["C"], e["kind"], e["row0"] = Cc, kind, r0
            grads[f"{name}.{key}.weight_orig"] = out_all[r0 * K:(r0 + Cc) * K].view(Cc, K)
            r0 += Cc
    # the table is the same from step to step once the allocator has settled (arena buffers, cached blocks): uploaded only when it
    # changes, through a pinned buffer (a pageable host-to-device copy waits for the stream to drain - twice per step here)
    raw = tab.view(np.uint8)
    hit = _CBN_TAB.get(dev)
    if hit is not None and hit[0].shape == raw.shape and np.array_equal(hit[0], raw):
        tab_dev = hit[1]
    else:
        pin = torch.from_numpy(raw.copy()).pin_memory()
        tab_dev = pin.to(dev, non_blocking=True)
        _CBN_TAB[dev] = (raw.copy(), tab_dev, pin)
    rowdot = torch.empty(rows, dtype=torch.float32, device=dev)
    ops.check(ops.lib().dge_cbn_sn_wgrad_group(ops._p(tab_dev), n, rows, max(Cs), ops._f32(cond.contiguous()), B, K, ops._p(rowdot), ops._stream()),
              "dge_cbn_sn_wgrad_group")
    pend.clear()


_ZERO_BC = {}


def _affine_coef(a):
    """in_bwd coefficients (A, Bc, Cc) = (a, 0, 0): g_x = a * g.  (one launch: the zero planes are cached per shape)"""
    key = (tuple(a.shape), a.device)
    z = _ZERO_BC.get(key)
    if z is None:
        z = _ZERO_BC[key] = torch.zeros(a.shape, dtype=torch.float32, device=a.device)
    return torch.stack((a.float(), z, z), dim=-1)


def big_encoder_backward(E, saved, g_z, g_cv=None):
    if saved is None:
        raise RuntimeError("E_BIG forward ran without saved activations")
    cache = E.__dict__.setdefault("_pack_cache", {})
    dev = g_z.device
    B = g_z.shape[0]
    R = saved["img"].shape[2]
    dt = ops.dtype_of(saved["x0"])
    cond = saved["cond"]
    grads = {}
    pend = []          # conditional-batch-norm parameter gradients: recorded per norm, run grouped behind the block loop

    def lin_bwd(lin, gy, x, name):
        W = lin.weight.detach()
        gx = torch.empty_like(x)
        ops.linear_t(gy, W, gx)
        gw, gb = torch.empty_like(W), torch.empty_like(lin.bias)
        ops.dense_wgrad(gy, x, gw, gb)
        grads[name + ".weight"], grads[name + ".bias"] = gw, gb
        return gx
    g_cvt = lin_bwd(E.new_final_2, g_z.float().contiguous(), saved["c_v"], "new_final_2")
    if g_cv is not None:
        g_cvt = g_cvt + g_cv.float()
    g_flat = lin_bwd(E.new_final_1, g_cvt.contiguous(), saved["flat"], "new_final_1")
    L = len(saved["blocks"])
    C_last = E.decode_block[L - 1].inputs
    g_out = ops.nchw_to_nhwc(g_flat.view(B, C_last, R >> (L - 1), R >> (L - 1)), B, dt)
    for j in range(L - 1, -1, -1):
        blk, rec = E.decode_block[j], saved["blocks"][j]
        pre = f"decode_block.{j}."
        Cc, C2 = blk.inputs, blk.outputs
        H = R >> j
        x, x1 = rec["x"], rec["x1"]
        red1 = ops.zeros((Cc, 2), dev)
        extra, extra_pool, extra_scale = None, False, 1.0
        if blk.has_second_conv:
            has3 = Cc != C2
            red2 = ops.zeros((C2, 2), dev)
            # x2 = lrelu(pre2) [then a second lrelu when has3: slope 0.2*0.2 on the negative side]; out = avg_pool(x2 + res)
            g_pre2 = ops.act_bwd(g_out, rec["x2"], rec["n2"], pool=True, scale=0.25, red=red2, slope=0.04 if has3 else 0.2)
            grads[pre + "bias_2"] = red2[:, 0].reshape(1, C2, 1, 1)
            grads[pre + "noise_weight_2"] = red2[:, 1].reshape(1, C2, 1, 1)
            gW2 = ops.zeros(tuple(blk.conv_2.weight.shape), dev)
            ops.conv_wgrad(g_pre2, x1, gW2, rec["a2"], rec["b2"])
            grads[pre + "conv_2.weight"] = gW2
            dots2 = ops.zeros((B, Cc, 2), dev)
            g_u2 = ops.conv2d(g_pre2, _packed(cache, blk.conv_2, dt, ops.PACK_DGRAD, H), Cc, 3, stats=dots2, dot_src=x1)
            _cbn_param_grads(blk.batch_norm_2, rec["c2"], dots2, cond, grads, pre + "batch_norm_2", pend)
            g_pre1 = ops.in_bwd(g_u2, x1, _affine_coef(rec["a2"]), noise=rec["n1"], act=True, red=red1)
            if has3:
                xp = rec["xp"]
                grads[pre + "conv_3.bias"] = ops.chan_sum(g_out)
                gW3 = ops.zeros(tuple(blk.conv_3.weight.shape), dev)
                ops.conv_wgrad(g_out, xp, gW3, rec["a3"], rec["b3"])
                grads[pre + "conv_3.weight"] = gW3
                dots3 = ops.zeros((B, Cc, 2), dev)
                g_u3 = ops.conv2d(g_out, _packed(cache, blk.conv_3, dt, ops.PACK_DGRAD), Cc, 1, stats=dots3, dot_src=xp)
                _cbn_param_grads(blk.batch_norm_3, rec["c3"], dots3, cond, grads, pre + "batch_norm_3", pend)
                extra = ops.in_bwd(g_u3, xp, _affine_coef(rec["a3"]))          # a3 * g at the pooled resolution
            else:
                extra = g_out
            extra_pool, extra_scale = True, 0.25
        else:
            g_pre1 = ops.act_bwd(g_out, x1, rec["n1"], pool=False, scale=1.0, red=red1)
        grads[pre + "bias_1"] = red1[:, 0].reshape(1, Cc, 1, 1)
        grads[pre + "noise_weight_1"] = red1[:, 1].reshape(1, Cc, 1, 1)
        gW1 = ops.zeros(tuple(blk.conv_1.weight.shape), dev)
        ops.conv_wgrad(g_pre1, x, gW1, rec["a1"], rec["b1"])
        grads[pre + "conv_1.weight"] = gW1
        dots1 = ops.zeros((B, Cc, 2), dev)
        g_u1 = ops.conv2d(g_pre1, _packed(cache, blk.conv_1, dt, ops.PACK_DGRAD, H), Cc, 3, stats=dots1, dot_src=x)
        _cbn_param_grads(blk.batch_norm_1, rec["c1"], dots1, cond, grads, pre + "batch_norm_1", pend)
        g_out = ops.in_bwd(g_u1, x, _affine_coef(rec["a1"]), extra=extra, extra_pool=extra_pool, extra_scale=extra_scale)
    _cbn_param_grads_flush(pend, cond, grads)
    fr = ops.fromrgb_bwd(g_out, saved["x0"], saved["img"].float())
    C0 = E.startf
    grads["FromRGB.from_rgb.weight"] = fr[:, :3].reshape(C0, 3, 1, 1)
    grads["FromRGB.from_rgb.bias"] = fr[:, 3]
    out = []
    for name, _ in E.named_parameters():
        g = grads.get(name)
        out.append(g.contiguous() if g is not None else None)
    return out


class BigEncoderFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E, img, cond_vector, noises, *params):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise ops.DgeError("E_BIG: gradients w.r.t. the input image / condition vector are not implemented on the HIP path; "
                               "detach them")
        need = any(ctx.needs_input_grad[4:])
        _, c_v, z, saved = big_encoder_forward(E, img.detach(), cond_vector, noises, save=need)
        ctx.E, ctx.saved_acts = E, saved
        return c_v, z

    @staticmethod
    def backward(ctx, g_cv, g_z):
        return (None, None, None, None) + tuple(big_encoder_backward(ctx.E, ctx.saved_acts, g_z, g_cv))
