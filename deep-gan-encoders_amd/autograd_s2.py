"""StyleGAN2 synthesis pipeline over the HIP ops: forward (reference
model/stylegan2_generator.py:492-539) and the hand-written data-gradient w.r.t. wp that the
encoder training needs (E_align_s2.py:160,204).  G's own parameters never receive gradients
here: the reference computes and discards them, results for E are identical."""
import math

import torch

from . import ops
from .stylegan2_generator import _dt


def _layer_fwd(L, x, w_row, randomize_noise):
    dt = ops.dtype_of(x)
    packed, wsq = L._prepared(dt)
    s = L.style(w_row)
    d = ops.linear(s, wsq, None, 1.0, 1.0, L.eps, ops.LIN_RSQRT, 1.0, square_input=True)
    if randomize_noise:
        noise = torch.randn(x.shape[0], L.res, L.res, device=x.device)
    else:
        noise = L.noise.reshape(1, L.res, L.res)
    y = ops.conv2d(x, packed, L.out_c, 3, up=L.up, in_scale=s, out_scale=d, bias=L.bias, bias_scale=L.bscale,
                   noise=noise, noise_w=L.noise_strength.detach().reshape(1), act=L.act, gain=L.gain)
    return y, s, d, noise


def synthesis_run(mod, wp, randomize_noise=False, save=False):
    dt = _dt(mod.compute_dtype)
    B = wp.shape[0]
    results = {"wp": wp}
    x = ops.nchw_to_nhwc(mod.early_layer.const.detach(), B, dt)
    saved = {"const": x, "layers": [], "rgb": []} if save else None
    image = None
    for i in range(mod.num_layers - 1):
        L = getattr(mod, f"layer{i}")
        y, s, d, noise = _layer_fwd(L, x, wp[:, i], randomize_noise)
        results[f"style{i:02d}"] = s
        if save:
            saved["layers"].append(dict(y=y, s=s, d=d, noise=noise))
        x = y
        if i % 2 == 0:
            O_ = getattr(mod, f"output{i // 2}")
            srgb = O_.style(wp[:, i + 1])
            image = ops.torgb(x, O_.weight, srgb, O_.bias, image, O_.wscale)
            results[f"output_style{i // 2}"] = srgb
            if save:
                saved["rgb"].append(dict(s=srgb))
    results["image"] = image
    return results, saved


def _dgrad_weight(L, dtype):
    mode = ops.PACK_UPFOLD_DGRAD if L.up else ops.PACK_DGRAD
    key = ("dg", dtype, L.weight._version, L.weight.data_ptr(), getattr(L.weight, "_dge_gen", 0))
    c = L._cache.get("dg")
    if c is None or c[0] != key:
        c = (key, ops.pack_conv_weight(L.weight, mode, dtype, L.wscale))
        L._cache["dg"] = c
    return c[1]


def synthesis_backward(mod, wp, saved, g_image):
    """d(image)/d(wp) contracted with g_image [B,3,R,R] -> g_wp [B,num_layers,512]."""
    B = wp.shape[0]
    dev = wp.device
    nl = mod.num_layers
    g_wp = ops.zeros((B, nl, mod.w_space_dim), dev)
    layers = saved["layers"]
    dt = ops.dtype_of(saved["const"])
    g_img = g_image.float().contiguous()
    # top layer: its output only feeds the last toRGB (image_k = rgb_k + up(image_{k-1}), :515-522)
    top = nl - 2
    Ot = getattr(mod, f"output{top // 2}")
    g_x, g_srgb = ops.torgb_bwd(g_img, layers[top]["y"], Ot.weight.detach().reshape(3, -1), saved["rgb"][top // 2]["s"], Ot.wscale)
    ops.linear_t(g_srgb, Ot.style.weight.detach(), g_wp[:, top + 1], scale=Ot.style.wscale, accumulate=True)
    if top // 2 > 0:
        g_img = ops.up2_bwd(g_img)
    for i in range(top, -1, -1):
        L = getattr(mod, f"layer{i}")
        rec = layers[i]
        # ---- backward through noise/bias/act/demod of layer i
        R = ops.zeros((B, L.out_c, 3), dev)
        g_y = ops.modconv_bwd_prep(g_x, rec["y"], rec["d"], rec["noise"], L.gain, R)
        x_in = saved["const"] if i == 0 else layers[i - 1]["y"]
        # toRGB gradient of the previous (even) layer joins through the epilogue addend
        addend = None
        if i >= 1 and (i - 1) % 2 == 0:
            kp = (i - 1) // 2
            Op = getattr(mod, f"output{kp}")
            # g_img has already been brought down to this resolution by up2_bwd above
            addend, g_srgb = ops.torgb_bwd(g_img, x_in, Op.weight.detach().reshape(3, -1), saved["rgb"][kp]["s"], Op.wscale)
            ops.linear_t(g_srgb, Op.style.weight.detach(), g_wp[:, i], scale=Op.style.wscale, accumulate=True)
            if kp > 0:
                g_img = ops.up2_bwd(g_img)
        st = ops.zeros((B, L.in_c, 2), dev)
        g_xprev = ops.conv2d(g_y, _dgrad_weight(L, dt), L.in_c, 3, in_s2d=L.up, out_scale=rec["s"], addend=addend,
                             add_scale=1.0, stats=st, dot_src=x_in)
        # ---- style / demodulation gradients -> g_wp[:, i]
        t = ops.demod_bwd(R, rec["d"], L.bias.detach(), L.noise_strength.detach().reshape(1), L.bscale)
        _, wsq = L._prepared(dt)
        gs_view = st.view(B, -1)      # [B, 2*Cin]: element (b, 2*i) = g_s[b,i]
        ops.linear_t(t, wsq, gs_view, mul=rec["s"], accumulate=True, incy=2, ldy=2 * L.in_c)
        ops.linear_t(gs_view, L.style.weight.detach(), g_wp[:, i], scale=L.style.wscale, accumulate=True, incx=2,
                     ldx=2 * L.in_c, O=L.in_c)
        g_x = g_xprev
    return g_wp


class SynthesisFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, wp, randomize_noise):
        need = ctx.needs_input_grad[1]
        results, saved = synthesis_run(mod, wp.detach(), randomize_noise, save=need)
        ctx.mod, ctx.saved_acts, ctx.wp = mod, saved, wp.detach()
        ctx.keys = [k for k in results if k not in ("wp", "image")]
        outs = [results[k] for k in ctx.keys]
        ctx.mark_non_differentiable(*outs)
        return (results["image"], *outs)

    @staticmethod
    def backward(ctx, g_image, *unused):
        if ctx.saved_acts is None:
            raise RuntimeError("synthesis was run without saved activations")
        g_wp = synthesis_backward(ctx.mod, ctx.wp, ctx.saved_acts, g_image)
        return None, g_wp, None


def synthesis_forward(mod, wp, randomize_noise=False):
    """SynthesisModule.forward: conv layer i is driven by wp[:, i]; the toRGB of block k by
    wp[:, 2k+1] (reference :511-517).  Returns the reference's result dict."""
    wp = wp if (wp.dtype == torch.float32 and wp.is_contiguous()) else wp.float().contiguous()
    if wp.requires_grad and torch.is_grad_enabled():
        outs = SynthesisFunction.apply(mod, wp, randomize_noise)
        keys = [f"style{i:02d}" for i in range(mod.num_layers - 1)]
        results = {"wp": wp}
        # keys are produced in the same order as synthesis_run builds them
        names = []
        for i in range(mod.num_layers - 1):
            names.append(f"style{i:02d}")
            if i % 2 == 0:
                names.append(f"output_style{i // 2}")
        for n, t in zip(names, outs[1:]):
            results[n] = t
        results["image"] = outs[0]
        return results
    results, _ = synthesis_run(mod, wp, randomize_noise, save=False)
    return results
