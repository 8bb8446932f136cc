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
        noise = ops.randn((x.shape[0], L.res, L.res), x.device)
    else:
        noise = L.noise.reshape(1, L.res, L.res)
    y = ops.conv2d(x, packed, L.out_c, 3, up=L.up, in_scale=s, out_scale=d, bias=L.bias, bias_scale=L.bscale,
                   noise=noise, noise_w=L.noise_strength.detach().reshape(1), act=L.act, gain=L.gain)
    return y, s, d, noise


def _style_tables(mod, B, dt):
    """Row-concatenated style weights / demodulation tables of all 17 + 9 modulated blocks (cached; G is frozen in E_align)
    so that every style vector and demodulation factor of a pass comes from two launches (dge_linear_rows, dge_demod_rows)."""
    nl = mod.num_layers
    groups = [(getattr(mod, f"layer{i}"), i) for i in range(nl - 1)] + [(getattr(mod, f"output{k}"), 2 * k + 1) for k in range(nl // 2)]
    key = (B, str(dt)) + tuple((p._version, p.data_ptr()) for L, _ in groups for p in (L.style.weight, L.style.bias, L.weight))
    hit = mod.__dict__.get("_style_tab")
    if hit is not None and hit[0] == key:
        return hit[1]
    dev = groups[0][0].weight.device
    K = mod.w_space_dim
    Wc = torch.cat([L.style.weight.detach() for L, _ in groups]).contiguous()
    bc = torch.cat([L.style.bias.detach() for L, _ in groups]).contiguous()
    xoff, ybase, ybs, s_off = [], [], [], []
    off = 0
    for L, row in groups:
        C = L.in_c
        s_off.append(B * off)
        xoff += [row * K] * C
        ybase += [B * off + c for c in range(C)]
        ybs += [C] * C
        off += C
    woff, sbase, cin, dbase, dbs, d_off, wsq = [], [], [], [], [], [], []
    doff, wpos = 0, 0
    for gi, (L, _) in enumerate(groups[:nl - 1]):
        _, wq = L._prepared(dt)
        wsq.append(wq.reshape(-1))
        d_off.append(B * doff)
        for o in range(L.out_c):
            woff.append(wpos + o * L.in_c); sbase.append(s_off[gi]); cin.append(L.in_c)
            dbase.append(B * doff + o); dbs.append(L.out_c)
        doff += L.out_c
        wpos += L.out_c * L.in_c
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
    st0 = groups[0][0].style
    for L, _ in groups:
        assert L.style.wscale == st0.wscale and L.style.bscale == st0.bscale and L.style.additional_bias == st0.additional_bias
    tab = dict(groups=groups, Wc=Wc, bc=bc, xoff=i32(xoff), ybase=i32(ybase), ybs=i32(ybs), s_off=s_off, s_total=B * off,
               wsq=torch.cat(wsq).contiguous(), woff=i32(woff), sbase=i32(sbase), cin=i32(cin), dbase=i32(dbase), dbs=i32(dbs),
               d_off=d_off, d_total=B * doff, wscale=st0.wscale, bscale=st0.bscale, add=st0.additional_bias, eps=groups[0][0].eps)
    mod.__dict__["_style_tab"] = (key, tab)
    return tab


def synthesis_run(mod, wp, randomize_noise=False, save=False):
    dt = _dt(mod.compute_dtype)
    B = wp.shape[0]
    results = {"wp": wp}
    nl = mod.num_layers
    # ---- all styles and demodulation factors of the pass: two launches instead of 2*17 + 9
    tab = _style_tables(mod, B, dt)
    wpc = wp if (wp.dtype == torch.float32 and wp.is_contiguous()) else wp.float().contiguous()
    s_all = torch.empty(tab["s_total"], dtype=torch.float32, device=wp.device)
    ops.linear_rows(wpc, nl * mod.w_space_dim, tab["xoff"], tab["Wc"], tab["bc"], s_all, tab["ybase"], tab["ybs"], B,
                    tab["wscale"], tab["bscale"], tab["add"])
    d_all = torch.empty(tab["d_total"], dtype=torch.float32, device=wp.device)
    ops.demod_rows(s_all, tab["wsq"], tab["woff"], tab["sbase"], tab["cin"], d_all, tab["dbase"], tab["dbs"], B, tab["eps"])
    s_of = lambda g, C: s_all[tab["s_off"][g]:tab["s_off"][g] + B * C].view(B, C)
    x = ops.nchw_to_nhwc(mod.early_layer.const.detach(), B, dt)
    saved = {"const": x, "layers": [], "rgb": []} if save else None
    image = None
    for i in range(nl - 1):
        L = getattr(mod, f"layer{i}")
        s = s_of(i, L.in_c)
        d = d_all[tab["d_off"][i]:tab["d_off"][i] + B * L.out_c].view(B, L.out_c)
        if randomize_noise:
            noise = ops.randn((x.shape[0], L.res, L.res), x.device)
        else:
            noise = L.noise.reshape(1, L.res, L.res)
        # the toRGB of the top layers rides in their conv epilogue where the streaming kernel offers it (512^2 / 1024^2: the activation
        # is not read back by a toRGB pass, and the last one is not even written when nothing else reads it); the skip image is
        # added afterwards
        fuse_rgb = (i % 2 == 0 and image is not None and not L.up and L.bias is not None and
                    ops.conv_rgb_supported(B, L.res, L.res, L.in_c, L.out_c, 3, dt))
        rgb = None
        if fuse_rgb:
            O_ = getattr(mod, f"output{i // 2}")
            srgb = s_of(nl - 1 + i // 2, O_.in_c)
            rgb = dict(w=O_.weight.detach().reshape(3, -1), style=srgb, bias=O_.bias.detach(), wscale=O_.wscale,
                       out=torch.empty((B, 3, L.res, L.res), dtype=torch.float32, device=x.device), skip_y=(not save) and i == nl - 2)
        y = L.conv(x, s, d, noise, dt, rgb=rgb)
        results[f"style{i:02d}"] = s
        if save:
            saved["layers"].append(dict(y=y, s=s, d=d, noise=noise))
        x = y
        if fuse_rgb:
            image = ops.rgb_upsample_add(rgb["out"], image)
            results[f"output_style{i // 2}"] = srgb
            if save:
                saved["rgb"].append(dict(s=srgb))
        elif i % 2 == 0:
            O_ = getattr(mod, f"output{i // 2}")
            srgb = s_of(nl - 1 + i // 2, O_.in_c)
            image = ops.torgb(x, O_.weight, srgb, O_.bias, image, O_.wscale)
            results[f"output_style{i // 2}"] = srgb
            if save:
                saved["rgb"].append(dict(s=srgb))
    results["image"] = image
    return results, saved


def _up_phase_form(L):
    """Up layers whose data gradient runs in phase form (FIR^T pass + 4-tap conv on the t grid, dge_fir_t2d / in_t2d: 16 tap-units
    per input pixel instead of the 36 of the folded space-to-depth form): the MFMA-bound ones, 32^2 .. 128^2 input (measured at
    batch 8, tools/perf_t2d.py, folded -> FIR pass + conv: 512->512 @32^2 221 -> 174 us, 256<-512 @64^2 310 -> 243, 128<-256 @128^2
    344 -> 313; at 256^2 / 512^2 the launch is bound by its epilogue and the extra pass loses: 430 -> 504, 951 -> 1223).  The small
    grids stay on the folded form (low-resolution kernel)."""
    import os
    return L.up and 32 <= L.res // 2 <= 128 and L.in_c >= 64 and L.in_c % 32 == 0 and L.out_c % 8 == 0 and not os.environ.get("DGE_NO_T2D")


def _dgrad_weight(L, dtype):
    hg = L.res // 2 if L.up else L.res          # grid the data-gradient conv runs on (space-to-depth grid for the up layers)
    if _up_phase_form(L):
        mode = ops.PACK_UPT2D_DGRAD
    else:
        mode = ops.pack_mode_for(L.weight, ops.PACK_UPFOLD_DGRAD if L.up else ops.PACK_DGRAD, hg, hg, dtype)
    key = ("dg", dtype, mode, L.weight._version, L.weight.data_ptr(), getattr(L.weight, "_dge_gen", 0))
    c = L._cache.get("dg")
    if c is None or c[0] != key:
        c = (key, ops.pack_conv_weight(L.weight, mode, dtype, L.wscale))
        L._cache["dg"] = c
    return c[1]


def _pp_dgrad(L, B, hg, dtype):
    """the data gradient of this layer runs on conv_pp (GEMM K = out channels, x 4 phases for the folded up layer; N = in channels)"""
    import os
    K = 4 * L.out_c if L.up else L.out_c
    return (not L.up or L.out_c % 32 == 0) and ops.conv_pp_supported(B, hg, hg, K, L.in_c, dtype) and not os.environ.get("DGE_NO_PP_DG")


def _dgrad_weight_pp(L, d, t2d=False):
    """data-gradient weight image of conv_pp.  Stride 1 / folded up layer: per sample, W'[b] = bf16(w * wscale * d[b, o]) (the up layer
    through its folded f32 rows, packed once per weight version); phase form: one shared 4-tap image, cached per weight version"""
    if not L.up:
        return ops.pack_conv_pp(L.weight, L.wscale, in_scale=d, dgrad=True)
    name = "dgpp_t2d" if t2d else "dgpp"
    key = (name, L.weight._version, L.weight.data_ptr(), getattr(L.weight, "_dge_gen", 0))
    c = L._cache.get(name)
    if c is None or c[0] != key:
        rows = ops.pack_conv_weight(L.weight, ops.PACK_UPT2D_DGRAD if t2d else ops.PACK_UPFOLD_DGRAD, ops.F32, L.wscale)
        c = (key, ops.pack_conv_pp_rows(rows, L.in_c, t2d=True) if t2d else rows)
        L._cache[name] = c
    return c[1] if t2d else ops.pack_conv_pp_rows(c[1], L.in_c, in_scale=d, in_period=L.out_c)


def synthesis_backward(mod, wp, saved, g_image):
    """d(image)/d(wp) contracted with g_image [B,3,R,R] -> g_wp [B,num_layers,512].

    The gradient travels down the chain as g_z (w.r.t. the pre-activation z = yraw*d + noise*ns + bias of each layer,
    stylegan2_generator.py:908-921): the data-gradient conv of layer i+1 differentiates layer i's noise / bias / lrelu tail in its own
    epilogue (it streams layer i's stored output as `dot_src` anyway) and also leaves the two per-(b,c) sums of the demodulation
    gradient there; the demodulation factor d multiplies g_z in the NEXT conv's prologue.  `fused=False` (deterministic mode) runs
    the same math with the tail backward as a pass of its own (dge_modconv_bwd_prep)."""
    B = wp.shape[0]
    dev = wp.device
    nl = mod.num_layers
    g_wp = ops.zeros((B, nl, mod.w_space_dim), dev)
    layers = saved["layers"]
    dt = ops.dtype_of(saved["const"])
    g_img = g_image.float().contiguous()
    fused = not ops.is_deterministic()
    # top layer: its output only feeds the last toRGB (image_k = rgb_k + up(image_{k-1}), :515-522)
    top = nl - 2
    Lt, Ot = getattr(mod, f"layer{top}"), getattr(mod, f"output{top // 2}")
    P = None
    if fused:
        g_x, g_srgb, P = ops.torgb_bwd_prep(g_img, layers[top]["y"], Ot.weight.detach().reshape(3, -1), saved["rgb"][top // 2]["s"], Ot.wscale,
                                            layers[top]["noise"], Lt.noise_strength.detach().reshape(1), Lt.gain)
    else:
        g_x, g_srgb = ops.torgb_bwd(g_img, layers[top]["y"], Ot.weight.detach().reshape(3, -1), saved["rgb"][top // 2]["s"], Ot.wscale)
    # fused mode: every style gradient leaves in ONE launch at the end (dge_s2_style_grads) - none of them feeds the chain
    blocks = [] if fused else None
    if fused:
        blocks.append(dict(gs=g_srgb, wstyle=Ot.style.weight.detach(), row=top + 1))
    else:
        ops.linear_t(g_srgb, Ot.style.weight.detach(), g_wp[:, top + 1], scale=Ot.style.wscale, accumulate=True)
    if top // 2 > 0:
        g_img = ops.up2_bwd(g_img)
    for i in range(top, -1, -1):
        L = getattr(mod, f"layer{i}")
        rec = layers[i]
        # ---- backward through noise/bias/act/demod of layer i
        if P is not None:
            g_y, d_in = g_x, rec["d"]                  # g_x is g_z already (with its sums in P); d joins in the conv prologue
        else:
            R = ops.zeros((B, L.out_c, 3), dev)
            g_y, d_in = ops.modconv_bwd_prep(g_x, rec["y"], rec["d"], rec["noise"], L.gain, R), None
        x_in = saved["const"] if i == 0 else layers[i - 1]["y"]
        # toRGB gradient of the previous (even) layer joins through the epilogue addend
        addend = None
        if i >= 1 and (i - 1) % 2 == 0:
            kp = (i - 1) // 2
            Op = getattr(mod, f"output{kp}")
            # g_img has already been brought down to this resolution by up2_bwd above
            addend, g_srgb = ops.torgb_bwd(g_img, x_in, Op.weight.detach().reshape(3, -1), saved["rgb"][kp]["s"], Op.wscale)
            if fused:
                blocks.append(dict(gs=g_srgb, wstyle=Op.style.weight.detach(), row=i))
            else:
                ops.linear_t(g_srgb, Op.style.weight.detach(), g_wp[:, i], scale=Op.style.wscale, accumulate=True)
            if kp > 0:
                g_img = ops.up2_bwd(g_img)
        st = ops.SlotStats(B, L.in_c, dev) if fused else ops.zeros((B, L.in_c, 2), dev)
        prep, P_next = None, None
        # (the space-to-depth data gradient of a narrow up layer - layer 15: 64 -> 32 channels - loses more in its 64-wide tile
        #  than the separate pass costs: measured 890 vs 747 us; tools/perf_prep.py)
        t2d = _up_phase_form(L)
        if fused and i >= 1 and (t2d or not (L.up and L.in_c < 128)):
            Lp = getattr(mod, f"layer{i - 1}")
            P_next = ops.SlotStats(B, L.in_c, dev)
            prep = dict(gain=Lp.gain, noise=layers[i - 1]["noise"], ns=Lp.noise_strength.detach().reshape(1), stats=P_next)
        hg = L.res // 2 if L.up else L.res
        if fused and d_in is not None and not t2d and _pp_dgrad(L, B, hg, dt):
            # MFMA-bound launches on the ping-pong kernel (csrc/conv_pp.hip): the demodulation factor is folded into a per-sample
            # weight image instead of scaling g_z in a prologue
            g_xprev = ops.conv_pp(g_y, _dgrad_weight_pp(L, d_in), L.in_c, dgrad=True, in_s2d=L.up, out_scale=rec["s"], addend=addend,
                                  add_scale=1.0, stats=st, dot_src=x_in, prep=prep)
        elif t2d and fused and prep is not None and _pp_dgrad(L, B, hg, dt):
            # phase form on the ping-pong kernel: the same FIR^T pass, then the 4-tap conv with ONE shared weight image
            g_xprev = ops.conv_pp(ops.fir_t2d(g_y, d_in), _dgrad_weight_pp(L, None, t2d=True), L.in_c, dgrad=True, in_t2d=True, out_scale=rec["s"],
                                  addend=addend, add_scale=1.0, stats=st, dot_src=x_in, prep=prep)
        elif t2d:      # phase form: FIR^T (times the demodulation factor) to the t grid, then the 4-tap conv
            g_xprev = ops.conv2d(ops.fir_t2d(g_y, d_in), _dgrad_weight(L, dt), L.in_c, 3, in_t2d=True, out_scale=rec["s"], addend=addend,
                                 add_scale=1.0, stats=st, dot_src=x_in, prep=prep)
        else:
            g_xprev = ops.conv2d(g_y, _dgrad_weight(L, dt), L.in_c, 3, in_s2d=L.up, in_scale=d_in, out_scale=rec["s"], addend=addend,
                                 add_scale=1.0, stats=st, dot_src=x_in, prep=prep)
        # ---- style / demodulation gradients -> g_wp[:, i]
        _, wsq = L._prepared(dt)
        if fused and P is not None:
            blocks.append(dict(P=P, st=st, d=rec["d"], s=rec["s"], bias=L.bias.detach(), wsq=wsq, bscale=L.bscale,
                               wstyle=L.style.weight.detach(), row=i))
            g_x, P = g_xprev, P_next
            continue
        if fused:            # (a layer whose tail backward ran as a pass of its own: its sums sit in R)
            st_t = ops.zeros((B, L.in_c, 2), dev)
            ops.check(ops.lib().dge_sum_slots(ops._p(st.buf), ops._p(st_t), st.nslot, st_t.numel(), 0, ops._stream()), "dge_sum_slots")
            st = st_t
        if P is not None:
            t = ops.demod_bwd_prep(P, rec["d"], L.bias.detach(), L.bscale)
        else:
            t = ops.demod_bwd(R, rec["d"], L.bias.detach(), L.noise_strength.detach().reshape(1), L.bscale)
        gs_view = st.view(B, -1)      # [B, 2*Cin]: element (b, 2*i) = g_s[b,i]
        ops.linear_t(t, wsq, gs_view, mul=rec["s"], accumulate=True, incy=2, ldy=2 * L.in_c)
        ops.linear_t(gs_view, L.style.weight.detach(), g_wp[:, i], scale=L.style.wscale, accumulate=True, incx=2,
                     ldx=2 * L.in_c, O=L.in_c)
        g_x, P = g_xprev, P_next
    if blocks:
        ops.s2_style_grads(blocks, g_wp, getattr(mod, "layer0").style.wscale)
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
