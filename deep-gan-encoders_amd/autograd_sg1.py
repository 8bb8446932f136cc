"""StyleGAN1 synthesis (reference model/stylegan1/net.py Generator.decode :331-336, DecodeBlock.forward
:141-169) as a HIP pipeline with a hand-written data gradient w.r.t. the styles, which is what the
E_align loop differentiates (E_align_s2.py:158: imgs2 = Gs.forward(w2, lod) with w2 from the encoder).
The generator's own parameters receive no gradients (only the encoder is optimised, :97).

Forward per block (see stylegan1.py): conv_1 -> blur+noise+bias+lrelu (y, statistics) ->
[instance norm + style_mod as one per-(b,c) affine in the prologue of] conv_2 +noise+bias+lrelu (x,
statistics) -> affine of the next block's conv_1 / the final toRGB.

Backward per block, from g_u = gradient w.r.t. the affine output u = style_mod(IN(x)):
  dots (sum g_u*x, sum g_u) -> dge_sg1_in_bwd_coef -> style gradient + coefficients,
  dge_in_bwd (instance norm + leaky-relu backward) -> data-gradient conv (conv_2, epilogue dots vs y)
  -> same for style_1 -> blur (self-adjoint) -> data gradient of conv_1 (space-to-depth read for the
  fused transposed conv, or plain + dge_nearest_up2_bwd for upscale2d+conv).
"""
import torch

from . import ops
from .stylegan2_generator import _dt


def decode_run(G, styles, lod, noises=None, save=False):
    dt = _dt(G.compute_dtype)
    dev = styles.device
    B = styles.shape[0]
    styles = styles.float().contiguous()
    ni = 0

    def noise_for(bn, res):
        nonlocal ni
        if noises is not None:
            t = noises[ni].to(dev).float().reshape(-1, res, res).contiguous()
        else:
            t = ops.randn((bn, res, res), dev)
        ni += 1
        return t

    x = ops.nchw_to_nhwc(G.const.detach(), B, dt)
    a = b = None                       # pending instance-norm + style_mod affine of x
    saved = [] if save else None
    for i in range(lod + 1):
        blk = G.decode_block[i]
        Cc = blk.outputs
        res = 4 << i
        st = ops.zeros((B, Cc, 2), dev)
        if blk.has_first_conv:
            if blk.fused_scale:
                t = ops.conv2d(x, blk._packed(blk.conv_1, dt, ops.PACK_SG1_UP), Cc, 3, up=True, in_scale=a, in_shift=b)
            else:
                t = ops.conv2d(x, blk._packed(blk.conv_1, dt, ops.PACK_FWD), Cc, 3, in_scale=a, in_shift=b, in_up2=True)
            y = ops.blur_noise_act(t, noise_for(B, res), blk.noise_weight_1.detach().reshape(-1), blk.bias_1.detach().reshape(-1),
                                   blur=True, stats=st)
        else:
            y = ops.blur_noise_act(x, noise_for(1, res), blk.noise_weight_1.detach().reshape(-1), blk.bias_1.detach().reshape(-1),
                                   blur=False, stats=st)
        _, sc1, sh1 = ops.stats_finalize(st, res * res)
        s1 = ops.linear(styles[:, 2 * i], blk.style_1.weight.detach(), blk.style_1.bias.detach())
        a, b = ops.affine_compose(sc1, sh1, s1)
        st2 = ops.zeros((B, Cc, 2), dev)
        x = ops.conv2d(y, blk._packed(blk.conv_2, dt, ops.PACK_FWD, res), Cc, 3, in_scale=a, in_shift=b,
                       noise=noise_for(B, res), noise_w=blk.noise_weight_2.detach().reshape(-1),
                       bias=blk.bias_2.detach().reshape(-1), act=ops.ACT_LRELU, stats=st2)
        _, sc2, sh2 = ops.stats_finalize(st2, res * res)
        s2 = ops.linear(styles[:, 2 * i + 1], blk.style_2.weight.detach(), blk.style_2.bias.detach())
        a, b = ops.affine_compose(sc2, sh2, s2)
        if save:
            saved.append(dict(y=y, x=x, sc1=sc1, sh1=sh1, s1=s1, sc2=sc2, sh2=sh2, s2=s2))
    xm = ops.blend(x, sc=a, sh=b)      # materialise the last style_mod for the 1x1 toRGB
    rgb = G.to_rgb[lod].to_rgb
    ones = torch.ones((B, x.shape[3]), dtype=torch.float32, device=dev)
    img = ops.torgb(xm, rgb.weight.detach().reshape(3, -1), ones, rgb.bias.detach(), None, 1.0)
    return img, saved


DEBUG_TAP = None     # tests/probes may set this to a dict: {index of the style_mod output: gradient w.r.t. it (NHWC)}


def decode_backward(G, lod, saved, g_image):
    """d(image)/d(styles) contracted with g_image [B,3,R,R] -> g_styles [B, 2*layer_count, latent]."""
    dev = g_image.device
    B = g_image.shape[0]
    dt = ops.dtype_of(saved[0]["x"])
    g_styles = ops.zeros((B, 2 * G.layer_count, G.latent_size), dev)
    rgb = G.to_rgb[lod].to_rgb
    top = saved[lod]
    ones = torch.ones((B, top["x"].shape[3]), dtype=torch.float32, device=dev)
    g_u, _ = ops.torgb_bwd(g_image.float().contiguous(), top["x"], rgb.weight.detach().reshape(3, -1), ones, 1.0)
    dots = ops.dot_stats(g_u, top["x"])
    for i in range(lod, -1, -1):
        if DEBUG_TAP is not None:
            DEBUG_TAP[2 * i + 1] = g_u
        blk = G.decode_block[i]
        rec = saved[i]
        Cc = blk.outputs
        res = 4 << i
        # ---- style_2 / instance norm / lrelu behind conv_2's output
        coef, gs = ops.sg1_in_bwd_coef(dots, rec["sc2"], rec["sh2"], rec["s2"], res * res)
        ops.linear_t(gs, blk.style_2.weight.detach(), g_styles[:, 2 * i + 1], accumulate=True)
        g_pre = ops.in_bwd(g_u, rec["x"], coef, act=True)
        dots = ops.zeros((B, Cc, 2), dev)
        g_u = ops.conv2d(g_pre, blk._packed(blk.conv_2, dt, ops.PACK_DGRAD, res), Cc, 3, stats=dots, dot_src=rec["y"])
        if DEBUG_TAP is not None:
            DEBUG_TAP[2 * i] = g_u
        # ---- style_1 / instance norm / lrelu behind the blurred conv_1 output
        coef, gs = ops.sg1_in_bwd_coef(dots, rec["sc1"], rec["sh1"], rec["s1"], res * res)
        ops.linear_t(gs, blk.style_1.weight.detach(), g_styles[:, 2 * i], accumulate=True)
        if not blk.has_first_conv:
            break                                            # block 0 starts from the constant input
        g_pre = ops.in_bwd(g_u, rec["y"], coef, act=True)
        g_t = ops.blur_noise_act(g_pre, None, None, None, blur=True, act=False)     # Blur is self-adjoint (symmetric, zero pad)
        xprev = saved[i - 1]["x"]
        Cp = blk.inputs
        if blk.fused_scale:
            dots = ops.zeros((B, Cp, 2), dev)
            g_u = ops.conv2d(g_t, blk._packed(blk.conv_1, dt, ops.PACK_SG1_UP_DGRAD), Cp, 3, in_s2d=True, stats=dots, dot_src=xprev)
        else:
            g_hi = ops.conv2d(g_t, blk._packed(blk.conv_1, dt, ops.PACK_DGRAD), Cp, 3)
            g_u, dots = ops.nearest_up2_bwd(g_hi, xprev)
    return g_styles


class DecodeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, G, styles, lod, noises):
        need = ctx.needs_input_grad[1]
        img, saved = decode_run(G, styles.detach(), lod, noises, save=need)
        ctx.G, ctx.lod, ctx.saved_acts = G, lod, saved
        return img

    @staticmethod
    def backward(ctx, g_image):
        if ctx.saved_acts is None:
            raise RuntimeError("StyleGAN1 decode ran without saved activations")
        return None, decode_backward(ctx.G, ctx.lod, ctx.saved_acts, g_image).to(g_image.dtype), None, None
