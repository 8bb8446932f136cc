"""E_PG.BE (reference model/E/E_PG.py:73-108,150-164) forward / hand-written backward pipelines over the HIP ops.

Per block: IN -> conv3x3 -> +noise -> +bias -> lrelu -> IN -> conv3x3 -> +noise -> +bias -> (+ affine-IN(conv1x1(residual)))
-> lrelu -> avg_pool2d; the last block stops after the first lrelu; head `new_final` on the NCHW-flattened activation
(the evident intent).  The gradient enters through the head output only.

Backward building blocks (all in libdge_hip.so): `act_bwd` (lrelu' + pooling adjoint + bias / noise-weight sums),
`conv_wgrad`, data-gradient convs whose epilogue yields the two per-(b,c) sums the instance-norm backward needs,
`in_bwd_coef` / `in_bwd` (instance-norm backward as one streaming affine pass) and, for the affine instance norm of
the residual branch, `sg1_in_bwd_coef` with (gamma - 1, beta) in the role of StyleGAN1's per-sample style: its style
gradients summed over the batch are the gamma / beta gradients.
"""
import torch

from . import ops
from .autograd_enc import _packed, draw_noises
from .stylegan2_generator import _dt


def pg_encoder_forward(E, img, noises=None, save=False):
    dt = _dt(E.compute_dtype)
    dev = img.device
    B, _, R, _ = img.shape
    if noises is None:
        noises = draw_noises(E, B, R, dev)
    cache = E.__dict__.setdefault("_pack_cache", {})
    zeros = lambda c: ops.zeros((B, c, 2), dev)
    fr = E.FromRGB.from_rgb
    stats = zeros(E.startf)
    x = ops.fromrgb(img.float(), fr.weight.detach(), fr.bias.detach(), dt, stats)
    saved = {"img": img, "x0": x, "blocks": []} if save else None
    ni = 0
    for j, blk in enumerate(E.decode_block):
        Cc, C2, H = blk.inputs, blk.outputs, R >> j
        musig1, sc1, sh1 = ops.stats_finalize(stats, H * H)
        st1 = zeros(Cc)
        n1 = noises[ni].reshape(B, H, H).contiguous(); ni += 1
        x1 = ops.conv2d(x, _packed(cache, blk.conv_1, dt, ops.PACK_FWD, H), Cc, 3, in_scale=sc1, in_shift=sh1, noise=n1,
                        noise_w=blk.noise_weight_1.detach().reshape(-1), bias=blk.bias_1.detach().reshape(-1),
                        act=ops.ACT_LRELU, stats=st1)
        rec = dict(x=x, musig1=musig1, sc1=sc1, sh1=sh1, n1=n1, x1=x1) if save else None
        if not blk.has_second_conv:
            if save:
                saved["blocks"].append(rec)
            x = x1
            break
        musig2, sc2, sh2 = ops.stats_finalize(st1, H * H)
        n2 = noises[ni].reshape(B, H, H).contiguous(); ni += 1
        pre2 = ops.conv2d(x1, _packed(cache, blk.conv_2, dt, ops.PACK_FWD, H), C2, 3, in_scale=sc2, in_shift=sh2, noise=n2,
                          noise_w=blk.noise_weight_2.detach().reshape(-1), bias=blk.bias_2.detach().reshape(-1))
        if Cc != C2:
            st3 = zeros(C2)
            r3 = ops.conv2d(x, _packed(cache, blk.conv_3, dt, ops.PACK_FWD), C2, 1, bias=blk.conv_3.bias.detach(), stats=st3)
            _, sc3, sh3 = ops.stats_finalize(st3, H * H)
            g, bta = blk.instance_norm_3.weight.detach(), blk.instance_norm_3.bias.detach()
            s = ops.blend(r3, z=pre2, sc=(sc3 * g).contiguous(), sh=(sh3 * g + bta).contiguous(), alpha=1.0, beta=1.0)
        else:
            r3 = sc3 = sh3 = None
            s = ops.blend(x, z=pre2, alpha=1.0, beta=1.0)
        a = ops.blur_noise_act(s, None, None, None, blur=False)          # leaky_relu(x + residual)
        nstats = zeros(C2)
        xn = ops.blend(a, pool=True, stats=nstats)                       # avg_pool2d
        if save:
            rec.update(musig2=musig2, sc2=sc2, sh2=sh2, n2=n2, r3=r3, sc3=sc3, sh3=sh3, a=a)
            saved["blocks"].append(rec)
        x, stats = xn, nstats
    xo = ops.nhwc_to_nchw(x)
    z = None
    if E.pggan:
        flat = xo.reshape(B, -1)
        z = ops.linear(flat, E.new_final.weight.detach(), E.new_final.bias.detach())
        if save:
            saved["flat"] = flat
    return xo, z, saved


def pg_encoder_backward(E, saved, g_z):
    """Gradients for E.parameters() in registration order."""
    if saved is None:
        raise RuntimeError("E_PG forward ran without saved activations")
    cache = E.__dict__.setdefault("_pack_cache", {})
    dev = g_z.device
    B = g_z.shape[0]
    R = saved["img"].shape[2]
    dt = ops.dtype_of(saved["x0"])
    grads = {}
    # head: z = flat @ W^T + b
    W = E.new_final.weight.detach()
    flat = saved["flat"]
    g_z = g_z.float().contiguous()
    g_flat = torch.empty_like(flat)
    ops.linear_t(g_z, W, g_flat)
    gw, gb = torch.empty_like(W), torch.empty_like(E.new_final.bias)
    ops.dense_wgrad(g_z, flat, gw, gb)
    grads["new_final.weight"], grads["new_final.bias"] = gw, gb
    L = len(saved["blocks"])
    C_last = E.decode_block[L - 1].inputs
    g_out = ops.nchw_to_nhwc(g_flat.view(B, C_last, R >> (L - 1), R >> (L - 1)), B, dt)
    for j in range(L - 1, -1, -1):
        blk, rec = E.decode_block[j], saved["blocks"][j]
        pre = f"decode_block.{j}."
        Cc, C2 = blk.inputs, blk.outputs
        H = R >> j
        N = H * H
        x, x1 = rec["x"], rec["x1"]
        red1 = ops.zeros((Cc, 2), dev)
        if blk.has_second_conv:
            has3 = Cc != C2
            # s = pre2 + res ; a = lrelu(s) ; out = avg_pool(a)
            red2 = ops.zeros((C2, 2), dev)
            g_s = ops.act_bwd(g_out, rec["a"], rec["n2"], pool=True, scale=0.25, red=red2)
            grads[pre + "bias_2"] = red2[:, 0].reshape(1, C2, 1, 1)
            grads[pre + "noise_weight_2"] = red2[:, 1].reshape(1, C2, 1, 1)
            gW2 = ops.zeros(tuple(blk.conv_2.weight.shape), dev)
            ops.conv_wgrad(g_s, x1, gW2, rec["sc2"], rec["sh2"])
            grads[pre + "conv_2.weight"] = gW2
            dots2 = ops.zeros((B, Cc, 2), dev)
            g_y2 = ops.conv2d(g_s, _packed(cache, blk.conv_2, dt, ops.PACK_DGRAD, H), Cc, 3, stats=dots2, dot_src=x1)
            coef2 = ops.in_bwd_coef(dots2, None, rec["musig2"], rec["sc2"], rec["sh2"], N)
            g_pre1 = ops.in_bwd(g_y2, x1, coef2, noise=rec["n1"], act=True, red=red1)
            if has3:
                gam, bta = blk.instance_norm_3.weight.detach(), blk.instance_norm_3.bias.detach()
                style = torch.cat([gam - 1.0, bta]).unsqueeze(0).expand(B, -1).contiguous()
                coef3, gstyle = ops.sg1_in_bwd_coef(ops.dot_stats(g_s, rec["r3"]), rec["sc3"], rec["sh3"], style, N)
                gsum = ops._sum_over_batch(gstyle)
                grads[pre + "instance_norm_3.weight"], grads[pre + "instance_norm_3.bias"] = gsum[:C2], gsum[C2:]
                g_r3 = ops.in_bwd(g_s, rec["r3"], coef3)
                grads[pre + "conv_3.bias"] = ops.chan_sum(g_r3)
                gW3 = ops.zeros(tuple(blk.conv_3.weight.shape), dev)
                ops.conv_wgrad(g_r3, x, gW3)
                grads[pre + "conv_3.weight"] = gW3
                extra = ops.conv2d(g_r3, _packed(cache, blk.conv_3, dt, ops.PACK_DGRAD), Cc, 1)
            else:
                extra = g_s
        else:
            g_pre1 = ops.act_bwd(g_out, x1, rec["n1"], pool=False, scale=1.0, red=red1)
            extra = None
        grads[pre + "bias_1"] = red1[:, 0].reshape(1, Cc, 1, 1)
        grads[pre + "noise_weight_1"] = red1[:, 1].reshape(1, Cc, 1, 1)
        gW1 = ops.zeros(tuple(blk.conv_1.weight.shape), dev)
        ops.conv_wgrad(g_pre1, x, gW1, rec["sc1"], rec["sh1"])
        grads[pre + "conv_1.weight"] = gW1
        dots1 = ops.zeros((B, Cc, 2), dev)
        g_y1 = ops.conv2d(g_pre1, _packed(cache, blk.conv_1, dt, ops.PACK_DGRAD, H), Cc, 3, stats=dots1, dot_src=x)
        coef1 = ops.in_bwd_coef(dots1, None, rec["musig1"], rec["sc1"], rec["sh1"], N)
        g_out = ops.in_bwd(g_y1, x, coef1, extra=extra, extra_pool=False, extra_scale=1.0)
    fr = ops.fromrgb_bwd(g_out, saved["x0"], saved["img"].float())
    C0 = E.startf
    grads["FromRGB.from_rgb.weight"] = fr[:, :3].reshape(C0, 3, 1, 1)
    grads["FromRGB.from_rgb.bias"] = fr[:, 3]
    out = []
    for name, _ in E.named_parameters():
        g = grads.get(name)
        out.append(g.contiguous() if g is not None else None)
    return out


class PGEncoderFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E, img, noises, *params):
        if ctx.needs_input_grad[1]:
            raise ops.DgeError("E_PG: the gradient w.r.t. the input image is not implemented on the HIP path; detach the image")
        need = any(ctx.needs_input_grad[3:])
        _, z, saved = pg_encoder_forward(E, img.detach(), noises, save=need)
        ctx.E, ctx.saved_acts = E, saved
        return z

    @staticmethod
    def backward(ctx, g_z):
        return (None, None, None) + tuple(pg_encoder_backward(ctx.E, ctx.saved_acts, g_z))
