"""TEST INFRASTRUCTURE ONLY (see oracle/__init__ note in DESIGN.md section 5): plain torch-fp32 CPU restatement of the fused
convolution launches of the hot path, one function per call-site family, used by the full-size per-kernel parity tests
(tests/test_fullsize_gpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Each function follows the reference lines it cites on UNFUSED tensors (NCHW f32), i.e. it is the reference's math, not
the kernel's schedule.  Pinning: the per-block goldens of tests/golden/s2_blocks.npz / enc_small.npz (outputs of the
reference's own modules, tools/gen_golden.py) are reproduced by these functions in tests/test_oracle_golden.py.
"""
import torch
import torch.nn.functional as F


def bf16_round(t):
    return t.to(torch.bfloat16).float()


def _ident(t):
    return t


def affine(x, sc, sh):
    """x*sc[b,c] + sh[b,c] with ONE rounding to f32 (computed in f64), i.e. the value of a fused multiply-add: the form in which
    a storage rounding applied afterwards (`q`) is well defined to the last bit."""
    return (x.double() * sc.double()[:, :, None, None] + sh.double()[:, :, None, None]).float()


def modconv(x, w, s, d, noise, noise_strength, bias, bscale, wscale, gain=2.0 ** 0.5, slope=0.2, q=_ident):
    """ModulateConvBlock.forward, stride-1 branch, in the shared-weight form the reference itself uses when
    `fused_modulate` is off (model/stylegan2_generator.py:876-877 x*style, :898-904 conv, :908-909 x*demod,
    :911-921 noise, bias, lrelu*sqrt(2)).  x [B,Cin,H,W], w [Cout,Cin,k,k], s [B,Cin], d [B,Cout] or None,
    noise [1|B,H,W] or None.  `q` is the storage rounding of the modulated activation (identity = the reference's fp32;
    bf16_round = the stated storage precision of the bf16 path, so that a kernel can be checked to the last bit of ITS
    arithmetic instead of through a tolerance that also has to cover the storage format)."""
    xs = x if s is None else q(x * s[:, :, None, None])
    y = F.conv2d(xs, w * wscale, padding=w.shape[-1] // 2)
    if d is not None:
        y = y * d[:, :, None, None]
    if noise is not None:
        y = y + noise[:, None] * noise_strength
    if bias is not None:
        y = y + bias[None, :, None, None] * bscale
    return torch.where(y > 0, y, slope * y) * gain


def modconv_folded(x, w, s, d, noise, noise_strength, bias, bscale, wscale, gain=2.0 ** 0.5, slope=0.2, q=_ident):
    """The same layer in the reference's FUSED-modulation form (:858-864: the style multiplies the weight, the demodulation
    divides it, one weight per sample): W'[b][o][i] = q(w*wscale * (s[b,i] * (gain * d[b,o]))); `q` = the storage rounding of the
    per-sample weight (what csrc/conv_stream.hip holds in registers; the f32 products are associated as the kernel's).  The
    activation gain is folded as well - lrelu(g*t) = g*lrelu(t) for g > 0 - so noise and bias carry it.  Mathematically identical
    to `modconv`; the two differ only in where a storage rounding falls."""
    ys = []
    for b in range(x.shape[0]):
        sb = torch.ones(w.shape[1]) if s is None else s[b]
        db = torch.full((w.shape[0],), float(gain)) if d is None else gain * d[b]
        wb = q((w * wscale) * (sb[None, :] * db[:, None])[:, :, None, None])
        ys.append(F.conv2d(x[b:b + 1], wb, padding=w.shape[-1] // 2))
    y = torch.cat(ys)
    if noise is not None:
        y = y + noise[:, None] * (noise_strength * gain)
    if bias is not None:
        y = y + (bias * bscale * gain)[None, :, None, None]
    return torch.where(y > 0, y, slope * y)


def upconv_fir(x, w, s, d, noise, noise_strength, bias, bscale, wscale, gain=2.0 ** 0.5, slope=0.2, q=_ident):
    """ModulateConvBlock.forward, scale_factor=2 branch (:879-896): x*style -> conv_transpose2d(flipped kernel, stride 2,
    padding 0) -> 4x4 FIR ([1,3,3,1] outer product, gain 4, pad 1; UpsamplingLayer-style filter :603-615) -> demod -> noise
    -> bias -> lrelu*sqrt(2)."""
    weight = (w * wscale).permute(2, 3, 1, 0).flip(0, 1).permute(2, 3, 0, 1)          # [in,out,k,k]
    xs = x if s is None else q(x * s[:, :, None, None])
    t = F.conv_transpose2d(xs, weight, stride=2, padding=0)
    k = torch.tensor([1., 3., 3., 1.])
    k2 = torch.outer(k, k)
    k2 = k2 / k2.sum() * 4.0
    B, C = t.shape[:2]
    y = F.conv2d(F.pad(t, (1, 1, 1, 1)).reshape(B * C, 1, t.shape[2] + 2, t.shape[3] + 2), k2[None, None])
    y = y.reshape(B, C, 2 * x.shape[2], 2 * x.shape[3])
    if d is not None:
        y = y * d[:, :, None, None]
    if noise is not None:
        y = y + noise[:, None] * noise_strength
    if bias is not None:
        y = y + bias[None, :, None, None] * bscale
    return torch.where(y > 0, y, slope * y) * gain


def up_linear(x, w, wscale):
    """the linear part of the up layer (transposed conv + FIR), whose adjoint the backward needs"""
    weight = (w * wscale).permute(2, 3, 1, 0).flip(0, 1).permute(2, 3, 0, 1)
    t = F.conv_transpose2d(x, weight, stride=2, padding=0)
    k = torch.tensor([1., 3., 3., 1.])
    k2 = torch.outer(k, k)
    k2 = k2 / k2.sum() * 4.0
    B, C = t.shape[:2]
    y = F.conv2d(F.pad(t, (1, 1, 1, 1)).reshape(B * C, 1, t.shape[2] + 2, t.shape[3] + 2), k2[None, None])
    return y.reshape(B, C, 2 * x.shape[2], 2 * x.shape[3])


def enc_conv(x, w, sc, sh, noise, noise_w, bias, slope=0.2, q=_ident):
    """BEBlock.forward conv_1 / conv_2 (model/E/E.py:57-62,68-75): instance norm applied as the per-(b,c) affine
    sc*x+sh (zero padding AFTER the norm, as nn.Conv2d pads its normalised input), conv, + noise_weight*noise,
    + bias, leaky_relu(0.2).  Returns (y, per-(b,c) sum, sum of squares of y) - the statistics the next norm reads."""
    xn = q(affine(x, sc, sh))
    y = F.conv2d(xn, w, padding=1)
    y = y + noise_w[None, :, None, None] * noise[:, None] + bias[None, :, None, None]
    y = torch.where(y > 0, y, slope * y)
    yd = y.double()
    return y, yd.sum((2, 3)), (yd * yd).sum((2, 3))


def enc_conv_folded(x, w, sc, sh, noise, noise_w, bias, slope=0.2, q=_ident):
    """enc_conv with the instance-norm scale folded into the weight, W'[b] = q(w * sc[b,i]), applied to x + sh/sc (zero padding
    after the shift): W'.(x + sh/sc) == w.(sc*x + sh) exactly when q is the identity."""
    ys = []
    for b in range(x.shape[0]):
        wb = q(w * sc[b][None, :, None, None])
        xs = x[b:b + 1] + (sh[b] / sc[b])[None, :, None, None]
        ys.append(F.conv2d(xs, wb, padding=1))
    y = torch.cat(ys)
    y = y + noise_w[None, :, None, None] * noise[:, None] + bias[None, :, None, None]
    y = torch.where(y > 0, y, slope * y)
    yd = y.double()
    return y, yd.sum((2, 3)), (yd * yd).sum((2, 3))


def enc_skip_conv(xp, w3, b3, x2, gain=0.889, add_scale=0.111):
    """BEBlock.forward residual join (E.py:77-83): out = 0.111*x2 + 0.889*(conv_3(pool(x)) + bias)."""
    y = (F.conv2d(xp, w3) + b3[None, :, None, None]) * gain + add_scale * x2
    yd = y.double()
    return y, yd.sum((2, 3)), (yd * yd).sum((2, 3))


def conv_dgrad(g, w, wscale=1.0):
    """adjoint of y = conv2d(x, w*wscale, padding=k//2) w.r.t. x"""
    return F.conv_transpose2d(g, w * wscale, padding=w.shape[-1] // 2)


def up_dgrad(g, w, wscale, hin):
    """adjoint of up_linear w.r.t. x (x [B,Cin,hin,hin])"""
    x = torch.zeros(g.shape[0], w.shape[1], hin, hin, requires_grad=True)
    y = up_linear(x, w, wscale)
    return torch.autograd.grad(y, x, g)[0]


def upfold_weights(w, wscale=1.0):
    """The up layer's linear part (`up_linear`: transposed conv with the flipped kernel, stride 2, then the 4x4 FIR,
    stylegan2_generator.py:879-896) as FOUR 3x3 stride-1 convolutions, one per output phase: y[2m+py, 2n+px] = conv3x3(x, Wf[2py+px])[m, n].
    Wf [4,O,I,3,3]; tap (a, c) reads x[m+a-1, n+c-1] and collects k1[ty] k1[tx] w[3-py-ty+2(a-1), 3-px-tx+2(c-1)], k1 = (1,3,3,1)/4.
    (The folded form dge_pack_conv_weight builds; checked against up_linear in tests/test_oracle_golden.py.)"""
    O, I = w.shape[:2]
    k1 = torch.tensor([0.25, 0.75, 0.75, 0.25], dtype=w.dtype)
    Wf = torch.zeros(4, O, I, 3, 3, dtype=w.dtype)
    for py in range(2):
        for px in range(2):
            for a in range(3):
                for c in range(3):
                    for ty in range(4):
                        wy = 3 - py - ty + 2 * (a - 1)
                        if not 0 <= wy <= 2:
                            continue
                        for tx in range(4):
                            wx = 3 - px - tx + 2 * (c - 1)
                            if 0 <= wx <= 2:
                                Wf[2 * py + px, :, :, a, c] += k1[ty] * k1[tx] * w[:, :, wy, wx]
    return Wf * wscale


def up_folded(x, Wf):
    """the four phase convolutions of `upfold_weights`, interleaved: [B,I,H,W] -> [B,O,2H,2W]"""
    B, _, H, W = x.shape
    y = torch.zeros(B, Wf.shape[1], 2 * H, 2 * W, dtype=x.dtype)
    for ph in range(4):
        y[:, :, (ph >> 1)::2, (ph & 1)::2] = F.conv2d(x, Wf[ph], padding=1)
    return y


def dgrad_folded(g, w, wscale, d, up=False, q=_ident):
    """Raw data gradient (before the style factor of the layer's input) of a modulated conv whose demodulation factor d [O] is
    folded into the weight the gradient conv reads, W' = q(w*wscale*d[o]) (stride 1) / q(Wf*d[o]) (up layer in folded form): the
    per-sample data-gradient weight image of csrc/conv_pp.hip.  g [1,O,Hg,Wg] -> [1,I,H,W], computed in f64."""
    gd = g.double()
    if not up:
        return F.conv_transpose2d(gd, q(w * wscale * d[:, None, None, None]).double(), padding=1).float()
    Wq = q(upfold_weights(w, wscale) * d[None, :, None, None, None]).double()
    x = torch.zeros(1, w.shape[1], g.shape[2] // 2, g.shape[3] // 2, dtype=torch.float64, requires_grad=True)
    return torch.autograd.grad(up_folded(x, Wq), x, gd)[0].float()


def conv_wgrad(g, xn, k):
    """dW[o,i,ky,kx] = sum_{b,y,x} g[b,o,y,x] * xn[b,i,y+ky-p,x+kx-p] (zero padding): k*k plain matrix products"""
    p = k // 2
    B, Ci, H, W = xn.shape
    Co = g.shape[1]
    xp = F.pad(xn, (p, p, p, p))
    gm = g.permute(1, 0, 2, 3).reshape(Co, -1)
    dw = torch.empty(Co, Ci, k, k)
    for ky in range(k):
        for kx in range(k):
            xm = xp[:, :, ky:ky + H, kx:kx + W].permute(1, 0, 2, 3).reshape(Ci, -1)
            dw[:, :, ky, kx] = gm @ xm.t()
    return dw
