"""TEST INFRASTRUCTURE ONLY: plain torch CPU restatement of the element-wise stages AROUND the convolutions of the hot path
(noise / bias / activation tails, toRGB, instance norm with its (mean, std) outputs, pooling / residual blend, FromRGB) and
of their gradients, for the full-size per-kernel parity tests (tests/test_fullsize_elem_gpu.py).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Every forward follows the reference lines it cites on unfused NCHW tensors; every backward is torch.autograd of that
forward - what the reference itself runs - never a hand-derived formula.  Functions take ONE sample ([1,C,H,W]) so that
full-size tensors (1024^2) stay cheap on the host.  Pinning: the forwards are the same lines oracle/ref_torch.py restates
and pins on the reference's block outputs (tests/test_oracle_golden.py); `tests/test_oracle_golden.py::test_elem_ref_*`
checks these functions against ref_torch on the golden inputs."""
import torch
import torch.nn.functional as F

from . import ref_torch as O


def _leaf(t):
    return t.detach().clone().double().requires_grad_(True)


def lrelu_inverse(x, gain, slope=0.2):
    """pre-activation z of a stored activation x = lrelu(z)*gain (the HIP path keeps only x)"""
    return torch.where(x > 0, x / gain, x / (slope * gain))


# ----------------------------------------------------------------------------- StyleGAN2 modulated-conv tail
def modconv_tail(yraw, d, noise, ns, bias, bscale, gain, slope=0.2):
    """model/stylegan2_generator.py:908-921: x = lrelu(yraw*d + noise*noise_strength + bias*bscale) * gain.
    yraw [1,C,H,W], d [C], noise [H,W], bias [C]."""
    z = yraw * d[None, :, None, None] + noise[None, None] * ns + bias[None, :, None, None] * bscale
    return F.leaky_relu(z, slope) * gain


def modconv_tail_bwd(x, gx, d, noise, gain, slope=0.2):
    """Gradient of `modconv_tail` given the STORED activation x and the upstream gradient gx: returns
    (g_yraw, R [C,3]) with R = (sum g_z*z, sum g_z*noise, sum g_z) - the three sums the demodulation / noise-strength / bias
    gradients are built from (d(d)/d(style) needs sum g_z*yraw*d = sum g_z*(z - noise*ns - bias*bscale))."""
    z = _leaf(lrelu_inverse(x.double(), gain, slope))
    out = F.leaky_relu(z, slope) * gain
    (gz,) = torch.autograd.grad(out, z, gx.double())
    R = torch.stack([(gz * z.detach()).sum(dim=(0, 2, 3)), (gz * noise.double()[None, None]).sum(dim=(0, 2, 3)), gz.sum(dim=(0, 2, 3))], dim=1)
    return gz * d.double()[None, :, None, None], R


# ----------------------------------------------------------------------------- toRGB + skip
def torgb(x, wrgb, style, bias, wscale, prev=None):
    """SynthesisModule.forward :515-522 with ModulateConvBlock k=1, demodulate=False, linear (:465-474): image = conv1x1(x*style,
    W*wscale) + bias (+ upsample(prev), UpsamplingLayer :603-615).  x [1,C,H,W], wrgb [3,C], style [C], prev [1,3,H/2,W/2]."""
    img = F.conv2d(x * style[None, :, None, None], (wrgb * wscale)[:, :, None, None]) + bias[None, :, None, None]
    if prev is not None:
        img = img + O.s2_upsample_skip(prev.float()).to(img.dtype)
    return img


def torgb_bwd(x, wrgb, style, wscale, gimg):
    """autograd of `torgb` w.r.t. x and style: (gx [1,C,H,W], gs [C])"""
    xl, sl = _leaf(x), _leaf(style)
    img = torgb(xl, wrgb.double(), sl, torch.zeros(3, dtype=torch.float64), wscale)
    gx, gs = torch.autograd.grad(img, (xl, sl), gimg.double())
    return gx, gs


# ----------------------------------------------------------------------------- encoder: activation + pool tail (E.py:73-78,84)
def enc_act_pool_bwd(a, gout, noise, scale, slope=0.2):
    """a = lrelu(pre + nw*noise + b) (E.py:73-74; stored), out = scale4 * avg_pool2d(a) (E.py:75 downscale + the 0.111 blend
    weight :84; `scale` = weight * 0.25 as the HIP call passes it).  Returns (g_pre, g_bias [C], g_noise_weight [C])."""
    pre = _leaf(torch.where(a > 0, a, a / slope))
    nw = torch.zeros(a.shape[1], dtype=torch.float64, requires_grad=True)
    b = torch.zeros(a.shape[1], dtype=torch.float64, requires_grad=True)
    act = F.leaky_relu(pre + nw[None, :, None, None] * noise.double()[None, None] + b[None, :, None, None], slope)
    out = (scale * 4.0) * F.avg_pool2d(act, 2, 2)
    gp, gb, gn = torch.autograd.grad(out, (pre, b, nw), gout.double())
    return gp, gb, gn


# ----------------------------------------------------------------------------- encoder: instance norm with (mean, std) heads
def enc_in_bwd(X, gy, gmu, gsg, extra=None, extra_scale=1.0, noise=None, act=False, slope=0.2, eps=1e-8):
    """E.py:51-57 (and :64-68): m = mean(X), s = sqrt(mean((X-m)^2)) feed inver_mod (gradients gmu, gsg [C]); y = InstanceNorm2d
    (eps 1e-8, biased variance) feeds the conv (gradient gy, may be None).  `extra` [1,C,H/2,W/2]: a second consumer of X through
    avg_pool2d (the residual branch, E.py:78), its gradient enters as extra_scale*4 * avg_pool adjoint.  With act, X = lrelu(pre +
    nw*noise + b) (:60-62) and the gradient continues to pre, b, nw.  Returns (g, g_bias, g_nw) (the last two None without act)."""
    if act:
        pre = _leaf(torch.where(X > 0, X, X / slope))
        nw = torch.zeros(X.shape[1], dtype=torch.float64, requires_grad=True)
        b = torch.zeros(X.shape[1], dtype=torch.float64, requires_grad=True)
        Xl = F.leaky_relu(pre + nw[None, :, None, None] * noise.double()[None, None] + b[None, :, None, None], slope)
        leaves = (pre, b, nw)
    else:
        Xl = _leaf(X)
        leaves = (Xl,)
    m = Xl.mean(dim=(2, 3))
    v = ((Xl - m[:, :, None, None]) ** 2).mean(dim=(2, 3))
    y = (Xl - m[:, :, None, None]) * torch.rsqrt(v + eps)[:, :, None, None]
    tot = (m * gmu.double()[None]).sum() + (v.sqrt() * gsg.double()[None]).sum()
    if gy is not None:
        tot = tot + (y * gy.double()).sum()
    if extra is not None:
        tot = tot + (F.avg_pool2d(Xl, 2, 2) * extra.double()).sum() * (extra_scale * 4.0)
    g = torch.autograd.grad(tot, leaves)
    return (g[0], g[1], g[2]) if act else (g[0], None, None)


# ----------------------------------------------------------------------------- encoder: pool / blend / IN-apply (E.py:75-78,84)
def blend(x, z=None, sc=None, sh=None, pool=False, alpha=1.0, beta=0.0):
    """y = alpha * P(x*sc + sh) + beta * z, P = avg_pool2d(2) or identity"""
    t = x.double()
    if sc is not None:
        t = t * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    if pool:
        t = F.avg_pool2d(t, 2, 2)
    t = alpha * t
    if z is not None:
        t = t + beta * z.double()
    return t


# ----------------------------------------------------------------------------- FromRGB (model/utils/net.py:231-240)
def fromrgb_bwd(x0, gx, img, slope=0.2):
    """x0 = lrelu(conv1x1(img, W) + b) stored; returns (gW [C,3], gb [C]) by autograd (W's value does not enter its gradient)."""
    C = x0.shape[1]
    W = torch.zeros(C, 3, 1, 1, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(C, dtype=torch.float64, requires_grad=True)
    pre0 = torch.where(x0 > 0, x0, x0 / slope).double()
    out = F.leaky_relu(pre0 + F.conv2d(img.double(), W, b), slope)
    gW, gb = torch.autograd.grad(out, (W, b), gx.double())
    return gW.reshape(C, 3), gb
