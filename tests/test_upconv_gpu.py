"""StyleGAN2 up layer at algorithmic cost (dge_upconv_fir: transposed conv in phase form on the MFMAs + FIR from LDS, reference
model/stylegan2_generator.py:879-896,911-921) against (i) a plain torch restatement of the reference lines on the same
inputs and (ii) the folded 3x3-per-phase form of dge_conv2d(up=1), including ragged tile edges (H not a multiple of 14)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference(x, w, s, d, noise, ns, bias, wscale, gain):
    """x [B,Cin,H,W] f32 CPU; the reference's non-fused path: x*s -> conv_transpose2d(flipped kernel, stride 2) -> FIR -> /norm
    -> noise -> bias -> lrelu * sqrt(2)."""
    weight = (w * wscale).permute(2, 3, 1, 0)                 # [k,k,in,out]
    weight = weight.flip(0, 1).permute(2, 3, 0, 1)             # [in,out,k,k]
    t = F.conv_transpose2d(x * s[:, :, None, None], weight, stride=2, padding=0)
    k = torch.tensor([1., 3., 3., 1.]); k2 = torch.outer(k, k); k2 = k2 / k2.sum() * 4.0
    C = t.shape[1]
    y = F.conv2d(F.pad(t, (1, 1, 1, 1)).reshape(-1, 1, t.shape[2] + 2, t.shape[3] + 2), k2[None, None]).reshape(t.shape[0], C, 2 * x.shape[2], 2 * x.shape[3])
    y = y * d[:, :, None, None] + noise[None] * ns + bias[None, :, None, None]
    return F.leaky_relu(y, 0.2) * gain


@pytest.mark.parametrize("cd,B,H,W,Cin,Cout", [("f32", 2, 8, 8, 32, 32), ("f32", 2, 19, 30, 48, 64), ("bf16", 2, 16, 16, 64, 32),
                                                ("bf16", 1, 33, 17, 128, 96)])
def test_upconv_fir_vs_reference_lines_and_folded_form(cd, B, H, W, Cin, Cout):
    from dge_amd import ops
    dt = ops.BF16 if cd == "bf16" else ops.F32
    g = torch.Generator().manual_seed(1234 + H * 7 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g)
    s = 1.0 + 0.3 * torch.randn(B, Cin, generator=g)
    d = 0.5 + torch.rand(B, Cout, generator=g)
    noise = torch.randn(1, 2 * H, 2 * W, generator=g)
    ns = torch.tensor([0.37])
    bias = 0.2 * torch.randn(Cout, generator=g)
    wscale, gain = 1.0 / np.sqrt(9 * Cin), float(np.sqrt(2.0))
    ref = _reference(x, w, s, d, noise, float(ns), bias, wscale, gain)
    assert ops.upconv_supported(Cin, Cout, dt)
    xd = ops.nchw_to_nhwc(x.cuda(), B, dt)
    args = dict(in_scale=s.cuda(), out_scale=d.cuda(), bias=bias.cuda(), bias_scale=1.0, noise=noise.cuda(), noise_w=ns.cuda(),
                act=ops.ACT_LRELU, gain=gain)
    y_new = ops.nhwc_to_nchw(ops.upconv_fir(xd, ops.pack_upconv_weight(w.cuda(), dt, wscale), Cout, **args)).cpu()
    y_old = ops.nhwc_to_nchw(ops.conv2d(xd, ops.pack_conv_weight(w.cuda(), ops.PACK_UPFOLD, dt, wscale), Cout, 3, up=True, **args)).cpu()
    scale = ref.abs().max().item()
    e_new, e_old = (y_new - ref).abs().max().item() / scale, (y_old - ref).abs().max().item() / scale
    tol = 2e-5 if cd == "f32" else 2e-2
    assert e_new < tol, (e_new, e_old)
    assert e_old < tol, (e_new, e_old)


def test_upconv_fir_per_sample_noise_and_linear_activation():
    """randomize_noise=True hands a [B,2H,2W] noise tensor (stylegan2_generator.py:912-913); linear activation / no demodulation
    are the toRGB-style settings of the same block class.  Both against the folded form."""
    from dge_amd import ops
    g = torch.Generator().manual_seed(7)
    B, H, W, Cin, Cout = 3, 16, 24, 64, 64
    x = ops.nchw_to_nhwc(torch.randn(B, Cin, H, W, generator=g).cuda(), B, ops.BF16)
    w = torch.randn(Cout, Cin, 3, 3, generator=g).cuda()
    s = (1.0 + 0.3 * torch.randn(B, Cin, generator=g)).cuda()
    noise = torch.randn(B, 2 * H, 2 * W, generator=g).cuda()
    ns = torch.tensor([0.5]).cuda()
    for kw in (dict(in_scale=s, noise=noise, noise_w=ns, act=ops.ACT_LRELU, gain=1.4142135), dict(in_scale=s, act=ops.ACT_NONE, gain=1.0),
               dict(act=ops.ACT_NONE, gain=1.0)):
        a = ops.upconv_fir(x, ops.pack_upconv_weight(w, ops.BF16, 0.04), Cout, **kw).float()
        b = ops.conv2d(x, ops.pack_conv_weight(w, ops.PACK_UPFOLD, ops.BF16, 0.04), Cout, 3, up=True, **kw).float()
        assert float((a - b).abs().max()) < 2e-2 * float(b.abs().max()), kw.keys()
