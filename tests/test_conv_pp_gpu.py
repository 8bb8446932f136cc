"""Parity of the ping-pong implicit GEMM (csrc/conv_pp.hip) at the shapes the benchmark runs it on, asserted by kernel name, plus
ragged shapes (partial tiles in both directions, odd chunk counts) and the shared-weight (LPIPS) form.  Bounds as in
tests/test_fullsize_gpu.py: every element within ONE bf16 rounding of the exact-arithmetic oracle that places the storage
rounding where the kernel does (oracle/conv_ref.py:modconv_folded - the reference's fused modulation :858-875), and a storage
bound against the plain f32 oracle."""
import math

import pytest
import torch

from oracle import conv_ref as CR

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _nchw(x_nhwc, b):
    return x_nhwc[b:b + 1].float().permute(0, 3, 1, 2).contiguous().cpu()


def _one_rounding(got, ref, slack=1e-5):
    return ((got - ref).abs() - (2.0 ** -8) * ref.abs() - slack * ref.abs().max()).max().item()


def _case(B, H, W, cin, cout, seed, modulated=True, samples=None):
    from dge_amd import ops
    from dge_amd._lib import last_kernel
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(B, H, W, cin, device=DEV, generator=g).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=DEV, generator=g).to(torch.bfloat16).float()
    wscale = 1.0 / math.sqrt(9 * cin)
    noise = torch.randn(1, H, W, device=DEV, generator=g)
    ns = torch.tensor([0.37], device=DEV)
    bias = 0.2 * torch.randn(cout, device=DEV, generator=g)
    gain = math.sqrt(2.0)
    assert ops.conv_pp_supported(B, H, W, cin, cout, ops.BF16)
    if modulated:
        s = 1.0 + 0.3 * torch.randn(B, cin, device=DEV, generator=g)
        d = 0.5 + torch.rand(B, cout, device=DEV, generator=g)
        wpp = ops.pack_conv_pp(w, wscale, in_scale=s, out_scale=d, gain=gain)
        y = ops.conv_pp(x, wpp, cout, bias=bias, bias_scale=1.0, noise=noise, noise_w=ns, act=ops.ACT_LRELU, gain=gain)
    else:
        s = d = None
        wpp = ops.pack_conv_pp(w, wscale)
        y = ops.conv_pp(x, wpp, cout, bias=bias, bias_scale=1.0, act=ops.ACT_RELU, gain=1.0)
    assert last_kernel() == "conv_pp<bf16,16,32,128>"
    for b in (samples if samples is not None else sorted({0, B - 1})):
        if modulated:
            a = (_nchw(x, b), w.cpu(), s[b:b + 1].cpu(), d[b:b + 1].cpu(), noise.cpu(), 0.37, bias.cpu(), 1.0, wscale)
            ref_q, ref = CR.modconv_folded(*a, q=CR.bf16_round), CR.modconv(*a)
        else:
            a = (_nchw(x, b), w.cpu(), None, None, None, 0.0, bias.cpu(), 1.0, wscale)
            ref_q = CR.modconv_folded(*a, gain=1.0, slope=0.0, q=CR.bf16_round)
            ref = CR.modconv(*a, gain=1.0, slope=0.0)
        got = _nchw(y, b)
        assert _one_rounding(got, ref_q) <= 0, (b, _one_rounding(got, ref_q))
        e = ((got - ref).abs().max() / ref.abs().max()).item()
        assert e < 8e-3, (b, e)


@pytest.mark.parametrize("cin,cout,R,B", [(128, 128, 256, 8), (256, 256, 128, 8), (512, 512, 64, 8)])
def test_generator_layers_fullsize(cin, cout, R, B):
    """layers 12 / 10 / 8 of the StyleGAN2-1024 synthesis at batch 8 (stylegan2_generator.py:855-922, stride-1 branch)"""
    _case(B, R, R, cin, cout, 4000 + cin)


@pytest.mark.parametrize("B,H,W,cin,cout", [(40, 50, 70, 64, 128), (12, 33, 97, 96, 256), (48, 44, 44, 128, 128), (16, 129, 63, 160, 128)])
def test_ragged_shapes(B, H, W, cin, cout):
    """partial tiles in y and x, 2 / 3 / 4 / 5 K chunks, every sample checked"""
    _case(B, H, W, cin, cout, 4100 + H, samples=range(0, B, max(1, B // 6)))


@pytest.mark.parametrize("B,H,W,cin,cout", [(16, 128, 128, 64, 128), (16, 64, 48, 256, 256)])
def test_shared_weights_bias_relu(B, H, W, cin, cout):
    """one shared weight image, bias + ReLU epilogue: the LPIPS VGG16 convs (third-party lpips algorithm, call site training_utils.py:93)"""
    _case(B, H, W, cin, cout, 4200 + H, modulated=False)
