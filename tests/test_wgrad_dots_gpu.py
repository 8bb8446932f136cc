"""dge_conv_wgrad_dots: the two sums of a layer's data gradient that the instance-norm backward of its input needs (model/E/E.py:51-53
differentiated), taken from the weight-gradient correlations, against (i) the exact sums of the oracle's data gradient
(oracle/conv_ref.py:conv_dgrad) and (ii) the statistics of the data-gradient launch that produced them before."""
import math

import pytest
import torch

from oracle import conv_ref as CR

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("B,H,W,cin,cout", [(8, 1024, 1024, 16, 32), (8, 512, 512, 32, 64), (4, 256, 256, 64, 64), (3, 100, 72, 32, 48), (2, 64, 80, 128, 128)])
def test_sums_of_the_data_gradient_from_the_weight_gradient(B, H, W, cin, cout):
    from dge_amd import ops
    from dge_amd._lib import last_kernel
    if ops.is_deterministic():
        pytest.skip("dge_conv_wgrad_dots returns 1 in deterministic mode (the data gradient keeps producing the sums)")
    gen = torch.Generator(device=DEV).manual_seed(5000 + H + cin)
    g = torch.randn(B, H, W, cout, device=DEV, generator=gen).to(torch.bfloat16)
    x = (1.5 * torch.randn(B, H, W, cin, device=DEV, generator=gen) + 0.3).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=DEV, generator=gen) / math.sqrt(9 * cin)
    sc = 0.5 + torch.rand(B, cin, device=DEV, generator=gen)
    sh = 0.3 * torch.randn(B, cin, device=DEV, generator=gen)
    dw = ops.zeros((cout, cin, 3, 3), DEV)
    dots = ops.SlotStats(B, cin, DEV)
    assert ops.conv_wgrad_dots(g, x, dw, sc, sh, w, dots)
    assert last_kernel().startswith("wgrad_dma<")
    got = dots.buf.sum(0).cpu().double()
    # the weight gradient itself is unchanged
    dw0 = ops.zeros((cout, cin, 3, 3), DEV)
    ops.conv_wgrad(g, x, dw0, sc, sh)
    assert ((dw - dw0).abs().max() / dw0.abs().max()).item() < 2e-5
    # the launch that produced the sums before: data gradient with dot_src = x
    st = ops.SlotStats(B, cin, DEV)
    ops.conv2d(g, ops.pack_conv_weight(w, ops.PACK_DGRAD, ops.BF16, 1.0), cin, 3, stats=st, dot_src=x)
    old = st.buf.sum(0).cpu().double()
    wq = CR.bf16_round(w.cpu())
    for b in sorted({0, B - 1}):
        gb = g[b:b + 1].float().permute(0, 3, 1, 2).cpu()
        xb = x[b:b + 1].float().permute(0, 3, 1, 2).cpu().double()
        gx = CR.conv_dgrad(gb.double(), wq.double())
        want = torch.stack([(gx * xb).sum((0, 2, 3)), gx.sum((0, 2, 3))], 1)
        absum = torch.stack([(gx * xb).abs().sum((0, 2, 3)), gx.abs().sum((0, 2, 3))], 1)
        e = ((got[b] - want).abs() / absum).max().item()
        assert e < 1e-5, (b, e)
        e_old = ((old[b] - want).abs() / absum).max().item()
        assert e_old < 1e-5, (b, e_old)
