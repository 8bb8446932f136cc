"""Conv weight-gradient kernel (dge_conv_wgrad) against a torch fp32 CPU restatement of the same
contraction: dW[o,i,ky,kx] = sum_{b,y,x} g[b,o,y,x] * Xn[b,i,y+ky-1,x+kx-1] with
Xn = X*sc[b,i] + sh[b,i] inside the image and zero padding outside (the instance-norm affine that
the encoder forward fuses into the conv prologue, reference model/E/E.py:50-85)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(g, x, sc, sh, k):
    xn = x if sc is None else x * sc[:, :, None, None] + sh[:, :, None, None]
    w = torch.zeros(g.shape[1], x.shape[1], k, k, requires_grad=True)
    y = torch.nn.functional.conv2d(xn, w, padding=k // 2)
    y.backward(g)
    return w.grad


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("B,H,W,cin,cout,k,affine", [
    (2, 24, 20, 16, 24, 3, True),        # ragged tiles, partial channel tiles
    (1, 16, 32, 40, 64, 3, False),       # Cin not a multiple of 32
    (3, 8, 8, 32, 32, 3, True),
    (2, 33, 17, 64, 8, 3, True),         # odd sizes
    (2, 16, 16, 32, 16, 1, False),       # 1x1 (FromRGB-style)
    (1, 4, 4, 64, 64, 3, True),          # single partial tile
])
def test_conv_wgrad_vs_torch(dtype, B, H, W, cin, cout, k, affine):
    from dge_amd import ops
    gen = torch.Generator().manual_seed(B * 1000 + H * 10 + cin)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    g = torch.randn(B, cout, H, W, generator=gen).to(tdt).float()
    x = torch.randn(B, cin, H, W, generator=gen).to(tdt).float()
    sc = (torch.rand(B, cin, generator=gen) + 0.5) if affine else None
    sh = torch.randn(B, cin, generator=gen) if affine else None
    want = _ref(g, x, sc, sh, k)
    gd = g.permute(0, 2, 3, 1).contiguous().to(tdt).cuda()
    xd = x.permute(0, 2, 3, 1).contiguous().to(tdt).cuda()
    dw = torch.zeros(cout, cin, k, k, device="cuda")
    ops.conv_wgrad(gd, xd, dw, None if sc is None else sc.cuda(), None if sh is None else sh.cuda())
    got = dw.cpu()
    # bf16: the affine result is rounded to bf16 before the MFMA (as in the forward conv): 2^-8 relative per term
    tol = 2e-2 if (dtype == "bf16" and affine) else 2e-3
    err = (got - want).norm() / want.norm()
    assert err < tol, f"relative L2 error {err:.3e}"
    assert torch.isfinite(got).all()


def test_conv_wgrad_accumulates():
    from dge_amd import ops
    g = torch.randn(1, 8, 8, 32, device="cuda").bfloat16()
    x = torch.randn(1, 8, 8, 32, device="cuda").bfloat16()
    dw = torch.zeros(32, 32, 3, 3, device="cuda")
    ops.conv_wgrad(g, x, dw)
    once = dw.clone()
    ops.conv_wgrad(g, x, dw)
    assert torch.allclose(dw, 2 * once, rtol=1e-5, atol=1e-5)
