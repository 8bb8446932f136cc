"""GPU parity of the loss kernels against the reference's golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from tests.conftest import golden
from tests.golden import recipe as R
from oracle import ref_torch as O

pytestmark = pytest.mark.gpu


def test_space_loss_vs_reference_golden():
    from dge_amd import losses
    g = golden("loss.npz")
    big_a = R.randn("loss.big_a", (2, 3, 512, 384), 2, 0.5).cuda()
    big_b = (R.randn("loss.big_a", (2, 3, 512, 384), 2, 0.5) + R.randn("loss.big_b", (2, 3, 512, 384), 2, 0.2)).cuda().requires_grad_(True)

    class StandIn:       # same stand-in LPIPS as tools/gen_golden.py: per-sample mean squared difference
        def value_and_grad(self, ap, bp, need_grad):
            d = bp - ap
            val = (d * d).mean().reshape(1)
            return val, (2 * d / d.numel() if need_grad else None)

    loss, info = losses.space_loss(big_a, big_b, lpips_model=StandIn())
    loss.backward()
    ref = g["img_info"]
    got = info.cpu().numpy()
    assert abs(float(loss) - float(g["img_loss"])) < 2e-4 * abs(float(g["img_loss"]))
    for i in range(7):
        assert abs(got[1 + i] - ref[i]) <= 5e-4 * abs(ref[i]) + 2e-6, (i, got[1 + i], ref[i])
    crop = big_b.grad[:, :, 100:116, 200:216].cpu().numpy()
    assert np.abs(crop - g["img_grad_b_crop"]).max() < 2e-3 * np.abs(g["img_grad_b_crop"]).max()
    gs = g["img_grad_b_sum"]
    assert abs(big_b.grad.double().abs().sum().item() - gs[1]) < 1e-3 * gs[1]

    w1 = R.randn("loss.w1", (2, 18, 512), 2).cuda()
    w2 = (R.randn("loss.w1", (2, 18, 512), 2) * 0.9 + R.randn("loss.w2", (2, 18, 512), 2, 0.3)).cuda().requires_grad_(True)
    loss, info = losses.space_loss(w1, w2, image_space=False)
    loss.backward()
    assert abs(float(loss) - float(g["w_loss"])) < 2e-4 * abs(float(g["w_loss"]))
    ref = g["w_info"]; got = info.cpu().numpy()
    for i in range(5):
        assert abs(got[1 + i] - ref[i]) <= 5e-4 * abs(ref[i]) + 2e-6, (i, got[1 + i], ref[i])
    assert np.abs(w2.grad.cpu().numpy() - g["w_grad"]).max() < 1e-3 * np.abs(g["w_grad"]).max()


def test_ssim_known_answers():
    from dge_amd import losses
    g = golden("loss.npz")
    a = R.randn("loss.a", (2, 3, 64, 64), 1, 0.5).clamp(-1, 1).cuda()
    b = (R.randn("loss.a", (2, 3, 64, 64), 1, 0.5).clamp(-1, 1) + R.randn("loss.b", (2, 3, 64, 64), 1, 0.1)).clamp(-1, 1).cuda()
    _, info = losses.space_loss(a, b)
    assert abs((1 - float(info[6])) - float(g["ssim_64"])) < 1e-5
    _, info = losses.space_loss(a, a)
    assert abs(float(info[6])) < 1e-6 and abs(float(info[1])) == 0.0      # ssim 1, mse 0 (comparing-baseline.py:88)
    a2 = R.randn("loss.a2", (1, 3, 40, 24), 1, 0.5).cuda()
    _, info = losses.space_loss(a2, a2 * 0.7 + 0.1)
    assert abs((1 - float(info[6])) - float(g["ssim_40x24"])) < 1e-5


@pytest.mark.parametrize("B,H,W", [(1, 1024, 1024), (2, 44, 30)])
def test_tsa_loss_and_gradient_vs_oracle(B, H, W):
    """loss_imgs + 5*loss_medium + 9*loss_small at 1024 (pool x4: the 16-byte forms of the three merged-window kernels) and on a
    small image whose width is not a multiple of 4 (their scalar forms), against the oracle's autograd."""
    from dge_amd import losses
    a = R.randn("tsa.a", (B, 3, H, W), 3, 0.4)
    b = (a * 0.8 + R.randn("tsa.b", (B, 3, H, W), 3, 0.2)).requires_grad_(True)
    zero_lp = lambda x, y: torch.zeros(x.shape[0], 1, 1, 1)
    tot = 0
    for wgt, (x1, x2) in zip((1, 5, 9), zip([a, *O.attention_crops(a)], [b, *O.attention_crops(b)])):
        l, _ = O.space_loss(x1, x2, lpips_fn=zero_lp)
        tot = tot + wgt * l
    tot.backward()
    bg = b.detach().cuda().requires_grad_(True)
    loss, info = losses.image_loss_tsa(a.cuda(), bg)
    loss.backward()
    assert abs(float(loss) - float(tot)) < 2e-4 * abs(float(tot))
    err = (bg.grad.cpu() - b.grad).abs().max() / b.grad.abs().max()
    assert err < 2e-3, err


def test_lreq_adam_vs_reference_golden():
    from dge_amd.custom_adam import LREQAdam
    g = golden("adam.npz")
    names = ["lin.weight", "lin.bias", "conv.weight", "plain"]
    shapes = [(7, 12), (7,), (6, 4, 3, 3), (1, 6, 1, 1)]
    coef = g["coef"]
    params = []
    for k, s, c in zip(names, shapes, coef):
        p = torch.nn.Parameter(R.randn("adam.p." + k, s, 0, 0.3).cuda())
        if c >= 0:
            p.lr_equalization_coef = float(c)
        params.append(p)
    opt = LREQAdam([{"params": params}], lr=0.0015, betas=(0.0, 0.99), weight_decay=0)
    for step in range(3):
        for k, s, p in zip(names, shapes, params):
            p.grad = R.randn(f"adam.g{step}." + k, s, 0, 0.01 * (step + 1)).cuda()
        if step == 1:
            params[3].grad = None
        opt.step()
        for k, p in zip(names, params):
            ref = g[f"s{step}:{k}"]
            assert np.abs(p.detach().cpu().numpy() - ref).max() < 1e-5 * np.abs(ref).max() + 1e-7, (step, k)
    with pytest.raises(ValueError):
        LREQAdam(params, betas=(0.5, 0.99))           # custom_adam.py:14
