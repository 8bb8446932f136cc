"""LPIPS structure on the HIP kernels vs the CPU restatement (oracle/lpips_ref.py) with the same
seeded stand-in weights.  Parity with the real `lpips` package is UNPINNED (package and weights
are absent from the reference tree and this image) - see DESIGN.md."""
import pytest
import torch

from tests.golden import recipe as R
from oracle import lpips_ref as LR


def test_state_dict_names_follow_the_lpips_package():
    from dge_amd.lpips import LPIPS
    m = LPIPS()
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert mine == LR.param_shapes()


@pytest.mark.gpu
@pytest.mark.parametrize("cd,shape", [("f32", (2, 3, 48, 40)), ("bf16", (1, 3, 64, 64)), ("f32", (1, 3, 44, 44))])
def test_lpips_value_and_gradient_vs_oracle(cd, shape):
    from dge_amd.lpips import LPIPS
    P = LR.seeded_params(0)
    m = LPIPS(compute_dtype=cd).cuda()
    m.load_state_dict(P)
    a = R.randn("lp.a", shape, 4, 0.5).clamp(-1, 1)
    b = (a * 0.7 + R.randn("lp.b", shape, 4, 0.3)).clamp(-1, 1).requires_grad_(True)
    ref = LR.lpips(P, a, b).mean()
    ref.backward()
    val, gb = m.value_and_grad(a.cuda(), b.detach().cuda())
    tol_v, tol_g = (2e-4, 5e-3) if cd == "f32" else (3e-2, 0.25)
    assert abs(float(val) - float(ref)) < tol_v * abs(float(ref)), (float(val), float(ref))
    ga, gr = gb.cpu().flatten(), b.grad.flatten()
    l2 = ((ga - gr).norm() / gr.norm()).item()
    assert l2 < tol_g, l2
