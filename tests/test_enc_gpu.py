"""GPU parity of the encoder HIP path against the reference's golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from tests.conftest import golden, with_fixture_params, meas, MODES
from tests.golden import recipe as R
from tests.helpers import enc_shapes
from oracle import ref_torch as O



def relerr(a, b):
    a = a.detach().float().cpu()
    b = torch.as_tensor(np.asarray(b)).float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def small_encoder(cd):
    from dge_amd.encoder import BE
    E = BE(startf=16, maxf=64, layer_count=4, compute_dtype=cd).cuda()
    E.load_state_dict(with_fixture_params(R.fill_encoder(enc_shapes(16, 64, 4), seed=21), golden("enc_small.npz")))
    return E


@pytest.mark.gpu
@pytest.mark.parametrize("cd", ["f32", "bf16"])
def test_encoder_forward_vs_reference_golden(cd):
    g = golden("enc_small.npz")
    E = small_encoder(cd)
    img = R.randn("enc.img", (2, 3, 32, 32), 9, 0.5).cuda()
    noises = [R.randn(f"enc.noise{i}", s, 9).cuda() for i, s in enumerate(O.enc_noise_shapes(4, 2, 32))]
    with torch.no_grad():
        x, w = E(img, noises=noises)
    tol = 2e-4 if cd == "f32" else 4e-2
    assert relerr(w, g["w"]) < tol, relerr(w, g["w"])
    assert relerr(x, g["x"]) < tol, relerr(x, g["x"])
    # index map (bit-exact requirement): w[:, 2(L-1-j)] = w2_j, w[:, 2(L-1-j)+1] = w1_j
    for j in range(4):
        assert relerr(w[:, 2 * (3 - j)], g[f"blk{j}_w2"]) < tol
        assert relerr(w[:, 2 * (3 - j) + 1], g[f"blk{j}_w1"]) < tol


def test_state_dict_surface():
    import json, os
    from tests.conftest import ROOT
    from dge_amd.encoder import BE
    ek = json.load(open(os.path.join(ROOT, "tests", "golden", "enc_keys.json")))
    for tag, (sf, lc) in (("1024_16_9", (16, 9)), ("256_64_7", (64, 7))):
        E = BE(startf=sf, maxf=512, layer_count=lc)
        sd = E.state_dict()
        assert list(sd.keys()) == list(ek[tag].keys())
        assert all(list(sd[k].shape) == ek[tag][k] for k in sd)
        coefs = ek[tag + "_lreq"]
        for k, p in E.named_parameters():
            c = getattr(p, "lr_equalization_coef", -1.0)
            assert abs(c - coefs[k]) < 1e-6 * max(1.0, abs(coefs[k])), k


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cd", ["f32", "bf16"])
def test_encoder_backward_vs_reference_golden(cd, mode):
    """All parameter gradients of <w, gw> against the reference's autograd (enc_small.npz).  mode "atomics" = the library's default
    reductions (f32 atomics in run-dependent order), bounds wide enough for their spread; "det" = tests/conftest.py's fixture."""
    from dge_amd import ops
    assert ops.is_deterministic() == (mode == "det")
    f32_bound = 1e-4 if mode == "det" else 2e-3
    l2_bound, cos_bound = (0.145, 0.994) if mode == "det" else (0.15, 0.99)
    g = golden("enc_small.npz")
    E = small_encoder(cd)
    img = R.randn("enc.img", (2, 3, 32, 32), 9, 0.5).cuda()
    noises = [R.randn(f"enc.noise{i}", s, 9).cuda() for i, s in enumerate(O.enc_noise_shapes(4, 2, 32))]
    x, w = E(img, noises=noises)
    gw = R.randn("enc.gw", tuple(w.shape), 9, 0.05).cuda()
    (w * gw).sum().backward()
    # f32: max-abs error relative to the tensor's max magnitude.  bf16: gradients travel through bf16
    # tensors and the instance-norm backward subtracts projections, so the bound is on direction and
    # L2 norm (cosine > 0.99, relative L2 < 0.15; f32 atomics make the sums run-to-run order dependent) on this deliberately tiny 32x32 / 4x4-bottleneck case.
    bad = {}
    worst = [0.0, 1.0]
    worst_f32 = 0.0
    for k, p in E.named_parameters():
        if "grad:" + k in g.files:
            assert p.grad is not None, k
            a, b = p.grad.float().cpu().flatten(), torch.from_numpy(g["grad:" + k]).flatten()
            if cd == "f32":
                e = ((a - b).abs().max() / b.abs().max()).item()
                worst_f32 = max(worst_f32, e)
                if not e < f32_bound:          # (measured 4.9e-6: kink-free fixture, deterministic run)
                    bad[k] = e
            else:
                l2 = ((a - b).norm() / b.norm()).item()
                cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
                worst[0], worst[1] = max(worst[0], l2), min(worst[1], cos)
                if not (l2 < l2_bound and cos > cos_bound):      # (deterministic run: worst tensor L2 0.096, cosine 0.9962; bounds 1.5x)
                    bad[k] = (l2, cos)
        else:
            assert p.grad is None, f"{k} must not receive a gradient (reference leaves it None)"
    meas("enc_bwd", cd=cd, mode=mode, worst_l2=worst[0], worst_cos=worst[1], worst_f32_maxrel=worst_f32)
    assert not bad, bad
    # retain_graph semantics: a second backward over the same saved activations works (E_align_s2.py:204-220)
    E.zero_grad()
    x, w = E(img, noises=noises)
    (w * gw).sum().backward(retain_graph=True)
    g1 = E.decode_block[0].conv_1.weight.grad.clone()
    (w * gw).sum().backward()
    assert relerr(E.decode_block[0].conv_1.weight.grad, (2 * g1).cpu()) < 1e-3   # f32 atomics: summation order varies
