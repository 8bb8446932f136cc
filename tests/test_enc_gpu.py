"""GPU parity of the encoder HIP path against the reference's golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from tests.conftest import golden
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
    E.load_state_dict(R.fill_encoder(enc_shapes(16, 64, 4), seed=21))
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
