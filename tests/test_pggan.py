"""PGGAN generator: oracle vs reference golden (CPU), HIP path vs golden (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from tests.conftest import golden, ROOT
from tests.golden import recipe as R
from oracle import ref_torch as O


def pg_shapes(resolution, fmaps_base=16 << 10, fmaps_max=512):
    nf = lambda r: min(fmaps_base // r, fmaps_max)
    s = {"lod": []}
    import math
    for k in range(int(math.log2(resolution)) - 1):
        res = 4 << k
        s[f"layer{2 * k}.weight"] = [nf(res), 512, 4, 4] if k == 0 else [nf(res), nf(res // 2), 3, 3]
        s[f"layer{2 * k}.bias"] = [nf(res)]
        s[f"layer{2 * k + 1}.weight"], s[f"layer{2 * k + 1}.bias"] = [nf(res), nf(res), 3, 3], [nf(res)]
        s[f"output{k}.weight"], s[f"output{k}.bias"] = [3, nf(res), 1, 1], [3]
    return s


def small_params():
    return {k: (R.randn("pg." + k, tuple(v), 51, 0.2 if k.endswith("bias") else 1.0) if len(v) else torch.zeros(()))
            for k, v in pg_shapes(32, 1024, 64).items()}


def relerr(a, b):
    a = a.detach().float().cpu(); b = torch.as_tensor(np.asarray(b)).float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def test_state_dict_and_oracle():
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "pggan_keys.json")))["256"]
    mine = pg_shapes(256)
    assert list(mine.keys()) == list(k.keys()) and all(mine[n] == k[n] for n in mine) and len(k) == 43
    from dge_amd.pggan_generator import PGGANGenerator
    sd = PGGANGenerator(256).state_dict()
    assert list(sd.keys()) == list(k.keys()) and all(list(sd[n].shape) == k[n] for n in sd)
    with pytest.raises(ValueError):
        PGGANGenerator(100)                                 # pggan_generator.py:68-70
    g = golden("pggan_small.npz")
    P = small_params()
    assert abs(R.checksum(P) - float(g["state_checksum"])) < 1e-6 * float(g["state_checksum"])
    assert relerr(O.pg_generator(P, R.randn("pg.z", (2, 512), 51)), g["image"]) < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("cd", ["f32", "bf16"])
def test_hip_generator_vs_reference_golden(cd):
    from dge_amd.pggan_generator import PGGANGenerator
    g = golden("pggan_small.npz")
    G = PGGANGenerator(32, fmaps_base=1024, fmaps_max=64, compute_dtype=cd).cuda()
    G.load_state_dict(small_params())
    r = G(R.randn("pg.z", (2, 512), 51).cuda())
    assert relerr(r["z"], g["z"]) < 1e-5
    e = relerr(r["image"], g["image"])
    assert e < (3e-4 if cd == "f32" else 6e-2), e


def _l2rel(a, b):
    a = a.detach().float().cpu().flatten(); b = torch.as_tensor(np.asarray(b)).float().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_oracle_latent_gradient_vs_reference_golden():
    """Pins the oracle's differentiated PGGAN generator on the reference's own d(image)/dz."""
    g = golden("pggan_grad.npz")
    z = R.randn("pg.z", (2, 512), 51).requires_grad_(True)
    img = O.pg_generator(small_params(), z)
    loss = (img * R.randn("pg.gimg", tuple(img.shape), 52)).sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert _l2rel(z.grad, g["g_z"]) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("cd", ["f32", "bf16"])
def test_hip_latent_gradient_vs_reference_golden(cd):
    """Hand-written data gradient of the PGGAN generator w.r.t. z (pixel-norm backward, lrelu / bias, conv data gradients
    with the fused nearest upsample's adjoint, the 4x4 'dense' first layer) against the reference's autograd."""
    from dge_amd.pggan_generator import PGGANGenerator
    g = golden("pggan_grad.npz")
    G = PGGANGenerator(32, fmaps_base=1024, fmaps_max=64, compute_dtype=cd).cuda()
    G.load_state_dict(small_params())
    for p in G.parameters():
        p.requires_grad_(False)
    z = R.randn("pg.z", (2, 512), 51).cuda().requires_grad_(True)
    img = G(z)["image"]
    loss = (img.float() * R.randn("pg.gimg", tuple(img.shape), 52).cuda()).sum()
    loss.backward()
    err = _l2rel(z.grad, g["g_z"])
    cos = torch.nn.functional.cosine_similarity(z.grad.float().cpu().flatten(), torch.as_tensor(g["g_z"]).flatten(), dim=0).item()
    if cd == "f32":
        assert abs(float(loss.detach()) - float(g["loss"])) < 2e-4 * abs(float(g["loss"])) and err < 3e-3, err
    else:
        assert cos > 0.98 and err < 0.2, (cos, err)
