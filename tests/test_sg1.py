"""StyleGAN1 generator / mapping: oracle vs reference golden (CPU) and the HIP
path vs golden (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from tests.conftest import golden, ROOT, with_fixture_params, meas
from tests.golden import recipe as R
from oracle import ref_torch as O


def sg1_shapes(startf, maxf, layer_count, latent=512):
    s = {}
    mul = 2 ** (layer_count - 1)
    inputs = min(maxf, startf * mul)
    s["const"] = [1, inputs, 4, 4]
    for i in range(layer_count):
        outputs = min(maxf, startf * mul)
        p = f"decode_block.{i}."
        for k in ("noise_weight_1", "bias_1", "noise_weight_2", "bias_2"):
            s[p + k] = [1, outputs, 1, 1]
        s[p + "blur.weight"] = [outputs, 1, 3, 3]
        if i != 0:
            s[p + "conv_1.weight"] = [inputs, outputs, 3, 3] if (4 << i) >= 128 else [outputs, inputs, 3, 3]
        s[p + "style_1.weight"], s[p + "style_1.bias"] = [2 * outputs, latent], [2 * outputs]
        s[p + "conv_2.weight"] = [outputs, outputs, 3, 3]
        s[p + "style_2.weight"], s[p + "style_2.bias"] = [2 * outputs, latent], [2 * outputs]
        inputs = outputs
        mul //= 2
    for i in range(layer_count):
        c = min(maxf, startf * 2 ** (layer_count - 1 - i))
        s[f"to_rgb.{i}.to_rgb.weight"], s[f"to_rgb.{i}.to_rgb.bias"] = [3, c, 1, 1], [3]
    return s


def small_params():
    shapes = sg1_shapes(32, 64, 6)
    sd = R.fill_encoder(shapes, seed=41)
    blur = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16.0
    for k in sd:
        if k.endswith("blur.weight"):
            sd[k] = blur.view(1, 1, 3, 3).repeat(shapes[k][0], 1, 1, 1)
    sd["const"] = R.randn("sg1.const", tuple(shapes["const"]), 41)
    return sd


def relerr(a, b):
    a = a.detach().float().cpu(); b = torch.as_tensor(np.asarray(b)).float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def test_shapes_match_reference_state_dicts():
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "sg1_keys.json")))
    for tag, (sf, lc) in (("256_64_7", (64, 7)), ("1024_16_9", (16, 9))):
        mine = sg1_shapes(sf, 512, lc)
        assert set(mine) == set(k["Gs_" + tag]) and all(mine[n] == k["Gs_" + tag][n] for n in mine)
    assert len(k["Gs_256_64_7"]) == 91 and len(k["Gs_1024_16_9"]) == 117 and len(k["Gm"]) == 16
    import dge_amd.stylegan1 as S
    for tag, (sf, lc) in (("256_64_7", (64, 7)), ("1024_16_9", (16, 9))):
        sd = S.Generator(startf=sf, maxf=512, layer_count=lc, latent_size=512).state_dict()
        assert list(sd.keys()) == list(k["Gs_" + tag].keys())
        assert all(list(sd[n].shape) == k["Gs_" + tag][n] for n in sd)
    sd = S.Mapping(num_layers=14).state_dict()
    assert {n: list(v.shape) for n, v in sd.items()} == k["Gm"]


def test_oracle_vs_reference_golden():
    g = golden("sg1_small.npz")
    P = small_params()
    assert abs(R.checksum(P) - float(g["state_checksum"])) < 1e-6 * float(g["state_checksum"])
    styles = R.randn("sg1.styles", (2, 12, 512), 6)
    noises = [R.randn(f"sg1.noise{i}", tuple(s), 6) for i, s in enumerate(g["noise_shapes"].tolist())]
    img = O.sg1_generator(P, styles, 5, noises)
    assert relerr(img, g["image"]) < 2e-4
    noises3 = [R.randn(f"sg1b.noise{i}", tuple(s), 6) for i, s in enumerate(g["noise_shapes"].tolist()[:8])]
    assert relerr(O.sg1_generator(P, styles, 3, noises3), g["image_lod3"]) < 2e-4
    M = {f"block_{i}.fc.{n}": R.randn(f"sg1m.block_{i}.fc.{n}", (512, 512) if n == "weight" else (512,), 42, 0.05 if n == "weight" else 0.01)
         for i in range(1, 9) for n in ("weight", "bias")}
    coefs = torch.tensor([0.7] * 6 + [1.0] * 6)
    w = O.sg1_mapping(M, R.randn("sg1m.z", (3, 512), 42), R.randn("sg1m.buffer1", (12, 512), 42, 0.5), coefs)
    assert relerr(w, g["mapping_w"]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("cd", ["f32", "bf16"])
def test_hip_generator_vs_reference_golden(cd):
    import dge_amd.stylegan1 as S
    g = golden("sg1_small.npz")
    G = S.Generator(startf=32, maxf=64, layer_count=6, latent_size=512, compute_dtype=cd).cuda()
    G.load_state_dict(small_params())
    styles = R.randn("sg1.styles", (2, 12, 512), 6).cuda()
    noises = [R.randn(f"sg1.noise{i}", tuple(s), 6) for i, s in enumerate(g["noise_shapes"].tolist())]
    img = G.forward(styles, 5, noises=noises)
    tol = 3e-4 if cd == "f32" else 6e-2
    assert relerr(img, g["image"]) < tol, relerr(img, g["image"])
    noises3 = [R.randn(f"sg1b.noise{i}", tuple(s), 6) for i, s in enumerate(g["noise_shapes"].tolist()[:8])]
    assert relerr(G.forward(styles, 3, noises=noises3), g["image_lod3"]) < tol


@pytest.mark.gpu
def test_hip_mapping_vs_reference_golden():
    import dge_amd.stylegan1 as S
    g = golden("sg1_small.npz")
    M = S.Mapping(num_layers=12).cuda()
    M.load_state_dict({k: R.randn("sg1m." + k, tuple(v.shape), 42, 0.05 if k.endswith("weight") else 0.01)
                       for k, v in M.state_dict().items()})
    M.buffer1 = R.randn("sg1m.buffer1", (12, 512), 42, 0.5)
    layer_idx = torch.arange(12)[None, :, None]
    coefs = torch.where(layer_idx < 6, 0.7 * torch.ones(1, 12, 1), torch.ones(1, 12, 1))
    w = M(R.randn("sg1m.z", (3, 512), 42), coefs_m=coefs)
    assert relerr(w, g["mapping_w"]) < 1e-5


def l2rel(a, b):
    a = a.detach().float().cpu(); b = torch.as_tensor(np.asarray(b)).float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_oracle_style_gradient_vs_reference_golden():
    """Pins the oracle's differentiated synthesis (autograd through the restatement) on the reference's own gradient."""
    g = golden("sg1_small.npz")
    gg = golden("sg1_grad.npz")
    P = with_fixture_params(small_params(), gg)
    for tag, lod, prefix, nn_ in (("", 5, "sg1", 12), ("_lod3", 3, "sg1b", 8)):
        styles = R.randn("sg1.styles", (2, 12, 512), 6).requires_grad_(True)
        noises = [R.randn(f"{prefix}.noise{i}", tuple(s), 6) for i, s in enumerate(g["noise_shapes"].tolist()[:nn_])]
        img = O.sg1_generator(P, styles, lod, noises)
        gimg = R.randn("sg1.gimg" + tag, tuple(img.shape), 7)
        loss = (img * gimg).sum()
        loss.backward()
        assert relerr(img, gg["image" + tag]) < 2e-4
        assert abs(float(loss) - float(gg["loss" + tag])) < 2e-4 * abs(float(gg["loss" + tag])) + 1e-3
        assert l2rel(styles.grad, gg["g_styles" + tag]) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("cd", ["f32", "bf16"])
def test_hip_style_gradient_vs_reference_golden(cd):
    """d(image)/d(styles) of the HIP pipeline (hand-written backward, autograd_sg1) against the reference's autograd:
    covers the fused transposed-conv block (128^2), the upscale2d+conv blocks and the constant-input block."""
    import dge_amd.stylegan1 as S
    g = golden("sg1_small.npz")
    gg = golden("sg1_grad.npz")
    G = S.Generator(startf=32, maxf=64, layer_count=6, latent_size=512, compute_dtype=cd).cuda()
    G.load_state_dict(with_fixture_params(small_params(), gg))
    for tag, lod, prefix, nn_ in (("", 5, "sg1", 12), ("_lod3", 3, "sg1b", 8)):
        styles = R.randn("sg1.styles", (2, 12, 512), 6).cuda().requires_grad_(True)
        noises = [R.randn(f"{prefix}.noise{i}", tuple(s), 6) for i, s in enumerate(g["noise_shapes"].tolist()[:nn_])]
        img = G.forward(styles, lod, noises=noises)
        gimg = R.randn("sg1.gimg" + tag, tuple(img.shape), 7).cuda()
        (img.float() * gimg).sum().backward()
        want = torch.as_tensor(gg["g_styles" + tag])
        got = styles.grad.float().cpu()
        err = l2rel(got, want)
        cos = torch.nn.functional.cosine_similarity(got.flatten(), want.flatten(), dim=0).item()
        meas("sg1_style_grad", cd=cd, lod=lod, l2=err, cos=cos)
        if cd == "f32":
            # 1e-4 at both lods (measured 2.1e-6 / 3.4e-6): the fixture's biases keep every pre-activation >= 1e-4 * max away from
            # the leaky-relu kink (tools/gen_golden.py: clear_kinks) and the run is deterministic (tests/conftest.py).  Before, ONE
            # pre-activation of block 2 within f32 rounding of zero moved the lod-3 gradient by 1.7 %.
            assert err < 1e-4 and cos > 0.999999, (tag, cos, err)
        else:
            # bf16 activations: the forward itself deviates by a few % of the image range (12 convs + 12 instance norms on
            # bf16-rounded activations), so the gradient is taken at a slightly different point (leaky-relu masks flip for
            # ~0.5 % of the elements per layer).  Top style row 1.7 %, lower rows 15-22 % L2, cosine 0.98 overall; judged
            # on direction and norm.  The f32 run above is the parity check of the backward formulas.
            # (deterministic run: lod 5 L2 0.227 / cosine 0.974, lod 3 0.150 / 0.989; bounds 1.5x)
            assert cos > (0.96 if lod == 5 else 0.984) and err < (0.34 if lod == 5 else 0.225), (tag, cos, err)
        # layers above the decoded level receive no gradient
        if 2 * (lod + 1) < got.shape[1]:
            assert float(got[:, 2 * (lod + 1):].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", [(32, 16), (64, 32)])
def test_fused_up_conv_and_its_data_gradient(cin, cout):
    """ConvTranspose2d(3, stride 2, pad 1) with transform_kernel (lreq.py:129-131,145-147) as the phase-folded implicit GEMM
    (DGE_PACK_SG1_UP) and its adjoint (DGE_PACK_SG1_UP_DGRAD, space-to-depth read), against torch on the CPU.  Cout = 16 is
    the top block of the FFHQ-1024 generator: a 32-wide N tile then spans two output phases."""
    import torch.nn.functional as F
    from dge_amd import ops
    gen = torch.Generator().manual_seed(cin + cout)
    B, H = 2, 12
    x = torch.randn(B, cin, H, H, generator=gen, requires_grad=True)
    w = torch.randn(cin, cout, 3, 3, generator=gen) * 0.1
    wp = F.pad(w, (1, 1, 1, 1))
    w4 = wp[:, :, 1:, 1:] + wp[:, :, :-1, 1:] + wp[:, :, 1:, :-1] + wp[:, :, :-1, :-1]
    y = F.conv_transpose2d(x, w4, stride=2, padding=1)
    gy = torch.randn(y.shape, generator=gen)
    (y * gy).sum().backward()
    xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    got = ops.conv2d(xd, ops.pack_conv_weight(w.cuda(), ops.PACK_SG1_UP, ops.F32), cout, 3, up=True)
    assert l2rel(got.permute(0, 3, 1, 2), y.detach()) < 1e-5
    gyd = gy.permute(0, 2, 3, 1).contiguous().cuda()
    st = torch.zeros(B, cin, 2, device="cuda")
    gx = ops.conv2d(gyd, ops.pack_conv_weight(w.cuda(), ops.PACK_SG1_UP_DGRAD, ops.F32), cin, 3, in_s2d=True, stats=st, dot_src=xd)
    assert l2rel(gx.permute(0, 3, 1, 2), x.grad) < 1e-5
    assert l2rel(st[..., 0], (x.grad * x.detach()).sum((2, 3))) < 1e-4 and l2rel(st[..., 1], x.grad.sum((2, 3))) < 1e-4
