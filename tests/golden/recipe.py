"""Deterministic tensor recipes shared by tools/gen_golden.py (run once in the build
container, with /root/reference importable) and by the tests (run anywhere).

Fixtures under tests/golden/ hold only *inputs that cannot be re-derived* and the
*expected outputs of the reference*; weights and most inputs are regenerated from the
recipes below so the committed files stay small.  Nothing here imports the reference.
"""
import zlib
import numpy as np
import torch

_torch_randn = torch.randn   # bound early: gen_golden.py patches torch.randn


def _seed_for(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF


def randn(name: str, shape, seed: int = 0, scale: float = 1.0, shift: float = 0.0) -> torch.Tensor:
    """N(shift, scale) tensor that depends only on (name, shape, seed)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed_for(name, seed))
    return _torch_randn(tuple(shape), generator=g, dtype=torch.float32) * scale + shift


def fill_like(shapes: dict, seed: int, rules=None) -> dict:
    """Fill a {key: shape} description with seeded values.

    `rules` is a list of (substring, scale, shift); first match wins.  Default N(0,1).
    Non-trivial values are used for tensors the reference initialises to 0/1 (biases,
    noise strengths, w_avg ...) so a kernel that ignores them cannot pass."""
    rules = rules or []
    out = {}
    for k, shp in shapes.items():
        scale, shift = 1.0, 0.0
        for sub, sc, sh in rules:
            if sub in k:
                scale, shift = sc, sh
                break
        out[k] = randn(k, shp, seed, scale, shift)
    return out


# StyleGAN2 generator (reference: model/stylegan2_generator.py ctor defaults are
# zeros for bias / noise_strength / w_avg).
S2_RULES = [
    ("filter.kernel", None, None),       # handled by caller (buffers keep reference values)
    ("upsample.kernel", None, None),
    ("noise_strength", 0.3, 0.1),
    ("style.bias", 0.2, 0.0),
    (".bias", 0.2, 0.0),
    ("w_avg", 0.5, 0.0),
]


def fill_s2(shapes: dict, seed: int) -> dict:
    out = {}
    fir = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32)
    fir = fir / fir.sum() * 4.0
    for k, shp in shapes.items():
        if k.endswith("filter.kernel") or k.endswith("upsample.kernel"):
            out[k] = torch.from_numpy(fir.copy()).view(1, 1, 4, 4)
            continue
        scale, shift = 1.0, 0.0
        for sub, sc, sh in S2_RULES:
            if sc is not None and sub in k:
                scale, shift = sc, sh
                break
        out[k] = randn(k, shp, seed, scale, shift)
    return out


# Encoder E.BE (reference: model/E/E.py; noise_weight_* / bias_* start at zero, lreq
# weights start at N(0, gain/sqrt(fan_in))).
def fill_encoder(shapes: dict, seed: int) -> dict:
    out = {}
    for k, shp in shapes.items():
        if "noise_weight" in k:
            out[k] = randn(k, shp, seed, 0.2, 0.05)
        elif k.endswith("bias") or "bias_" in k:
            out[k] = randn(k, shp, seed, 0.1, 0.0)
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else 1
            out[k] = randn(k, shp, seed, float(np.sqrt(2.0) / np.sqrt(max(fan_in, 1))), 0.0)
    return out


def checksum(sd: dict) -> float:
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


def fill_biggan(shapes: dict, seed: int) -> dict:
    """BigGAN-deep state (reference model/biggan_generator.py).  Spectral-norm u/v vectors are
    brought close to the leading singular pair by 5 deterministic power iterations so that
    sigma = u.Wv is O(1) (random u,v would give sigma ~ 0 and overflowing weights)."""
    out = {}
    for k, shp in shapes.items():
        if k.endswith("running_vars"):
            out[k] = randn(k, shp, seed, 0.2, 1.0).abs() + 0.3
        elif k.endswith("running_means"):
            out[k] = randn(k, shp, seed, 0.3)
        elif k.endswith("weight_u") or k.endswith("weight_v"):
            v = randn(k, shp, seed)
            out[k] = v / v.norm()
        elif k.endswith("gamma"):
            out[k] = torch.tensor([0.7])
        elif k.endswith(".bias"):
            out[k] = randn(k, shp, seed, 0.1)
        elif k.endswith("bn.weight"):
            out[k] = randn(k, shp, seed, 0.2, 1.0)
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else 1
            out[k] = randn(k, shp, seed, 1.0 / np.sqrt(max(fan_in, 1)))
    for k in list(out):
        if k.endswith("weight_orig"):
            base = k[: -len("weight_orig")]
            w = out[k].reshape(out[k].shape[0], -1)
            u, v = out[base + "weight_u"], out[base + "weight_v"]
            for _ in range(5):
                v = torch.mv(w.t(), u); v = v / v.norm()
                u = torch.mv(w, v); u = u / u.norm()
            out[base + "weight_u"], out[base + "weight_v"] = u, v
    return out


def fill_encbig(shapes: dict, seed: int) -> dict:
    """E_BIG.BE state: lreq/plain convs like fill_encoder, conditional-BN parts like fill_biggan."""
    bn = {k: v for k, v in shapes.items() if "batch_norm" in k}
    rest = {k: v for k, v in shapes.items() if "batch_norm" not in k}
    out = fill_encoder(rest, seed)
    out.update(fill_biggan(bn, seed))
    return {k: out[k] for k in shapes}


# Grad-CAM++ attention path (reference metric/grad_cam.py on a vgg16-layout classifier): narrow stand-in widths,
# an input whose last-conv map is 6x6 and whose pooled map (3x3) is *smaller* than the 7x7 adaptive pool target.
GRADCAM_CFG = dict(widths=(32, 32, "M", 32, 32, "M", 64, 64, 64, "M", 64, 64, 64, "M", 64, 64, 64, "M"),
                   fc=128, classes=40, seed=3, N=3, H=96, W=96)


def gradcam_images(tag: str, N: int, H: int, W: int) -> torch.Tensor:
    """Smooth-ish images in [-1,1] (low-resolution noise upsampled + fine noise)."""
    lo = randn("gradcam.lo." + tag, (N, 3, H // 8, W // 8), 0, 0.8)
    up = torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False)
    return (up + randn("gradcam.hi." + tag, (N, 3, H, W), 0, 0.15)).clamp(-1, 1)
