"""Weight-import path of the LPIPS / VGG16 modules (host logic, no GPU): the two files a user of the reference has
(torchvision vgg16 checkpoint + the lpips package's linear heads, training_utils.py:93 / requirements.txt:12) map onto the
module's parameters; training refuses to run on stand-in weights unless told so."""
import pytest
import torch


def _fake_files(seed=0, full_lpips=False):
    from dge_amd.lpips import _VGG, _CONV_IDX, _SLICE
    g = torch.Generator().manual_seed(seed)
    vgg, lin = {}, {}
    ci = 0
    for item in _VGG:
        if item == "M":
            continue
        cin, cout = item
        name = f"net.slice{_SLICE[ci]}.{_CONV_IDX[ci]}" if full_lpips else f"features.{_CONV_IDX[ci]}"
        vgg[name + ".weight"] = torch.randn(cout, cin, 3, 3, generator=g)
        vgg[name + ".bias"] = torch.randn(cout, generator=g)
        ci += 1
    if not full_lpips:
        vgg["classifier.0.weight"] = torch.randn(4, 4, generator=g)         # ignored
    for k, c in enumerate([64, 128, 256, 512, 512]):
        lin[f"lin{k}.model.1.weight"] = torch.rand(1, c, 1, 1, generator=g)
        lin[f"lins.{k}.model.1.weight"] = lin[f"lin{k}.model.1.weight"]          # duplicate keys of newer lpips versions
    return vgg, lin


def test_lpips_load_pretrained_maps_torchvision_and_lpips_names(tmp_path):
    from dge_amd.lpips import LPIPS, _CONV_IDX, _SLICE
    vgg, lin = _fake_files()
    torch.save(vgg, tmp_path / "vgg16.pth")
    torch.save(lin, tmp_path / "vgg_lin.pth")
    m = LPIPS(compute_dtype="f32")
    assert m.pretrained is False
    m.load_pretrained(str(tmp_path / "vgg16.pth"), str(tmp_path / "vgg_lin.pth"))
    assert m.pretrained is True
    sd = m.state_dict()
    for ci, idx in enumerate(_CONV_IDX):
        assert torch.equal(sd[f"net.slice{_SLICE[ci]}.{idx}.weight"], vgg[f"features.{idx}.weight"])
        assert torch.equal(sd[f"net.slice{_SLICE[ci]}.{idx}.bias"], vgg[f"features.{idx}.bias"])
    for k in range(5):
        assert torch.equal(sd[f"lin{k}.model.1.weight"], lin[f"lin{k}.model.1.weight"])
    # a complete lpips.LPIPS(net='vgg').state_dict() in one file works as well
    full, lin2 = _fake_files(seed=1, full_lpips=True)
    full.update({k: v for k, v in lin2.items() if k.startswith("lin")})
    m2 = LPIPS(compute_dtype="f32").load_pretrained(full)
    assert m2.pretrained and torch.equal(m2.state_dict()["net.slice1.0.weight"], full["net.slice1.0.weight"])


def test_lpips_load_pretrained_fails_loudly_on_incomplete_files():
    from dge_amd.lpips import LPIPS
    vgg, lin = _fake_files()
    bad = dict(vgg); bad.pop("features.28.weight")
    with pytest.raises(KeyError):
        LPIPS(compute_dtype="f32").load_pretrained(bad, lin)
    with pytest.raises(KeyError):
        LPIPS(compute_dtype="f32").load_pretrained(vgg, {})
    wrong = dict(vgg); wrong["features.0.weight"] = torch.zeros(64, 4, 3, 3)
    m = LPIPS(compute_dtype="f32")
    with pytest.raises(ValueError):
        m.load_pretrained(wrong, lin)
    assert m.pretrained is False


def test_training_refuses_standin_lpips_unless_allowed():
    from dge_amd.lpips import LPIPS
    from dge_amd.e_align import load_lpips_weights
    m = LPIPS(compute_dtype="f32")
    with pytest.raises(RuntimeError, match="STAND-IN"):
        load_lpips_weights(m)
    with pytest.warns(UserWarning, match="STAND-IN"):
        assert load_lpips_weights(m, allow_standin=True) is m
    vgg, lin = _fake_files()
    assert load_lpips_weights(m, vgg, lin).pretrained is True
    assert load_lpips_weights(None) is None


def test_gradcam_vgg16_load_pretrained_sets_the_flag():
    from dge_amd.grad_cam import VGG16
    m = VGG16(widths=(8, "M", 8, "M"), fc=16, num_classes=10, compute_dtype="f32")
    sd = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    assert m.pretrained is False
    m.load_pretrained(sd)
    assert m.pretrained is True and torch.equal(m.state_dict()["features.0.weight"], sd["features.0.weight"])
