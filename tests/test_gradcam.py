"""Grad-CAM++ attention path: reference metric/grad_cam.py driven as in
E_mis_align_cropping_s1.py:99-106,159-170.  Golden vectors (tests/golden/gradcam.npz) come from the reference's own
GradCamPlusPlus / GradCAM / GuidedBackPropagation / mask2cam classes run on a torchvision-layout VGG16 with narrow
seeded stand-in weights (tools/gen_golden_gradcam.py; cv2.resize / applyColorMap restated, see oracle/gradcam_ref.py).
CPU: the oracle restatement reproduces them.  GPU: the HIP path reproduces them through the C ABI."""
import numpy as np
import pytest
import torch

from tests.conftest import golden
from tests.golden import recipe as R
from oracle import gradcam_ref as GR

CFG = R.GRADCAM_CFG


def _state():
    net = GR.VGG16Ref(CFG["widths"], CFG["fc"], CFG["classes"])
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    return shapes, GR.seeded_state(shapes, CFG["seed"])


def _imgs(tag):
    return R.gradcam_images("a" if tag == "1" else "b", CFG["N"], CFG["H"], CFG["W"])


# ------------------------------------------------------------------------------------------------ CPU: oracle vs golden
@pytest.mark.parametrize("tag", ["1", "2"])
def test_oracle_reproduces_the_reference_classes(tag):
    g = golden("gradcam.npz")
    _, sd = _state()
    imgs = _imgs(tag)
    m, index_max, logits, grad, _ = GR.grad_cam_pp(sd, imgs, None, guided=True)
    assert index_max == int(g["index_max_" + tag])
    np.testing.assert_allclose(logits.numpy(), g["logits_" + tag], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(grad.numpy(), g["gradient_" + tag], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(m.numpy(), g["mask_" + tag], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(GR.grad_cam_pp(sd, imgs, None, plain=True)[0].numpy(), g["mask_plain_" + tag], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(GR.guided_backprop(sd, imgs).numpy(), g["gbp_" + tag], rtol=1e-4, atol=1e-9)
    h, c = GR.mask2cam(torch.tensor(g["mask_" + tag]).double(), imgs)
    np.testing.assert_allclose(h.numpy(), g["heat_" + tag], atol=1e-6)
    np.testing.assert_allclose(c.numpy(), g["cam_" + tag], atol=1e-5)


def test_module_surface_follows_torchvision_vgg16_and_the_reference_names():
    from dge_amd import grad_cam as G
    shapes, _ = _state()
    m = G.VGG16(CFG["widths"], CFG["fc"], CFG["classes"])
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == shapes
    assert m.final_layer == str(golden("gradcam.npz")["final_layer"]) == "features.28"
    full = GR.VGG16Ref.__init__.__defaults__        # (widths, fc, classes) of torchvision's vgg16
    assert tuple(G.VGG16_WIDTHS) == tuple(full[0]) and full[1:] == (4096, 1000)
    for name in ("GradCAM", "GradCamPlusPlus", "GuidedBackPropagation", "mask2cam"):
        assert hasattr(G, name)
    assert (G._jet_table()[:, ::-1] == GR.jet_lut()).all()      # the oracle's table is B,G,R (OpenCV), the module's R,G,B


def test_cv2_resize_restatement_basic_properties():
    a = np.arange(12, dtype=np.float64).reshape(3, 4)
    assert np.array_equal(GR.cv2_resize_linear(a, (4, 3)), a)                       # identity size
    up = GR.cv2_resize_linear(a, (8, 6))
    assert up.shape == (6, 8) and up[0, 0] == a[0, 0] and up[-1, -1] == a[-1, -1]    # clamped corners
    assert abs(up[0, 1] - (0.75 * a[0, 0] + 0.25 * a[0, 1])) < 1e-6                  # half-pixel centres


def test_grad_cam_refuses_cpu_tensors():
    from dge_amd import grad_cam as G
    from dge_amd._lib import DgeError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = G.VGG16(CFG["widths"], CFG["fc"], CFG["classes"])
    with pytest.raises(DgeError):
        G.GradCamPlusPlus(m, m.final_layer)(torch.zeros(1, 3, 32, 32), None)


# ------------------------------------------------------------------------------------------------ GPU: HIP path vs golden
def _net(cd):
    from dge_amd import grad_cam as G
    _, sd = _state()
    m = G.VGG16(CFG["widths"], CFG["fc"], CFG["classes"], compute_dtype=cd)
    m.load_state_dict(sd)
    return G, m.cuda()


def _nchw(t):
    return t.float().permute(0, 3, 1, 2).cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["1", "2"])
def test_gradcampp_gbp_mask2cam_f32_vs_reference(tag):
    g = golden("gradcam.npz")
    G, net = _net("f32")
    imgs = _imgs(tag).cuda()
    gcpp = G.GradCamPlusPlus(net, net.final_layer)
    gbp = G.GuidedBackPropagation(net)                       # shared network: from here on every backward is guided
    plain = G.GradCAM(net, net.final_layer)
    logits = net(imgs).cpu().numpy()
    ref = g["logits_" + tag]
    assert np.abs(logits - ref).max() < 2e-4 * np.abs(ref).max()
    mask = gcpp(imgs, None)
    assert int(gcpp.index[0]) == int(g["index_max_" + tag])
    assert gcpp.index[1:].cpu().tolist() == ref.argmax(1).tolist()
    grad, gref = _nchw(gcpp.gradient), g["gradient_" + tag]
    assert np.abs(grad - gref).max() < 2e-3 * np.abs(gref).max()
    assert mask.shape == (CFG["N"], 1, CFG["H"], CFG["W"]) and mask.dtype == torch.float32
    # tolerance: f32 accumulation order of the 13 convs; masks are in [0,1]
    assert np.abs(mask.cpu().numpy() - g["mask_" + tag]).max() < 2e-3
    assert np.abs(plain(imgs, None).cpu().numpy() - g["mask_plain_" + tag]).max() < 2e-3
    assert np.abs(gcpp(imgs, np.array([3] * CFG["N"])).cpu().numpy() - g["mask_idx_" + tag]).max() < 2e-3
    gi, gir = gbp(imgs).cpu().numpy(), g["gbp_" + tag]
    assert np.abs(gi - gir).max() < 3e-3 * np.abs(gir).max()
    m2, gi2 = gcpp.with_input_gradient(imgs)                 # shared forward/backward: identical results
    assert torch.equal(m2, mask) and np.array_equal(gi2.cpu().numpy(), gi)
    # mask2cam on the reference's mask (isolates it from the network): the JET index is uint8(255*mask) by truncation,
    # so a pixel whose 255*mask sits within rounding of an integer may take the neighbouring table entry (one step is
    # <= 4/255 per channel); everything else is exact to f32 rounding
    heat, cam = G.mask2cam(torch.from_numpy(g["mask_" + tag]).cuda(), imgs)
    dh = np.abs(heat.cpu().numpy() - g["heat_" + tag])
    assert dh.max() <= 4.0 / 255 + 1e-6 and (dh > 1e-6).mean() < 1e-3
    dc = np.abs(cam.cpu().numpy() - g["cam_" + tag])
    assert dc.max() < 1e-2 and (dc > 1e-5).mean() < 1e-3


@pytest.mark.gpu
def test_unguided_network_and_oracle_agree():
    """Without GuidedBackPropagation the ReLU backward is the plain one (net.guided False)."""
    G, net = _net("f32")
    _, sd = _state()
    imgs = _imgs("1")
    ref = GR.grad_cam_pp(sd, imgs, None, guided=False)
    gcpp = G.GradCamPlusPlus(net, net.final_layer)
    mask = gcpp(imgs.cuda(), None)
    gref = ref[3].numpy()
    assert np.abs(_nchw(gcpp.gradient) - gref).max() < 2e-3 * np.abs(gref).max()
    assert np.abs(mask.cpu().numpy() - ref[0].numpy()).max() < 2e-3


@pytest.mark.gpu
def test_gradcampp_bf16_close_to_reference():
    g = golden("gradcam.npz")
    G, net = _net("bf16")
    imgs = _imgs("1").cuda()
    G.GuidedBackPropagation(net)
    gcpp = G.GradCamPlusPlus(net, net.final_layer)
    mask = gcpp(imgs, [int(g["index_max_1"])] * CFG["N"]).cpu().numpy()     # class fixed: bf16 logits may reorder near-ties
    ref = g["mask_1"]
    # bf16 storage through 13 convs: the map is a normalised sum of >= 0 features, judged on correlation and mean error
    assert np.corrcoef(mask.ravel(), ref.ravel())[0, 1] > 0.98
    assert np.abs(mask - ref).mean() < 0.03


@pytest.mark.gpu
def test_full_size_vgg16_properties():
    """The real widths (138 M parameters, stand-in weights) at 256x256, bf16: shapes and the invariants of the maps."""
    from dge_amd import grad_cam as G
    net = G.VGG16(compute_dtype="bf16").cuda()
    gcpp = G.GradCamPlusPlus(net, net.final_layer)
    gbp = G.GuidedBackPropagation(net)
    imgs = R.gradcam_images("a", 2, 256, 256).cuda()
    mask = gcpp(imgs, None)
    assert gcpp.feature.shape == (2, 16, 16, 512) and mask.shape == (2, 1, 256, 256)
    m = mask.cpu().numpy()
    assert np.isfinite(m).all() and m.min() >= 0.0 and m.max() <= 1.0 + 1e-6
    assert (gcpp.gradient.float() >= 0).all()                # guided gradients are non-negative
    gi = gbp(imgs)
    assert gi.shape == imgs.shape and torch.isfinite(gi).all() and float(gi.abs().max()) > 0
    heat, cam = G.mask2cam(mask, imgs)
    c = cam.cpu().numpy()
    assert abs(c[0].max() - 1.0) < 1e-5 and abs(c[1].max() - 1.0) < 1e-5 and np.isfinite(c).all()
    assert heat.min() >= 0 and heat.max() <= 1


@pytest.mark.gpu
def test_mis_align_iteration_matches_reference_run():
    """Two iterations of E_mis_align_cropping_s1.py's loop body (mtype 2) against the reference's own run of it
    (tests/golden/step_misalign.npz, tools/gen_golden_gradcam.py): masks, the four logged image-space losses, the latent
    loss and the encoder parameters after the (latent-only) update."""
    import dge_amd
    from dge_amd.encoder import BE
    from dge_amd.lpips import LPIPS
    from dge_amd.mis_align import MisAlignStep
    from tests.helpers import s2_shapes, enc_shapes
    from oracle import ref_torch as O
    from oracle import lpips_ref as LR
    g = golden("step_misalign.npz")
    Gen = dge_amd.StyleGAN2Generator(64, fmaps_base=2048, fmaps_max=128, compute_dtype="f32").cuda()
    Gen.load_state_dict(R.fill_s2(s2_shapes(64, fmaps_base=2048, fmaps_max=128), seed=11))
    Gen.train()
    for p in Gen.parameters():
        p.requires_grad_(False)
    E = BE(startf=16, maxf=64, layer_count=5, compute_dtype="f32").cuda()
    E.load_state_dict(R.fill_encoder(enc_shapes(16, 64, 5), seed=31))
    LP = LPIPS(compute_dtype="f32").cuda()
    LP.load_state_dict(LR.seeded_params(0))
    _, net = _net("f32")
    st = MisAlignStep(Gen, E, LP, net, lr=0.0015, batch_size=2)
    new_z = R.randn("step.new_z", (2, 512), 1).cuda()
    rel = lambda a, b: float(np.abs(np.asarray(a) - b).max() / (np.abs(b).max() + 1e-30))
    for it in range(2):
        z = R.randn(f"step.z{it}", (2, 512), 1)
        noises = [R.randn(f"step.it{it}.noise{i}", s, 1).cuda() for i, s in enumerate(O.enc_noise_shapes(5, 2, 64))]
        r = st.step(it, z=z, noises=noises, new_z=new_z)
        assert rel(r["w2"].cpu().numpy(), g[f"it{it}_w2"]) < 2e-3
        for k in ("mask_1", "mask_2"):
            assert np.abs(r[k].cpu().numpy() - g[f"it{it}_{k}"]).max() < 5e-3, (it, k)
        ref_l = g[f"it{it}_losses"]          # loss_tsa, imgs, mask, Gcam, grad, w
        got = [float(r["loss_tsa"]), float(r["info_imgs"][0]), float(r["info_mask"][0]), float(r["info_Gcam"][0]),
               float(r["info_grad"][0]), float(r["loss_w"])]
        for a, b in zip(got, ref_l):
            assert abs(a - b) < 5e-3 * abs(b), (it, got, ref_l)
        ref_info = g[f"it{it}_info"]         # rows imgs, mask, Gcam, grad, w; columns mse, mean, std, kl, cos, ssim, lpips
        for row, key in enumerate(("info_imgs", "info_mask", "info_Gcam", "info_grad")):
            info = r[key].cpu().numpy()
            for col in (0, 4, 5, 6):
                assert abs(info[1 + col] - ref_info[row, col]) < 1e-2 * abs(ref_info[row, col]) + 1e-6, (it, key, col)
        sd = E.state_dict()
        for key in g.files:
            if key.startswith(f"it{it}_after:"):
                assert rel(sd[key.split(":", 1)[1]].cpu().numpy(), g[key]) < 1e-4, (it, key)
        cs = float(g[f"it{it}_param_checksum"])
        assert abs(R.checksum({k: v.cpu() for k, v in sd.items()}) - cs) < 1e-5 * cs
