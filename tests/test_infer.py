"""latent editing on the reference's shipped W+ codes / InterfaceGAN boundaries (exact index semantics),
the G -> E -> G round trip and the comparing-baseline metrics."""
import numpy as np
import pytest
import torch

from tests.conftest import golden


def test_edit_latent_reproduces_the_reference_script_bit_exactly():
    from dge_amd.infer import edit_latent
    g = golden("latent_fixtures.npz")
    w = torch.as_tensor(g["w_i4_msk"])
    assert tuple(w.shape) == (1, 18, 512)
    for tag, dname, bonus, start, end in (("a", "eyeglasses", 70, 0, 3), ("b", "smile", 100, 0, 4)):
        out = edit_latent(w, g["dir_" + dname], bonus, start, end)
        assert torch.equal(out, torch.as_tensor(g["edit_" + tag]))                       # latent indexing: bit-exact
        assert torch.equal(out[0, start + end:], w[0, start + end:]) and not torch.equal(out[0, :end], w[0, :end])
    for name in ("age", "eyeglasses", "gender", "pose", "smile"):
        d = g["dir_" + name]
        assert d.shape == (1, 512) and abs(float(np.linalg.norm(d)) - 1.0) < 1e-6       # unit boundaries


def test_load_images_follows_the_script_transforms(tmp_path):
    """rec_real_img.py:84-101 / training_utils.py:11-15 on files written here: shapes, range, the `*2-1` scale and both resamplers."""
    from PIL import Image
    from dge_amd.infer import load_images
    rng = np.random.default_rng(1)
    paths = []
    for k, (h, w) in enumerate(((40, 56), (64, 64))):
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = str(tmp_path / f"img{k}.png")
        Image.fromarray(arr).save(p)
        paths.append((p, arr))
    x = load_images([p for p, _ in paths], 32, device="cpu")
    assert x.shape == (2, 3, 32, 32) and x.dtype == torch.float32 and float(x.min()) >= -1 and float(x.max()) <= 1
    same = load_images([paths[1][0]], 64, device="cpu")                                  # no resize needed: exact pixels
    assert torch.equal(same[0], torch.from_numpy(paths[1][1]).permute(2, 0, 1).float() / 255 * 2 - 1)
    ref = np.asarray(Image.open(paths[0][0]).convert("RGB").resize((32, 32), Image.BILINEAR), dtype=np.float32) / 255 * 2 - 1
    assert np.allclose(x[0].permute(1, 2, 0).numpy(), ref, atol=1e-6)
    bic = load_images([paths[0][0]], 32, bicubic=True, device="cpu")
    refb = np.asarray(Image.open(paths[0][0]).convert("RGB").resize((32, 32)), dtype=np.float32) / 255 * 2 - 1
    assert np.allclose(bic[0].permute(1, 2, 0).numpy(), refb, atol=1e-6)


def test_skimage_ssim_restatement_known_answers():
    from oracle.metrics_ref import skimage_ssim
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 255, (40, 36, 3))
    assert abs(skimage_ssim(x, x) - 1.0) < 1e-12                                       # identical images
    assert abs(skimage_ssim(x, 255 - x) - skimage_ssim(255 - x, x)) < 1e-12              # symmetric
    flat = np.full((20, 20, 3), 100.0)
    c1 = (0.01 * 255) ** 2                                                               # constant images: only the mean term is left
    assert abs(skimage_ssim(flat, flat + 20) - (2 * 100 * 120 + c1) / (100 ** 2 + 120 ** 2 + c1)) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 3, 64, 48), (1, 3, 23, 39), (3, 1, 7, 7)])
def test_ssim_skimage_on_device_vs_oracle(shape):
    """comparing-baseline.py:25 statistic (7x7 uniform window, sample covariance, cropped border) against the scipy-based
    restatement; float32 accumulation on the device: 1e-4."""
    from dge_amd.infer import ssim_skimage
    from oracle.metrics_ref import skimage_ssim
    from tests.golden import recipe as R
    a = R.randn("ssim7.a", shape, 2, 0.5).clamp(-1, 1)
    b = (a * 0.8 + R.randn("ssim7.b", shape, 2, 0.2)).clamp(-1, 1)
    got = ssim_skimage(a.cuda(), b.cuda()).cpu().numpy()
    for i in range(shape[0]):
        ref = skimage_ssim(((a[i] + 1) * 127.5).permute(1, 2, 0).numpy(), ((b[i] + 1) * 127.5).permute(1, 2, 0).numpy())
        assert abs(got[i] - ref) < 1e-4, (i, got[i], ref)
    assert np.allclose(ssim_skimage(a.cuda(), a.cuda()).cpu().numpy(), 1.0, atol=1e-5)


@pytest.mark.gpu
def test_edit_then_synthesis_and_metrics_on_device(tmp_path):
    """embeded_img_edit.py end to end on a StyleGAN1 generator (seeded weights; the FFHQ checkpoint is not shipped): the edited
    code changes the image, the untouched rows leave the fine layers' inputs identical; metrics against a numpy restatement."""
    from dge_amd.stylegan1 import Generator
    from dge_amd.infer import edit_latent, image_metrics, save_image
    g = golden("latent_fixtures.npz")
    torch.manual_seed(0)
    L = 7                                                   # 256x256 generator, 14 style rows: the first 14 of the 18 rows are used
    Gs = Generator(startf=64, maxf=512, layer_count=L, latent_size=512, channels=3, compute_dtype="bf16").cuda()
    w = torch.as_tensor(g["w_i4_msk"])[:, :2 * L].cuda() * 0.05
    w_e = edit_latent(w, g["dir_eyeglasses"], bonus=3.0, start=0, end=3)
    noises = None
    with torch.no_grad():
        torch.manual_seed(1); a = Gs.forward(w, L - 1)
        torch.manual_seed(1); b = Gs.forward(w_e.cuda(), L - 1)
    assert a.shape == (1, 3, 256, 256) and float((a - b).abs().max()) > 0
    m = image_metrics(a, b)
    a255, b255 = (a.double().cpu() + 1) * 127.5, (b.double().cpu() + 1) * 127.5
    mse = float(((a255 - b255) ** 2).mean())
    cos = float((a.double().cpu().flatten() @ b.double().cpu().flatten()) / (a.double().norm() * b.double().norm()).cpu())
    assert abs(float(m["mse"]) - mse) < 1e-3 * mse and abs(float(m["psnr"]) - 10 * np.log10(255.0 ** 2 / mse)) < 1e-2
    assert abs(float(m["cosine"]) - cos) < 1e-4
    from oracle.metrics_ref import skimage_ssim
    assert abs(float(m["ssim"]) - skimage_ssim(a255[0].permute(1, 2, 0).numpy(), b255[0].permute(1, 2, 0).numpy())) < 1e-4
    same = image_metrics(a, a)
    assert float(same["mse"]) == 0.0 and abs(float(same["cosine"]) - 1.0) < 1e-5       # comparing-baseline.py:88 known answer
    save_image(torch.cat([a, b]), str(tmp_path / "edit.png"))
    from PIL import Image
    assert Image.open(tmp_path / "edit.png").size == (512, 256)


@pytest.mark.gpu
def test_reconstruct_round_trip_runs_for_stylegan2():
    import dge_amd
    from dge_amd.encoder import BE
    from dge_amd.e_align import EAlignStep
    from dge_amd.infer import reconstruct
    from tests.golden import recipe as R
    from tests.helpers import s2_shapes, enc_shapes
    G = dge_amd.StyleGAN2Generator(64, fmaps_base=2048, fmaps_max=128, compute_dtype="bf16").cuda()
    G.load_state_dict(R.fill_s2(s2_shapes(64, fmaps_base=2048, fmaps_max=128), seed=11))
    G.eval()
    E = BE(startf=16, maxf=64, layer_count=5, compute_dtype="bf16").cuda()
    E.load_state_dict(R.fill_encoder(enc_shapes(16, 64, 5), seed=31))
    st = EAlignStep(G, E, None, batch_size=2)
    r = reconstruct(st)
    assert r["imgs1"].shape == r["imgs2"].shape == (2, 3, 64, 64) and r["w2"].shape == r["w1"].shape == (2, 10, 512)
    assert not r["imgs2"].requires_grad and torch.isfinite(r["imgs2"]).all()
    # rec_real_img.py: images (here the generator's own) -> E -> G one at a time
    from dge_amd.infer import reconstruct_images
    w2, imgs2 = reconstruct_images(st, r["imgs1"])
    assert w2.shape == (2, 10, 512) and imgs2.shape == (2, 3, 64, 64) and torch.isfinite(imgs2).all()


@pytest.mark.gpu
def test_reconstruct_matches_the_oracle_round_trip():
    """inferE.py:101-141 (G -> E -> G once) through `reconstruct()` against the CPU oracle on the same weights, z and injected
    encoder noise: imgs1, w2, imgs2 (f32 path: tight; bf16: the storage bounds of the step test)."""
    import dge_amd
    from dge_amd.encoder import BE
    from dge_amd.e_align import EAlignStep
    from dge_amd.infer import reconstruct
    from tests.golden import recipe as R
    from tests.helpers import s2_shapes, enc_shapes
    from oracle import ref_torch as O
    PG = R.fill_s2(s2_shapes(64, fmaps_base=2048, fmaps_max=128), seed=11)
    PE = R.fill_encoder(enc_shapes(16, 64, 5), seed=31)
    z = R.randn("rec.z", (2, 512), 3)
    noises = [R.randn(f"rec.n{i}", s, 3) for i, s in enumerate(O.enc_noise_shapes(5, 2, 64))]
    with torch.no_grad():
        _, wp, imgs1 = O.s2_generator_eval(PG, z)
        _, w2 = O.enc_forward(PE, imgs1, noises)
        imgs2 = O.s2_synthesis(PG, w2)
    rel = lambda a, b: ((a.detach().float().cpu() - b).abs().max() / b.abs().max()).item()
    for cd, tol in (("f32", (5e-4, 2e-3, 2e-3)), ("bf16", (1.5e-2, 1.2e-2, 2.2e-2))):
        G = dge_amd.StyleGAN2Generator(64, fmaps_base=2048, fmaps_max=128, compute_dtype=cd).cuda()
        G.load_state_dict(PG)
        G.eval()
        E = BE(startf=16, maxf=64, layer_count=5, compute_dtype=cd).cuda()
        E.load_state_dict(PE)
        with torch.no_grad():
            r = reconstruct(EAlignStep(G, E, None, batch_size=2), z=z, noises=[n.cuda() for n in noises])
        e = (rel(r["imgs1"], imgs1), rel(r["w2"], w2), rel(r["imgs2"], imgs2))
        print(f"reconstruct {cd}: imgs1 {e[0]:.2e} w2 {e[1]:.2e} imgs2 {e[2]:.2e}")
        assert e[0] < tol[0] and e[1] < tol[1] and e[2] < tol[2], (cd, e)
        assert rel(r["w1"], wp) < 1e-5


def _args(**kw):
    import argparse
    from dge_amd.e_align import add_model_args
    a = add_model_args(argparse.ArgumentParser()).parse_args([])
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_load_models_reads_the_three_checkpoint_containers(tmp_path):
    """E_align_s2.py:30-35 (mtype 1: directory with Gs_dict.pth / Gm_dict.pth / center_tensor.pt), :51-55 (mtype 2 / 3: dict
    with `generator_smooth`, falling back to `generator`), bare encoder state_dict: written, read back through
    e_align.load_models on the host (map_location='cpu'), parameters identical."""
    from dge_amd.e_align import load_models, build_models, build_models_sg1, build_models_pg
    # mtype 2
    G, E, _ = build_models(64, 16, "f32", device="cpu", lpips=False, seed=3, fmaps_base=2048, fmaps_max=128, enc_maxf=64)
    torch.save({"generator_smooth": G.state_dict(), "generator": {k: v + 1 for k, v in G.state_dict().items()}}, tmp_path / "g2.pth")
    torch.save(E.state_dict(), tmp_path / "e2.pth")
    a = _args(mtype=2, img_size=64, start_features=16, compute_dtype="f32", fmaps_base=2048, fmaps_max=128, enc_maxf=64,
              checkpoint_dir_GAN=str(tmp_path / "g2.pth"), checkpoint_dir_E=str(tmp_path / "e2.pth"))
    G2, Gm2, E2, LP2 = load_models(a, device="cpu", lpips=False)
    assert Gm2 is None and LP2 is None
    assert all(torch.equal(v, G2.state_dict()[k]) for k, v in G.state_dict().items())
    assert all(torch.equal(v, E2.state_dict()[k]) for k, v in E.state_dict().items())
    torch.save({"generator": G.state_dict()}, tmp_path / "g2b.pth")                    # no smoothed copy in the file
    a.checkpoint_dir_GAN = str(tmp_path / "g2b.pth")
    G3 = load_models(a, device="cpu", lpips=False)[0]
    assert all(torch.equal(v, G3.state_dict()[k]) for k, v in G.state_dict().items())
    # mtype 1: a directory (the script concatenates strings: the trailing slash is the caller's)
    Gs, Gm, E1, _ = build_models_sg1(32, 16, "f32", device="cpu", lpips=False, seed=4)
    d = tmp_path / "sg1"
    d.mkdir()
    torch.save(Gs.state_dict(), d / "Gs_dict.pth")
    torch.save(Gm.state_dict(), d / "Gm_dict.pth")
    center = torch.randn(2 * 4, 512)
    torch.save(center, d / "center_tensor.pt")
    a1 = _args(mtype=1, img_size=32, start_features=16, compute_dtype="f32", checkpoint_dir_GAN=str(d) + "/")
    Gs2, Gm2, _, _ = load_models(a1, device="cpu", lpips=False)
    assert all(torch.equal(v, Gs2.state_dict()[k]) for k, v in Gs.state_dict().items())
    assert all(torch.equal(v, Gm2.state_dict()[k]) for k, v in Gm.state_dict().items())
    assert torch.equal(Gm2.buffer1, center)
    # mtype 3
    Gp, Ep, _ = build_models_pg(32, 16, "f32", device="cpu", lpips=False, seed=5)
    torch.save({"generator_smooth": Gp.state_dict()}, tmp_path / "g3.pth")
    a3 = _args(mtype=3, img_size=32, start_features=16, compute_dtype="f32", checkpoint_dir_GAN=str(tmp_path / "g3.pth"))
    Gp2 = load_models(a3, device="cpu", lpips=False)[0]
    assert all(torch.equal(v, Gp2.state_dict()[k]) for k, v in Gp.state_dict().items())
    with pytest.raises(ValueError):
        load_models(_args(mtype=7), device="cpu", lpips=False)


@pytest.mark.gpu
def test_inference_entry_points_write_the_reference_file_layout(tmp_path):
    """`python -m dge_amd.infer infer|rec|synth` (inferE.py:101-141, rec_real_img.py:84-127, synthesized_IMG.py:97-145) from
    checkpoint files: the written reconstructions equal a direct E -> G call on the same inputs."""
    import numpy as np
    from PIL import Image
    from dge_amd import infer
    from dge_amd.e_align import build_models
    G, E, _ = build_models(64, 16, "bf16", device="cuda", lpips=False, seed=3, fmaps_base=2048, fmaps_max=128, enc_maxf=64)
    torch.save({"generator_smooth": G.state_dict()}, tmp_path / "g.pth")
    torch.save(E.state_dict(), tmp_path / "e.pth")
    common = ["--mtype", "2", "--img_size", "64", "--start_features", "16", "--fmaps_base", "2048", "--fmaps_max", "128",
              "--enc_maxf", "64", "--checkpoint_dir_GAN", str(tmp_path / "g.pth"), "--checkpoint_dir_E", str(tmp_path / "e.pth")]
    out = infer.main(["infer", "--batch_size", "3", "--out", str(tmp_path / "o1")] + common)
    assert [p.split("/")[-1] for p in out] == ["v2ep4.png"]
    assert Image.open(out[0]).size == (3 * 66 + 2, 2 * 66 + 2)
    out = infer.main(["synth", "--batch_size", "2", "--iterations", "2", "--out", str(tmp_path / "o2")] + common)
    assert [p.split("/")[-1] for p in out] == ["id0_00000.png", "id0_00001.png"]
    # rec: two image files -> per-image real / reconstruction PNGs
    src = tmp_path / "imgs"
    src.mkdir()
    rng = np.random.RandomState(0)
    for i in range(2):
        Image.fromarray(rng.randint(0, 255, (80, 72, 3), dtype=np.uint8)).save(src / f"im{i}.png")
    out = infer.main(["rec", "--img_dir", str(src), "--out", str(tmp_path / "o3")] + common)
    assert [p.split("/")[-1] for p in out] == ["00000_realimg.png", "00000_mtv_rec.png", "00001_realimg.png", "00001_mtv_rec.png"]
    # same inputs through the library: identical pixels (eval-mode models, deterministic forward)
    from dge_amd.e_align import EAlignStep
    G.eval(); E.eval()
    st = EAlignStep(G, E, None, batch_size=2)
    imgs = infer.load_images([str(src / "im0.png"), str(src / "im1.png")], 64)
    from dge_amd.e_align import set_seed
    set_seed(4)                                               # the entry point's default --seed: same encoder noise draws
    _, rec = infer.reconstruct_images(st, imgs)
    want = ((rec[1:2].float().cpu() * 0.5 + 0.5).clamp(0, 1)[0].permute(1, 2, 0) * 255.0 + 0.5).to(torch.uint8).numpy()
    got = np.asarray(Image.open(out[3]))[2:-2, 2:-2]
    # bf16 forward; the instance-norm statistics end in f32 atomics whose order differs run to run: a few 8-bit levels
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 8 and d.mean() < 0.5, (d.max(), d.mean())
