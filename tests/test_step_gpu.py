"""One full two-phase E_align_s2 iteration on the HIP path against the REFERENCE's own run of
the same iteration (tests/golden/step_s2.npz from tools/gen_golden.py): losses, w2, images and
encoder parameters after each optimiser phase, including quirk Q3 (second backward with the
already-updated weights) and Q1 (G in train mode)."""
import numpy as np
import pytest
import torch

from tests.conftest import golden
from tests.golden import recipe as R
from tests.helpers import s2_shapes, enc_shapes
from oracle import ref_torch as O
from oracle import lpips_ref as LR

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = a.detach().float().cpu()
    b = torch.as_tensor(np.asarray(b)).float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def test_two_phase_step_matches_reference_run():
    import dge_amd
    from dge_amd.encoder import BE
    from dge_amd.lpips import LPIPS
    from dge_amd.e_align import EAlignStep
    g = golden("step_s2.npz")
    G = dge_amd.StyleGAN2Generator(64, fmaps_base=2048, fmaps_max=128, compute_dtype="f32").cuda()
    G.load_state_dict(R.fill_s2(s2_shapes(64, fmaps_base=2048, fmaps_max=128), seed=11))
    G.train()
    for p in G.parameters():
        p.requires_grad_(False)
    E = BE(startf=16, maxf=64, layer_count=5, compute_dtype="f32").cuda()
    E.load_state_dict(R.fill_encoder(enc_shapes(16, 64, 5), seed=31))
    LP = LPIPS(compute_dtype="f32").cuda()
    LP.load_state_dict(LR.seeded_params(0))
    st = EAlignStep(G, E, LP, lr=0.0015, batch_size=2)
    new_z = R.randn("step.new_z", (2, 512), 1).cuda()
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: new_z.clone()
    try:
        for it in range(2):
            z = R.randn(f"step.z{it}", (2, 512), 1)
            noises = [R.randn(f"step.it{it}.noise{i}", s, 1).cuda() for i, s in enumerate(O.enc_noise_shapes(5, 2, 64))]
            r = st.step(it, z=z, noises=noises)
            assert relerr(r["w1"], g[f"it{it}_w1"]) < 1e-4
            assert relerr(r["imgs1"], g[f"it{it}_imgs1"]) < 5e-4
            assert relerr(r["w2"], g[f"it{it}_w2"]) < 2e-3
            assert relerr(r["imgs2"], g[f"it{it}_imgs2"]) < 2e-3
            ref_l = g[f"it{it}_losses"]
            info = r["info_img"].cpu().numpy()
            got = [float(r["loss_tsa"]), info[0, 0], info[1, 0], info[2, 0], float(r["loss_w"])]
            for a, b in zip(got, ref_l):
                assert abs(a - b) < 2e-3 * abs(b), (it, got, ref_l)
            ref_info = g[f"it{it}_info"]
            for row in range(3):
                for col in (0, 4, 5, 6):            # mse, cos, ssim, lpips
                    assert abs(info[row, 1 + col] - ref_info[row, col]) < 3e-3 * abs(ref_info[row, col]) + 1e-6, (it, row, col)
            sd = E.state_dict()
            for key in g.files:
                if key.startswith(f"it{it}_after_phase2:"):
                    k = key.split(":", 1)[1]
                    # parameter change of one step is ~lr*coef: compare the UPDATE, not the value
                    before = R.fill_encoder(enc_shapes(16, 64, 5), seed=31)[k] if it == 0 else None
                    e = relerr(sd[k], g[key])
                    assert e < 1e-4, (it, k, e)
                    if before is not None:
                        du_ref = torch.as_tensor(g[key]) - before
                        du = sd[k].cpu() - before
                        if du_ref.abs().max() > 0:
                            assert ((du - du_ref).abs().max() / du_ref.abs().max()).item() < 0.05, (it, k)
            assert abs(R.checksum({k: v.cpu() for k, v in sd.items()}) - float(g[f"it{it}_param_checksum"])) < 1e-5 * float(g[f"it{it}_param_checksum"])
            assert relerr(G.truncation.w_avg, g[f"it{it}_w_avg"]) < 1e-5
    finally:
        torch.randn_like = orig
