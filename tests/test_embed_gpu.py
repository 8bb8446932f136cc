"""Real-image inversion loop (reference embedding_img.py:84-127, BASELINE config 5 at reduced size) on the HIP path against
two iterations of the reference's own loop body (tests/golden/embed_sg1.npz, tools/gen_golden.py `embed`): StyleGAN1
synthesis + E_Blur, gradients through both encoder outputs, through the second encoder call's input image and through Gs;
quirk Q3 (second backward with already-updated weights) included."""
import numpy as np
import pytest
import torch

from tests.conftest import golden
from tests.golden import recipe as R
from oracle import lpips_ref as LR

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = a.detach().float().cpu()
    b = torch.as_tensor(np.asarray(b)).float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def l2rel(a, b):
    a = a.detach().float().cpu().flatten(); b = torch.as_tensor(np.asarray(b)).float().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_embedding_loop_matches_reference_run():
    import dge_amd.stylegan1 as S
    from dge_amd.encoder_variants import BlurBE
    from dge_amd.lpips import LPIPS
    from dge_amd.embedding import EmbedStep
    from tests.test_sg1 import sg1_shapes
    g = golden("embed_sg1.npz")
    L = 5
    Gs = S.Generator(startf=16, maxf=64, layer_count=L, latent_size=512, compute_dtype="f32").cuda()
    shapes = sg1_shapes(16, 64, L)
    sd = R.fill_encoder(shapes, seed=43)
    blur = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16.0
    for k in sd:
        if k.endswith("blur.weight"):
            sd[k] = blur.view(1, 1, 3, 3).repeat(shapes[k][0], 1, 1, 1)
    sd["const"] = R.randn("sg1step.const", tuple(shapes["const"]), 43)
    Gs.load_state_dict(sd)
    for p in Gs.parameters():
        p.requires_grad_(False)
    E = BlurBE(startf=16, maxf=64, layer_count=L, compute_dtype="f32").cuda()
    esd = R.fill_encoder({k: list(v.shape) for k, v in E.state_dict().items()}, seed=71)
    for k in esd:
        if k.endswith("blur.weight"):
            esd[k] = E.state_dict()[k].clone()
    E.load_state_dict(esd)
    LP = LPIPS(compute_dtype="f32").cuda()
    LP.load_state_dict(LR.seeded_params(0))
    st = EmbedStep(Gs, E, LP, lr=0.01)
    st.begin_image()
    imgs1 = torch.as_tensor(g["imgs1"]).cuda()
    nshapes = [tuple(s) for s in g["noise_shapes"].tolist()]
    assert len(nshapes) == 9 + 10 + 9
    hooks = {}
    for it in range(2):
        nz = [R.randn(f"embed.it{it}.noise{i}", s, 2) for i, s in enumerate(nshapes)]
        grads1 = {}
        # capture the phase-1 gradients before the optimizer consumes them
        orig_step = st.opt.step
        calls = []

        def spy(*a, **kw):
            calls.append({k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in E.named_parameters()})
            return orig_step(*a, **kw)
        st.opt.step = spy
        r = st.step(imgs1, noises=([n.cuda() for n in nz[:9]], nz[9:19], [n.cuda() for n in nz[19:]]))
        st.opt.step = orig_step
        assert relerr(r["w1"], g[f"it{it}_w1"]) < 1e-3
        assert relerr(r["imgs2"], g[f"it{it}_imgs2"]) < 2e-3
        assert relerr(r["w2"], g[f"it{it}_w2"]) < 2e-3
        assert relerr(r["const2"], g[f"it{it}_const2"]) < 1e-3 and relerr(r["const3"], g[f"it{it}_const3"]) < 2e-3
        ref_l = g[f"it{it}_losses"]           # msiv, imgs, medium, small, w, c1
        info = r["info_img"].cpu().numpy()
        got = [float(r["loss_msiv"]), info[0, 0], info[1, 0], info[2, 0], float(r["loss_w"]), float(r["loss_c1"])]
        for a, b in zip(got, ref_l):
            assert abs(a - b) < 3e-3 * abs(b), (it, got, ref_l)
        for key in g.files:
            if key.startswith(f"it{it}_grad1:") or key.startswith(f"it{it}_grad2:"):
                phase = 0 if "_grad1:" in key else 1
                k = key.split(":", 1)[1]
                e = l2rel(calls[phase][k], g[key])
                print("graderr", it, phase, k, f"{e:.4f}")
                # phase 2 runs with the weights already moved by a sign-like first Adam step (lr 0.01): elements whose
                # phase-1 gradient is within rounding of zero step the other way, which perturbs the phase-2 gradient
                assert e < (5e-3 if phase == 0 and it == 0 else 6e-2), (it, phase, k, e)
        ck = R.checksum({k: v.cpu() for k, v in E.state_dict().items() if not k.endswith("blur.weight")})
        assert abs(ck - float(g[f"it{it}_param_checksum"])) < 2e-4 * float(g[f"it{it}_param_checksum"])


def test_graph_replay_equals_eager_iterations():
    """The captured hipGraph of one inversion iteration (EmbedStep.capture / replay) reproduces eager iterations: same
    static noise, same start, 4 iterations each way; encoder parameters and outputs agree to atomics-order noise.  Exercises
    the device-side Adam step factor (the only host-varying quantity of an iteration)."""
    import dge_amd.stylegan1 as S
    from dge_amd.encoder_variants import BlurBE
    from dge_amd.lpips import LPIPS
    from dge_amd.embedding import EmbedStep
    torch.manual_seed(0)
    L = 5

    def make():
        torch.manual_seed(1)
        Gs = S.Generator(startf=16, maxf=64, layer_count=L, latent_size=512, compute_dtype="f32").cuda()
        for p in Gs.parameters():
            p.requires_grad_(False)
        E = BlurBE(startf=16, maxf=64, layer_count=L, compute_dtype="f32").cuda()
        LP = LPIPS(compute_dtype="f32").cuda()
        LP.load_state_dict(LR.seeded_params(0))
        return EmbedStep(Gs, E, LP, lr=0.002)

    g = golden("embed_sg1.npz")
    nshapes = [tuple(s) for s in g["noise_shapes"].tolist()]
    nz = [R.randn(f"embed.it0.noise{i}", s, 2).cuda() for i, s in enumerate(nshapes)]
    noises = (nz[:9], nz[9:19], nz[19:])
    imgs1 = torch.as_tensor(g["imgs1"]).cuda()
    a = make(); a.begin_image()
    wa = [a.step(imgs1, noises)["w1"].clone() for _ in range(4)]
    b = make()
    b.capture(imgs1, noises, warmup=1)       # runs 2 iterations (1 warm-up + the captured one) from the same start ...
    b.begin_image()                          # ... then restart from the checkpoint with fresh optimizer state, graph intact
    wb = [b.replay()["w1"].clone() for _ in range(4)]
    torch.cuda.synchronize()
    # First iteration: identical up to atomics order.  Later ones: beta1 = 0 Adam is sign-like in its first steps, so
    # rounding-level differences of near-zero gradients flip update signs; two EAGER runs already differ by 2-3 % in w1
    # from the second iteration on (tools/probes/dbg_graph.py), the replayed graph stays inside that band.
    assert relerr(wb[0], wa[0].cpu().numpy()) < 1e-4
    for i in range(1, 4):
        assert relerr(wb[i], wa[i].cpu().numpy()) < 8e-2, i
    for (k, pa), (_, pb) in zip(a.E.named_parameters(), b.E.named_parameters()):
        # 8 optimiser steps of at most lr * 10 each (sign-like): parameters may differ by a few such steps
        assert float((pb.detach() - pa.detach()).abs().max()) < 0.02, k


def test_invert_second_image_group_is_inverted_against_its_own_image():
    """embedding.invert() over TWO image groups with the hipGraph launch (embedding_img.py:74-84: the loop re-loads the encoder and
    runs on the next image): the captured iteration reads a static input buffer, so the second group has to be copied into it
    (EmbedStep.set_image).  One iteration from the checkpoint: w1 = E_ckpt(image), so the replayed result must equal the eager
    one on the SAME image and differ from the first image's."""
    import dge_amd.stylegan1 as S
    from dge_amd.encoder_variants import BlurBE
    from dge_amd.lpips import LPIPS
    from dge_amd.embedding import EmbedStep, invert
    L = 5

    def make():
        torch.manual_seed(1)
        Gs = S.Generator(startf=16, maxf=64, layer_count=L, latent_size=512, compute_dtype="f32").cuda()
        for p in Gs.parameters():
            p.requires_grad_(False)
        E = BlurBE(startf=16, maxf=64, layer_count=L, compute_dtype="f32").cuda()
        LP = LPIPS(compute_dtype="f32").cuda()
        LP.load_state_dict(LR.seeded_params(0))
        return EmbedStep(Gs, E, LP, lr=0.002)

    g = golden("embed_sg1.npz")
    img_a = torch.as_tensor(g["imgs1"]).cuda()
    img_b = (R.randn("embed.second_image", tuple(img_a.shape), 3, 0.4)).cuda().clamp(-1, 1)
    e = make()
    wa_e = invert(e, img_a, iterations=1, launch="eager")["w1"].clone()
    wb_e = invert(e, img_b, iterations=1, launch="eager")["w1"].clone()
    s = make()
    wa_g = invert(s, img_a, iterations=2, launch="graph")["w1"].clone()      # captures (1 warm-up iteration) + 1 replay
    wb_g = invert(s, img_b, iterations=1, launch="graph")["w1"].clone()      # replay on the second image from the checkpoint
    torch.cuda.synchronize()
    assert relerr(wb_e, wa_e.cpu().numpy()) > 1e-2                           # the two images give different codes ...
    assert relerr(wb_g, wb_e.cpu().numpy()) < 1e-4, relerr(wb_g, wb_e.cpu().numpy())      # ... and the replay follows the image
    with pytest.raises(ValueError):
        s.set_image(img_b[:, :, :16])
