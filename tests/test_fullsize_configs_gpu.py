"""Full-size evidence for BASELINE configs 1, 2, 4, 5 (tests/test_fullsize_gpu.py and the full-size step test cover config 3).

`launch_t` / the streaming dispatch pick the kernel instantiation from the LAUNCH SHAPE, so the golden-size tests of these
configs (32^2 .. 128^2) never reach the instantiations their benchmarks run (profiles/*other_configs*: StyleGAN1 at 256^2 batch
8 / 32 and at 1024^2, PGGAN-256, BigGAN-deep-256, E_Blur at 1024^2 batch 1).  Here every module of those configs runs at its
benchmarked shape and batch; the result is compared with the CPU oracle (oracle/ref_torch.py, pinned on the reference's own
outputs in the per-module tests) on samples 0 and B-1, the encoder gradients with the oracle's autograd (the loss weights only
those two samples, so the batch-summed parameter gradients are comparable), and the set of conv-family instantiations the run
selected (dge_last_kernel after every launch) is asserted BY NAME against the list in the test: a dispatch change shows up here.
bf16 bounds = 2x the values measured on MI355X (in the comments)."""
import math

import numpy as np
import pytest
from tests.conftest import meas
import torch

from tests.golden import recipe as R
from tests.helpers import enc_shapes
from oracle import ref_torch as O

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


class Census:
    """records dge_last_kernel() after every conv-family entry point of dge_amd.ops"""
    NAMES = ("conv2d", "conv_pp", "upconv_fir", "conv_wgrad")

    def __enter__(self):
        from dge_amd import ops
        from dge_amd._lib import last_kernel
        self.ops, self.seen, self.orig = ops, set(), {}
        for n in self.NAMES:
            f = getattr(ops, n)
            self.orig[n] = f

            def wrap(*a, _f=f, **k):
                r = _f(*a, **k)
                self.seen.add(last_kernel())
                return r
            setattr(ops, n, wrap)
        return self

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(self.ops, n, f)

    def check(self, expected):
        assert self.seen == set(expected), ("new", sorted(self.seen - set(expected)), "gone", sorted(set(expected) - self.seen))


def _two(t, B):
    return torch.cat([t[:1], t[B - 1:B]]) if B > 1 else t[:1]


def _grad_report(named, ref):
    cos_min, l2_max, worst = 1.0, 0.0, None
    for k, gr in ref.items():
        if gr is None:
            continue
        g = named[k].detach().float().cpu().reshape(-1).double(); r = gr.reshape(-1).double()
        if r.norm().item() == 0:
            continue
        cos_min = min(cos_min, (g @ r).item() / (g.norm().item() * r.norm().item() + 1e-300))
        l2 = ((g - r).norm() / r.norm()).item()
        if l2 > l2_max:
            l2_max, worst = l2, k
    print("worst gradient tensor:", worst)
    return cos_min, l2_max


# --------------------------------------------------------------------------------------------- StyleGAN1 (configs 2 and 5)
SG1_KERNELS = {
    (64, 7, 8): ["conv_igemm<bf16,16,16,128,32,3,2,2>", "conv_igemm<bf16,16,16,64,32,3,4,1>", "conv_igemm<bf16,16,16,64,32,3,4,1>+tr",
                 "conv_igemm<bf16,8,8,64,128,3,2,2>", "conv_small<bf16,8,8,64,512>"],
    (64, 7, 32): ["conv_igemm<bf16,16,16,128,32,3,2,2>", "conv_igemm<bf16,16,16,128,32,3,2,2>+tr", "conv_igemm<bf16,16,16,64,32,3,4,1>",
                  "conv_igemm<bf16,16,16,64,32,3,4,1>+tr", "conv_igemm<bf16,8,8,64,128,3,2,2>", "conv_small<bf16,8,8,64,512>"],
    (16, 9, 1): ["conv_igemm<bf16,16,16,32,32,3,4,1>", "conv_igemm<bf16,16,16,64,32,3,4,1>", "conv_igemm<bf16,8,8,64,128,3,2,2>",
                 "conv_stream<bf16,16,16,enc_stats>", "conv_small<bf16,8,8,64,512>"],
}
IG = "conv_igemm<bf16,"
PG_KERNELS = [IG + "16,16,128,32,3,2,2>+tr", IG + "16,16,64,32,3,4,1>+tr", IG + "8,8,64,128,3,2,2>", "conv_stream<bf16,64,64,gen>"]
BG_KERNELS = [IG + "16,16,128,32,1,2,2>", IG + "16,16,128,32,3,2,2>+tr", IG + "16,16,32,32,3,4,1>+tr", IG + "16,16,64,32,1,4,1>",
              IG + "16,16,64,32,3,4,1>+tr", IG + "8,8,64,128,1,2,2>", IG + "8,8,64,128,3,2,2>"]
BE256_KERNELS = {
    8: [IG + "16,16,128,32,1,2,2>", IG + "16,16,128,32,3,2,2>", IG + "16,16,128,32,3,2,2>+tr", IG + "16,16,64,32,1,4,1>", IG + "16,16,64,32,3,4,1>",
        IG + "16,16,64,32,3,4,1>+tr", IG + "8,8,64,128,1,2,2>", "conv_small<bf16,8,8,64,512>", "conv_stream<bf16,64,64,dot>", "conv_wgrad_tr<1,16>",
        "conv_wgrad_tr<3,16>", "conv_wgrad_tr<3,8>", "wgrad_dma<16,64,64,2>"],
    32: [IG + "16,16,128,32,1,2,2>", IG + "16,16,128,32,3,2,2>", IG + "16,16,128,32,3,2,2>+tr", IG + "16,16,64,32,1,4,1>", IG + "16,16,64,32,3,4,1>",
         IG + "16,16,64,32,3,4,1>+tr", "conv_pw<bf16,64,128>", "conv_small<bf16,8,8,64,512>", "conv_stream<bf16,64,64,dot>", "conv_wgrad_tr<1,16>",
         "conv_wgrad_tr<3,16>", "conv_wgrad_tr<3,8>", "wgrad_dma<16,64,64,2>"],
}
BLUR1024_KERNELS = [IG + "16,16,32,32,1,4,1>", IG + "16,16,32,32,3,4,1>", IG + "16,16,64,32,1,4,1>", IG + "16,16,64,32,3,4,1>", IG + "16,16,64,32,3,4,1>+tr",
                    IG + "8,8,64,128,1,2,2>", IG + "8,8,64,128,3,2,2>", IG + "8,8,64,64,1,2,2>", "conv_pw<bf16,16,32>", "conv_pw<bf16,32,32>",
                    "conv_stream<bf16,16,16,dot>", "conv_stream<bf16,16,16,enc_stats>", "conv_stream<bf16,16,32,gen>", "conv_stream<bf16,32,16,gen>",
                    "conv_stream<bf16,32,32,dot>", "conv_stream<bf16,32,64,gen>", "conv_stream<bf16,64,32,gen>", "conv_stream<bf16,64,64,dot>",
                    "conv_wgrad_tr<1,16>", "conv_wgrad_tr<3,16>", "conv_wgrad_tr<3,8>", "wgrad_dma<16,32,32,3>", "wgrad_dma<16,64,32,2>", "wgrad_dma<16,64,64,2>", "wgrad_dma<8,64,64,3>",
                    "conv_small<bf16,8,8,64,512>"]


@pytest.mark.parametrize("startf,L,B", [(64, 7, 8), (64, 7, 32), (16, 9, 1)])
def test_stylegan1_generator_fullsize(startf, L, B):
    """configs 2 (Cat-256: startf 64, 7 blocks, batch 8 and the ablation batch 32) and 5 (FFHQ-1024: startf 16, 9 blocks, batch 1):
    Generator.decode (model/stylegan1/net.py:331-336) incl. the fused ConvTranspose2d up blocks and the 16-channel block at 1024^2"""
    import dge_amd.stylegan1 as S
    from tests.test_sg1 import sg1_shapes
    shapes = sg1_shapes(startf, 512, L)
    P = R.fill_encoder(shapes, seed=141)
    blur = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16.0
    for k in P:
        if k.endswith("blur.weight"):
            P[k] = blur.view(1, 1, 3, 3).repeat(shapes[k][0], 1, 1, 1)
    P["const"] = R.randn("sg1f.const", tuple(shapes["const"]), 141)
    G = S.Generator(startf=startf, maxf=512, layer_count=L, latent_size=512, compute_dtype="bf16").cuda()
    G.load_state_dict(P)
    styles = R.randn("sg1f.styles", (B, 2 * L, 512), 6)
    noises = []
    for i in range(L):
        noises += [R.randn(f"sg1f.noise{2 * i}", (B, 1, 4 << i, 4 << i), 6), R.randn(f"sg1f.noise{2 * i + 1}", (B, 1, 4 << i, 4 << i), 6)]
    with Census() as c, torch.no_grad():
        img = G.forward(styles.cuda(), L - 1, noises=[n.cuda() for n in noises])
    ref = O.sg1_generator(P, _two(styles, B), L - 1, [_two(n, B) for n in noises])
    e = relerr(_two(img, B), ref)
    print(f"StyleGAN1 startf={startf} L={L} B={B}: image err {e:.3e}", sorted(c.seen))
    assert e < (0.11 if L == 7 else 0.15), e      # measured 5.4e-2 / 4.8e-2 (14 instance norms deep), 7.5e-2 (18 deep, 1024^2): bf16 storage
    c.check(SG1_KERNELS[(startf, L, B)])


# --------------------------------------------------------------------------------------------- PGGAN-256 (configs 1 / 3)
def test_pggan_generator_fullsize():
    """PGGANGenerator.forward at 256^2, batch 8 (model/pggan/pggan_generator.py:154-204)"""
    from dge_amd.pggan_generator import PGGANGenerator
    from tests.test_pggan import pg_shapes
    B = 8
    P = {k: (R.randn("pgf." + k, tuple(v), 151, 0.2 if k.endswith("bias") else 1.0) if len(v) else torch.zeros(())) for k, v in pg_shapes(256).items()}
    G = PGGANGenerator(256, compute_dtype="bf16").cuda()
    G.load_state_dict(P)
    z = R.randn("pgf.z", (B, 512), 151)
    with Census() as c, torch.no_grad():
        img = G(z.cuda())["image"]
    ref = O.pg_generator(P, _two(z, B))
    e = relerr(_two(img, B), ref)
    print(f"PGGAN-256 B={B}: image err {e:.3e}", sorted(c.seen))
    assert e < 4e-2, e                  # measured 1.8e-2
    c.check(PG_KERNELS)


# --------------------------------------------------------------------------------------------- BigGAN-deep-256 (config 4)
def test_biggan_generator_fullsize():
    """BigGAN.forward, biggan-deep-256 configuration, batch 8, truncation 0.4 (model/biggan_generator.py:232-256,296-304)"""
    from dge_amd.biggan_generator import BigGAN, BigGANConfig
    from tests.test_biggan import DEEP256
    B = 8
    G = BigGAN(BigGANConfig.from_dict(DEEP256), compute_dtype="bf16").cuda()
    P = R.fill_biggan({n: list(v.shape) for n, v in G.state_dict().items()}, 171)
    G.load_state_dict(P)
    G.eval()
    z = R.randn("bgf.z", (B, 128), 171, 0.4)
    onehot = torch.zeros(B, 1000); onehot[:, 207] = 1.0
    with Census() as c, torch.no_grad():
        img, cond = G(z.cuda(), onehot.cuda(), 0.4)
    ref, rcond = O.bg_generator(P, DEEP256, _two(z, B), _two(onehot, B), 0.4)
    e = relerr(_two(img, B), ref)
    print(f"BigGAN-deep-256 B={B}: image err {e:.3e}", sorted(c.seen))
    assert relerr(_two(cond, B), rcond) < 1e-6
    assert e < 2.1e-2, e                # measured 1.0e-2
    c.check(BG_KERNELS)


# --------------------------------------------------------------------------------------------- encoders
def _enc_case(E, P, img, noises, fwd_ref, B, tag, tol_out, tol_l2, extra_in=None, only_w=False):
    """forward on the whole batch, loss weights only samples 0 and B-1 -> parameter gradients comparable with the 2-sample oracle"""
    E.load_state_dict(P)
    with Census() as c:
        out = E(img.cuda(), *([extra_in.cuda()] if extra_in is not None else []), noises=[n.cuda() for n in noises])
        outs = [o for o in out if torch.is_tensor(o) and o.dim() > 0 and o.requires_grad]
        if only_w:                        # E.BE: the E_align losses back-propagate through w only (E_align_s2.py:203-221)
            outs = outs[-1:]
        gws = [R.randn(f"{tag}.g{i}", tuple(o.shape), 9, 0.05) for i, o in enumerate(outs)]
        for g in gws:
            if B > 2:
                g[1:B - 1] = 0
        sum((o * g.cuda()).sum() for o, g in zip(outs, gws)).backward()
    Pr = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("blur.weight")) for k, v in P.items()}
    rout = fwd_ref(Pr, _two(img, B), [_two(n, B) for n in noises])
    routs = [o for o in rout if torch.is_tensor(o) and o.dim() > 0 and o.requires_grad]
    if only_w:
        routs = routs[-1:]
    assert len(routs) == len(outs)
    for i, (o, r) in enumerate(zip(outs, routs)):
        e = relerr(_two(o, B), r)
        meas(f"fullsize_cfg_out[{tag}-{B}-{i}]", err=e, bound=tol_out)
        assert e < tol_out, (tag, e)
    sum((r * _two(g, B)).sum() for r, g in zip(routs, gws)).backward()
    cos_min, l2_max = _grad_report({k: p.grad for k, p in E.named_parameters() if p.grad is not None},
                                   {k: v.grad for k, v in Pr.items() if v.requires_grad and v.grad is not None})
    print(f"{tag} B={B}: grad cos_min {cos_min:.4f} l2_max {l2_max:.3e}", sorted(c.seen))
    meas(f"fullsize_cfg_grad[{tag}-{B}]", cos_min=cos_min, l2_max=l2_max, bound_l2=tol_l2)
    assert cos_min > 1 - tol_l2 / 2.5 and l2_max < tol_l2, (tag, cos_min, l2_max)
    return c


@pytest.mark.parametrize("B", [8, 32])
def test_encoder_be256_fullsize(B):
    """config 2: E.BE(startf=64, layer_count=7) at 256^2, batch 8 / 32 (model/E/E.py:50-85,122-136), forward + every parameter gradient"""
    from dge_amd.encoder import BE
    L, S = 7, 256
    P = R.fill_encoder(enc_shapes(64, 512, L), seed=121)
    img = R.randn("be256.img", (B, 3, S, S), 9, 0.5)
    noises = [R.randn(f"be256.noise{i}", s, 9) for i, s in enumerate(O.enc_noise_shapes(L, B, S))]
    E = BE(startf=64, maxf=512, layer_count=L, compute_dtype="bf16").cuda()
    # worst per-tensor gradient error: decode_block.1.noise_weight_2 (a [1,C,1,1] sum of g * noise over 8 Mpixel: heavy cancellation in
    # bf16-stored operands), measured L2 0.28 / cosine 0.961; the f32 run of tests/test_enc_gpu.py is the parity check of the formulas
    c = _enc_case(E, P, img, noises, lambda Pr, im, nz: O.enc_forward(Pr, im, nz), B, "E.BE-256", 5e-2, 0.56, only_w=True)
    c.check(BE256_KERNELS[B])


def test_encoder_blur1024_fullsize():
    """config 5: E_Blur.BE(startf=16, layer_count=9) at 1024^2, batch 1 (model/E/E_Blur.py:50-85,88-135: blur + stride-2
    transform_kernel convs down to 128^2), forward + every parameter gradient, gradients entering through both outputs"""
    from dge_amd.encoder_variants import BlurBE
    L, S, B = 9, 1024, 1
    E = BlurBE(startf=16, maxf=512, layer_count=L, compute_dtype="bf16").cuda()
    P = R.fill_encoder({k: list(v.shape) for k, v in E.state_dict().items()}, seed=161)
    for k in P:
        if k.endswith("blur.weight"):
            P[k] = E.state_dict()[k].detach().cpu().clone()
    fused = [bool(b.fused_scale) for b in E.decode_block]
    img = R.randn("eb1024.img", (B, 3, S, S), 61, 0.5)
    shapes = []
    for j in range(L):
        r = S >> j
        shapes.append((B, 1, r, r))
        if j != L - 1:
            shapes.append((B, 1, r // 2, r // 2) if fused[j] else (B, 1, r, r))
    noises = [R.randn(f"eb1024.noise{i}", s, 61) for i, s in enumerate(shapes)]
    # measured: worst tensor decode_block.0.noise_weight_1, L2 0.38 / cosine 0.93 (tests/test_encvar.py holds the f32 run of the same formulas)
    # (outputs: measured 0.04 - 0.058 of max over repeated runs - batch 1, nine bf16 instance norms in a row, and the f32 atomics order of
    #  their statistics differs from run to run; the f32 run of the same formulas is in tests/test_encvar.py)
    c = _enc_case(E, P, img, noises, lambda Pr, im, nz: O.enc_blur_forward(Pr, im, nz, fused), B, "E_Blur-1024", 8e-2, 0.75)
    c.check(BLUR1024_KERNELS)


def test_benchmarked_step_runs_on_the_round4_kernel_forms():
    """config 3 as bench.py runs it (StyleGAN2-1024, E.BE startf 16, bf16, batch 8): the kernels this round put on the path are the ones
    that run - asserted by name (dge_last_kernel) over one complete two-phase step.  Values: tests/test_step_gpu.py (same step
    against the CPU oracle) and the per-launch suites."""
    from dge_amd import ops
    from dge_amd.e_align import EAlignStep, build_models
    if ops.is_deterministic():
        pytest.skip("deterministic mode keeps the atomics-free kernel forms")
    dev = torch.device("cuda", 0)
    G, E, LP = build_models(1024, 16, "bf16", dev, seed=0)
    G.train()
    st = EAlignStep(G, E, LP, batch_size=8)
    st.step(0)                                   # (fills the pack caches; the first LPIPS call stays on one stream)
    Census.NAMES = Census.NAMES + ("conv_wgrad_dots",)
    try:
        with Census() as c:
            st.step(1)
    finally:
        Census.NAMES = Census.NAMES[:-1]
    torch.cuda.synchronize()
    want = {"conv_pp<bf16,16,32,128>",                                                   # generator layers 8 / 10 / 12, LPIPS conv2_x / conv3_x
            "conv_pp<bf16,16,32,128>+dg+prep", "conv_pp<bf16,16,32,128>+dg+s2d+prep",    # synthesis backward: layers 12 / 10 / 8, 13
            "conv_pp<bf16,16,32,128>+dg+t2d+prep",                                       # ... 11 / 9 in phase form
            "conv_stream<bf16,32,16,dot_in>", "conv_stream<bf16,64,32,dot_in>",          # encoder backward: conv_2 of blocks 0 / 1
            "conv_stream<bf16,32,32,dot_inx>", "conv_stream<bf16,64,64,dot_inx>",        # conv_1 of blocks 1 / 2
            "conv_stream<bf16,16,16,dot_fromrgb>"}                                       # conv_1 of block 0 + FromRGB gradients
    assert want <= c.seen, sorted(want - c.seen)
    gone = {"conv_stream<bf16,16,16,dot>", "conv_stream<bf16,32,32,dot>", "conv_stream<bf16,32,16,dot>", "conv_stream<bf16,64,32,dot>"}
    assert not (gone & c.seen), sorted(gone & c.seen)
