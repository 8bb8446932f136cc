"""Full-size parity of the NON-conv kernels of the benchmarked step (BASELINE config 3: StyleGAN2-1024 + E.BE(startf=16),
batch 8, bf16): the streaming kernels around the convolutions are a third of the step and their grid-stride loops, per-sample
partial-sum flushes and > 2^28-element index paths are only reached at the benchmark's shapes.  Each kernel runs at its true
shape on the device and is compared with oracle/elem_ref.py (forward lines of the reference + torch.autograd, pinned in
tests/test_oracle_golden.py::test_elem_ref_*) on samples 0 and B-1 for the element-wise outputs and on EVERY sample for the
per-(sample, channel) reductions.

Bounds (stated at each assert): element-wise bf16 outputs within ONE bf16 rounding of the oracle's f64 value
(|y - ref| <= 2^-8 |ref| + 1e-5 max|ref|); f32 images 2e-5 of max; reductions 5e-5 of the sum of absolute terms (f32
accumulation in a different order)."""
import math

import pytest
import torch

from oracle import elem_ref as ER

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gen(seed):
    return torch.Generator(device=DEV).manual_seed(seed)


def _act(B, H, W, C, g, scale=1.0):
    return (torch.randn(B, H, W, C, device=DEV, generator=g) * scale).to(torch.bfloat16)


def _nchw(x_nhwc, b):
    return x_nhwc[b:b + 1].float().permute(0, 3, 1, 2).contiguous().cpu()


def _one_rounding(got, ref, slack=1e-5):
    ref = ref.double()
    return ((got.double() - ref).abs() - (2.0 ** -8) * ref.abs() - slack * ref.abs().max()).max().item()


def _red_err(got, ref, absum):
    """max |got - ref| / sum|terms| over channels"""
    return ((got.double().cpu() - ref.double()).abs() / (absum.double() + 1e-30)).max().item()


SAMPLES = lambda B: sorted({0, B - 1})


@pytest.mark.parametrize("C,R,B", [(32, 1024, 8), (64, 512, 8), (512, 16, 8)])
def test_modconv_bwd_prep_fullsize(C, R, B):
    """backward through noise / bias / lrelu*sqrt(2) / demodulation (stylegan2_generator.py:908-921) with the three per-(b,c)
    sums of the demodulation gradient - dge_modconv_bwd_prep"""
    from dge_amd import ops
    g = _gen(100 + C)
    gain = math.sqrt(2.0)
    x = _act(B, R, R, C, g, 1.5)
    gx = _act(B, R, R, C, g)
    d = 0.5 + torch.rand(B, C, device=DEV, generator=g)
    noise = torch.randn(1, R, R, device=DEV, generator=g)
    Rs = torch.zeros(B, C, 3, device=DEV)
    gy = ops.modconv_bwd_prep(gx, x, d, noise, gain, Rs)
    for b in range(B):
        ref_gy, ref_R = ER.modconv_tail_bwd(_nchw(x, b), _nchw(gx, b), d[b].cpu(), noise[0].cpu(), gain)
        if b in SAMPLES(B):
            assert _one_rounding(_nchw(gy, b), ref_gy) <= 0, b
        gz = ref_gy / d[b].cpu().double()[None, :, None, None]
        z = ER.lrelu_inverse(_nchw(x, b).double(), gain)
        absum = torch.stack([(gz * z).abs().sum((0, 2, 3)), (gz * noise[0].cpu().double()).abs().sum((0, 2, 3)), gz.abs().sum((0, 2, 3))], 1)
        assert _red_err(Rs[b], ref_R, absum) < 5e-5, b


@pytest.mark.parametrize("C,R,B,prev", [(32, 1024, 8, True), (64, 512, 8, True), (512, 8, 8, True), (512, 4, 8, False)])
def test_torgb_and_its_adjoint_fullsize(C, R, B, prev):
    """toRGB + skip upsample (stylegan2_generator.py:515-522, :465-474, :603-615) - dge_torgb (both kernels: one thread / one wave
    per pixel), dge_torgb_bwd, dge_up2_bwd"""
    from dge_amd import ops
    g = _gen(200 + C)
    x = _act(B, R, R, C, g)
    wrgb = torch.randn(3, C, 1, 1, device=DEV, generator=g)
    s = 1.0 + 0.3 * torch.randn(B, C, device=DEV, generator=g)
    bias = 0.1 * torch.randn(3, device=DEV, generator=g)
    pv = torch.randn(B, 3, R // 2, R // 2, device=DEV, generator=g) if prev else None
    wscale = 1.0 / math.sqrt(C)
    img = ops.torgb(x, wrgb, s, bias, pv, wscale)
    gimg = torch.randn(B, 3, R, R, device=DEV, generator=g)
    gxd, gs = ops.torgb_bwd(gimg, x, wrgb.reshape(3, -1), s, wscale)
    for b in range(B):
        xb = _nchw(x, b)
        if b in SAMPLES(B):
            ref = ER.torgb(xb.double(), wrgb.reshape(3, C).cpu().double(), s[b].cpu().double(), bias.cpu().double(), wscale,
                           pv[b:b + 1].cpu() if prev else None)
            e = ((img[b:b + 1].cpu().double() - ref).abs().max() / ref.abs().max()).item()
            assert e < 2e-5, (b, e)                       # f32 image, f32 accumulation over C channels
        ref_gx, ref_gs = ER.torgb_bwd(xb, wrgb.reshape(3, C).cpu(), s[b].cpu(), wscale, gimg[b:b + 1].cpu())
        if b in SAMPLES(B):
            assert _one_rounding(_nchw(gxd, b), ref_gx) <= 0, b
        t = ref_gx / s[b].cpu().double()[None, :, None, None]
        assert _red_err(gs[b], ref_gs, (t * xb.double()).abs().sum((0, 2, 3))) < 5e-5, b
    if prev:
        # adjoint identity of the skip upsample: <up(p), g> = <p, up2_bwd(g)>   (UpsamplingLayer :603-615)
        from oracle import ref_torch as O
        gp = ops.up2_bwd(gimg)
        for b in SAMPLES(B):
            lhs = (O.s2_upsample_skip(pv[b:b + 1].cpu()).double() * gimg[b:b + 1].cpu().double()).sum()
            rhs = (pv[b:b + 1].cpu().double() * gp[b:b + 1].cpu().double()).sum()
            assert abs(float(lhs - rhs)) < 1e-5 * (pv[b].numel() ** 0.5) * 4, b


@pytest.mark.parametrize("C,R,B", [(32, 1024, 8), (64, 512, 8), (512, 32, 8)])
def test_act_bwd_pool_fullsize(C, R, B):
    """adjoint of lrelu -> avg_pool2d -> 0.111 blend of the encoder main branch with bias_2 / noise_weight_2 / conv_3.bias
    reductions (model/E/E.py:73-78,84) - dge_act_bwd<3>"""
    from dge_amd import ops
    g = _gen(300 + C)
    a = _act(B, R, R, C, g)
    gup = _act(B, R // 2, R // 2, C, g)
    noise = torch.randn(B, R, R, device=DEV, generator=g)
    red = torch.zeros(3, C, device=DEV)
    gpre = ops.act_bwd(gup, a, noise, pool=True, scale=0.111 * 0.25, red=red, planar=True)
    tot, absum = torch.zeros(3, C, dtype=torch.float64), torch.zeros(3, C, dtype=torch.float64)
    for b in range(B):
        gp, gb, gn = ER.enc_act_pool_bwd(_nchw(a, b), _nchw(gup, b), noise[b].cpu(), 0.111 * 0.25)
        if b in SAMPLES(B):
            assert _one_rounding(_nchw(gpre, b), gp) <= 0, b
        gu = _nchw(gup, b).double()
        tot += torch.stack([gb, gn, gu.sum((0, 2, 3))])
        absum += torch.stack([gp.abs().sum((0, 2, 3)), (gp * noise[b].cpu().double()).abs().sum((0, 2, 3)), gu.abs().sum((0, 2, 3))])
    assert _red_err(red, tot, absum) < 5e-5


@pytest.mark.parametrize("C,R,B,mode", [(16, 1024, 8, "act"), (16, 1024, 8, "extra"), (32, 512, 8, "act"), (64, 256, 8, "extra"),
                                       (512, 16, 8, "act"), (512, 4, 8, "nogy")])
def test_instance_norm_backward_fullsize(C, R, B, mode):
    """instance norm + (mean, std) heads backward (model/E/E.py:51-57,64-68) as the per-channel affine g_X = A g_y + B X + C:
    dge_stats_finalize_slots -> dge_in_bwd_coef_slots -> dge_in_bwd, in the three forms the encoder backward uses: with the lrelu /
    bias_1 / noise_weight_1 tail ("act"), with the pooled residual-branch gradient as a second addend ("extra"), and the last
    block's conv-less form where only the heads carry gradient ("nogy")."""
    from dge_amd import ops
    g = _gen(400 + C + R)
    X = _act(B, R, R, C, g, 1.3)
    gy = None if mode == "nogy" else _act(B, R, R, C, g)
    gms = torch.randn(B, 2 * C, device=DEV, generator=g)
    noise = torch.randn(B, R, R, device=DEV, generator=g) if mode == "act" else None
    extra = _act(B, R // 2, R // 2, C, g) if mode == "extra" else None
    Xd = X.double()
    stats = torch.stack([Xd.sum((1, 2)), (Xd * Xd).sum((1, 2))], dim=-1).float().contiguous()          # [B,C,2]
    musig, sc, sh = ops.stats_finalize(stats, R * R)
    dots = None
    if gy is not None:
        gd = gy.double()
        dots = torch.stack([(gd * Xd).sum((1, 2)), gd.sum((1, 2))], dim=-1).float().contiguous()
    del Xd
    coef = ops.in_bwd_coef(dots, gms, musig, sc, sh, R * R)
    red = torch.zeros(2, C, device=DEV) if mode == "act" else None
    gout = ops.in_bwd(gy, X, coef, extra=extra, extra_pool=extra is not None, extra_scale=0.25, noise=noise, act=(mode == "act"),
                      red=red, planar=True)
    tot, absum = torch.zeros(2, C, dtype=torch.float64), torch.zeros(2, C, dtype=torch.float64)
    for b in (range(B) if mode == "act" else SAMPLES(B)):
        ref, gb, gn = ER.enc_in_bwd(_nchw(X, b), None if gy is None else _nchw(gy, b), gms[b, :C].cpu(), gms[b, C:].cpu(),
                                    extra=None if extra is None else _nchw(extra, b), extra_scale=0.25,
                                    noise=None if noise is None else noise[b].cpu(), act=(mode == "act"))
        if b in SAMPLES(B):
            # the kernel evaluates A*g + B*X + C in f32 from f32 coefficients built out of f32 sums over 10^6 pixels: the
            # coefficient error (1e-6 relative) times |X| adds to the one-rounding bound
            viol = _one_rounding(_nchw(gout, b), ref, slack=2e-4)
            assert viol <= 0, (b, viol)
        if mode == "act":
            tot += torch.stack([gb, gn])
            absum += torch.stack([ref.abs().sum((0, 2, 3)), (ref * noise[b].cpu().double()).abs().sum((0, 2, 3))])
    if mode == "act":
        assert _red_err(red, tot, absum) < 1e-4


@pytest.mark.parametrize("C,R,B", [(16, 1024, 8), (16, 70, 3)])
def test_last_instance_norm_backward_with_fromrgb_gradients_fullsize(C, R, B):
    """The last launch of the encoder backward (dge_in_bwd_fromrgb): block 0's instance-norm backward with the pooled
    residual-branch gradient, its result g_x0 reduced in registers to the FromRGB parameter gradients (net.py:231-240) instead of
    being stored.  Oracle: enc_in_bwd (f64) -> fromrgb_bwd on the UNROUNDED g_x0 (the two-launch path rounded it to bf16)."""
    from dge_amd import ops
    import torch.nn.functional as F
    g = _gen(700 + R)
    img = torch.randn(B, 3, R, R, device=DEV, generator=g)
    w = torch.randn(C, 3, 1, 1, device=DEV, generator=g)
    bias = 0.2 * torch.randn(C, device=DEV, generator=g)
    st = torch.zeros(B, C, 2, device=DEV)
    x0 = ops.fromrgb(img, w, bias, ops.BF16, st)
    gy = _act(B, R, R, C, g)
    gms = torch.randn(B, 2 * C, device=DEV, generator=g)
    extra = _act(B, R // 2, R // 2, C, g)
    musig, sc, sh = ops.stats_finalize(st, R * R)
    gd, Xd = gy.double(), x0.double()
    dots = torch.stack([(gd * Xd).sum((1, 2)), gd.sum((1, 2))], dim=-1).float().contiguous()
    del gd, Xd
    out = ops.in_bwd_fromrgb(gy, x0, (dots, gms, musig, sc, sh, R * R), img, extra=extra, extra_pool=True, extra_scale=0.25)   # [4, C]
    # the two-launch path it replaces, for reference: same sums up to the bf16 rounding of g_x0
    two = ops.fromrgb_bwd(ops.in_bwd(gy, x0, (dots, gms, musig, sc, sh, R * R), extra=extra, extra_pool=True, extra_scale=0.25), x0, img,
                          planar=True)
    tot, absum = torch.zeros(4, C, dtype=torch.float64), torch.zeros(4, C, dtype=torch.float64)
    for b in range(B):
        ib = img[b:b + 1].cpu()
        gx0, _, _ = ER.enc_in_bwd(_nchw(x0, b), _nchw(gy, b), gms[b, :C].cpu(), gms[b, C:].cpu(), extra=_nchw(extra, b), extra_scale=0.25,
                                  noise=None, act=False)
        gW, gb = ER.fromrgb_bwd(_nchw(x0, b), gx0, ib)
        gp = gx0.double() * torch.where(_nchw(x0, b) > 0, 1.0, 0.2).double()
        tot += torch.cat([gW.t(), gb[None]])
        absum += torch.cat([torch.stack([(gp * ib.double()[:, k:k + 1]).abs().sum((0, 2, 3)) for k in range(3)]), gp.abs().sum((0, 2, 3))[None]])
    # f32 coefficients out of f32 sums over up to 10^6 pixels (1e-6 relative) times |X|, then f32 accumulation: 1e-4 of sum |terms|
    assert _red_err(out, tot, absum) < 1e-4
    assert _red_err(two, tot, absum) < 2e-3           # (what the bf16 rounding of the stored gradient cost the old path)


@pytest.mark.parametrize("C,R,B,form", [(16, 1024, 8, "pool"), (32, 512, 8, "pool"), (512, 32, 8, "res_stats"), (512, 4, 8, "in_blend")])
def test_blend_fullsize(C, R, B, form):
    """avg_pool2d / residual blend / instance-norm apply (model/E/E.py:75-78,84) - dge_blend, with the fused (sum, sumsq)"""
    from dge_amd import ops
    g = _gen(500 + C)
    x = _act(B, R, R, C, g)
    if form == "pool":
        y = ops.blend(x, pool=True)
        ref = lambda b: ER.blend(_nchw(x, b), pool=True)
    elif form == "res_stats":
        z = _act(B, R // 2, R // 2, C, g)
        st = torch.zeros(B, C, 2, device=DEV)
        y = ops.blend(x, z=z, pool=True, alpha=0.111, beta=1.0, stats=st)
        ref = lambda b: ER.blend(_nchw(x, b), z=_nchw(z, b), pool=True, alpha=0.111, beta=1.0)
    else:
        z = _act(B, R, R, C, g)
        sc = 0.5 + torch.rand(B, C, device=DEV, generator=g)
        sh = torch.randn(B, C, device=DEV, generator=g)
        y = ops.blend(x, z=z, sc=sc, sh=sh, alpha=0.111, beta=0.889)
        ref = lambda b: ER.blend(_nchw(x, b), z=_nchw(z, b), sc=sc[b].cpu(), sh=sh[b].cpu(), alpha=0.111, beta=0.889)
    for b in (range(B) if form == "res_stats" else SAMPLES(B)):
        r = ref(b)
        if b in SAMPLES(B):
            assert _one_rounding(_nchw(y, b), r) <= 0, b
        if form == "res_stats":
            # statistics are taken from the f32 values before the output rounding
            want = torch.stack([r.sum((0, 2, 3)), (r * r).sum((0, 2, 3))], 1)
            absum = torch.stack([r.abs().sum((0, 2, 3)), (r * r).sum((0, 2, 3))], 1)
            assert _red_err(st[b], want, absum) < 5e-5, b


@pytest.mark.parametrize("C,R,B", [(16, 1024, 8)])
def test_fromrgb_and_its_weight_gradient_fullsize(C, R, B):
    """FromRGB (model/utils/net.py:231-240): forward with the fused statistics, weight / bias gradient - dge_fromrgb, dge_fromrgb_bwd"""
    from dge_amd import ops
    import torch.nn.functional as F
    g = _gen(600)
    img = torch.randn(B, 3, R, R, device=DEV, generator=g)
    w = torch.randn(C, 3, 1, 1, device=DEV, generator=g)
    bias = 0.2 * torch.randn(C, device=DEV, generator=g)
    st = torch.zeros(B, C, 2, device=DEV)
    x0 = ops.fromrgb(img, w, bias, ops.BF16, st)
    gx = _act(B, R, R, C, g)
    out = ops.fromrgb_bwd(gx, x0, img, planar=True)           # [4, C]
    tot, absum = torch.zeros(4, C, dtype=torch.float64), torch.zeros(4, C, dtype=torch.float64)
    for b in range(B):
        ib = img[b:b + 1].cpu()
        ref = F.leaky_relu(F.conv2d(ib.double(), w.cpu().double(), bias.cpu().double()), 0.2)
        if b in SAMPLES(B):
            assert _one_rounding(_nchw(x0, b), ref) <= 0, b
        want = torch.stack([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))], 1)
        assert _red_err(st[b], want, torch.stack([ref.abs().sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))], 1)) < 5e-5, b
        gW, gb = ER.fromrgb_bwd(_nchw(x0, b), _nchw(gx, b), ib)
        gp = _nchw(gx, b).double() * torch.where(_nchw(x0, b) > 0, 1.0, 0.2).double()
        tot += torch.cat([gW.t(), gb[None]])
        absum += torch.cat([torch.stack([(gp * ib.double()[:, k:k + 1]).abs().sum((0, 2, 3)) for k in range(3)]), gp.abs().sum((0, 2, 3))[None]])
    assert _red_err(out, tot, absum) < 5e-5
