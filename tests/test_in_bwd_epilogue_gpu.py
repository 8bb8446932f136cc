"""The instance-norm + activation backward of a layer's input applied in the data-gradient epilogue (dge_conv_desc.in_bwd_coef; model/E/E.py
:50-62 differentiated), with its coefficients from the weight-gradient launch (dge_conv_wgrad_dots), against the chain it replaces -
data gradient with dot statistics, then dge_in_bwd_fused in a pass of its own - and against the exact-arithmetic oracle."""
import math

import pytest
import torch

from oracle import conv_ref as CR

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _not_in_deterministic_mode():
    """the epilogue forms reduce with f32 atomics: refused in deterministic mode (the separate dge_in_bwd passes run there)"""
    from dge_amd import ops
    if ops.is_deterministic():
        pytest.skip("instance-norm backward epilogues are not offered in deterministic mode")


@pytest.mark.parametrize("B,H,W,c2,cc", [(8, 1024, 1024, 32, 16), (8, 512, 512, 64, 32), (3, 136, 200, 32, 16), (5, 128, 160, 64, 32)])
def test_fused_against_separate_passes_and_oracle(B, H, W, c2, cc):
    """conv_2 of an encoder block: cc -> c2 channels; its data gradient c2 -> cc followed by the backward of instance norm 2 and of
    conv_1's noise / bias / lrelu tail"""
    from dge_amd import ops
    from dge_amd._lib import last_kernel
    gen = torch.Generator(device=DEV).manual_seed(6000 + H + cc)
    g = torch.randn(B, H, W, c2, device=DEV, generator=gen).to(torch.bfloat16)                      # g_pre2
    x1 = (1.5 * torch.randn(B, H, W, cc, device=DEV, generator=gen) + 0.3).to(torch.bfloat16)       # lrelu output of conv_1's tail
    w = torch.randn(c2, cc, 3, 3, device=DEV, generator=gen) / math.sqrt(9 * cc)
    sc = 0.5 + torch.rand(B, cc, device=DEV, generator=gen)
    sh = 0.3 * torch.randn(B, cc, device=DEV, generator=gen)
    musig = torch.cat([0.3 * torch.randn(B, cc, device=DEV, generator=gen), 0.5 + torch.rand(B, cc, device=DEV, generator=gen)], 1)
    gms = torch.randn(B, 2 * cc, device=DEV, generator=gen)
    noise = torch.randn(B, H, W, device=DEV, generator=gen)
    wp = ops.pack_conv_weight(w, ops.PACK_DGRAD, ops.BF16, 1.0)
    N = H * W
    # ---- the separate passes
    dots0 = ops.SlotStats(B, cc, DEV)
    gy = ops.conv2d(g, wp, cc, 3, stats=dots0, dot_src=x1)
    red0 = ops.zeros((2, cc), DEV)
    ref_pre = ops.in_bwd(gy, x1, (dots0, gms, musig, sc, sh, N), noise=noise, act=True, red=red0, planar=True)
    # ---- fused
    assert ops.conv_in_bwd_supported(B, H, W, c2, cc, ops.BF16)
    dw = ops.zeros((c2, cc, 3, 3), DEV)
    dots = ops.SlotStats(B, cc, DEV)
    assert ops.conv_wgrad_dots(g, x1, dw, sc, sh, w, dots)
    coef = ops.in_bwd_coef(dots, gms, musig, sc, sh, N)
    red = ops.SlotStats(B, cc, DEV)
    got = ops.conv2d(g, wp, cc, 3, dot_src=x1, in_bwd=dict(coef=coef, noise=noise, red=red))
    assert last_kernel() == f"conv_stream<bf16,{c2},{cc},dot_in>"
    redt = red.buf.sum((0, 1)).t().cpu()                      # [2, cc]
    # (i) against the separate passes: they round g_y to bf16 before the affine, the epilogue does not - one bf16 rounding of g_y
    #     times A, relative to the result's scale
    a, b_ = got.float(), ref_pre.float()
    scale = (sc.abs().amax() * gy.float().abs().amax()).item()
    assert ((a - b_).abs().max().item()) < 2.0 ** -7 * scale
    # the reductions (bias / noise-weight gradients of conv_1's tail) cancel heavily: judged against the sums of absolute terms
    absum = torch.stack([b_.abs().sum((0, 1, 2)), (b_.abs() * noise.abs()[..., None]).sum((0, 1, 2))]).cpu()
    e = ((redt - red0.cpu()).abs() / absum).max().item()
    assert e < 1e-4, e
    # (ii) against the oracle in exact arithmetic on the same operands (bf16 weights, f64 accumulation), two samples
    wq = CR.bf16_round(w.cpu())
    coef_c = coef.cpu().double()
    for b in sorted({0, B - 1}):
        gb = g[b:b + 1].float().permute(0, 3, 1, 2).cpu().double()
        xb = x1[b:b + 1].float().permute(0, 3, 1, 2).cpu().double()
        gx = CR.conv_dgrad(gb, wq.double())
        A, Bc, Cc = (coef_c[b, :, k][None, :, None, None] for k in range(3))
        want = (A * gx + Bc * xb + Cc) * torch.where(xb > 0, 1.0, 0.2)
        gotb = got[b:b + 1].float().permute(0, 3, 1, 2).cpu().double()
        # A is folded into the bf16 weights (one operand rounding: 2^-9 relative per term, averaged over 9*c2 terms) + the output rounding
        viol = ((gotb - want).abs() - 2.0 ** -8 * want.abs() - 2.0 ** -9 * (A.abs() * gx.abs()).amax()).max().item()
        assert viol <= 0, (b, viol)


@pytest.mark.parametrize("B,H,W,with_extra", [(8, 1024, 1024, True), (3, 136, 200, True), (4, 128, 160, False)])
def test_fromrgb_reduction_in_the_last_data_gradient(B, H, W, with_extra):
    """conv_1 of block 0 (16 -> 16): its data gradient, the backward of instance norm 1, the pooled skip gradient and the FromRGB
    parameter gradients (model/utils/net.py:231-240 differentiated) in one launch, against the chain it replaces (data gradient with
    dot statistics, then dge_in_bwd_fromrgb)"""
    from dge_amd import ops
    from dge_amd._lib import last_kernel
    cc = 16
    gen = torch.Generator(device=DEV).manual_seed(6100 + H)
    g = torch.randn(B, H, W, cc, device=DEV, generator=gen).to(torch.bfloat16)                      # g_pre1
    img = torch.rand(B, 3, H, W, device=DEV, generator=gen) * 2 - 1
    wfr = torch.randn(cc, 3, device=DEV, generator=gen)
    bfr = 0.1 * torch.randn(cc, device=DEV, generator=gen)
    x0, img4 = ops.fromrgb(img, wfr, bfr, ops.BF16, img4=True)
    assert torch.equal(img4[..., :3], img.permute(0, 2, 3, 1)) and bool((img4[..., 3] == 1).all())
    w = torch.randn(cc, cc, 3, 3, device=DEV, generator=gen) / math.sqrt(9 * cc)
    sc = 0.5 + torch.rand(B, cc, device=DEV, generator=gen)
    sh = 0.3 * torch.randn(B, cc, device=DEV, generator=gen)
    musig = torch.cat([0.3 * torch.randn(B, cc, device=DEV, generator=gen), 0.5 + torch.rand(B, cc, device=DEV, generator=gen)], 1)
    gms = torch.randn(B, 2 * cc, device=DEV, generator=gen)
    extra = torch.randn(B, H // 2, W // 2, cc, device=DEV, generator=gen).to(torch.bfloat16) if with_extra else None
    wp = ops.pack_conv_weight(w, ops.PACK_DGRAD, ops.BF16, 1.0)
    N = H * W
    dots0 = ops.SlotStats(B, cc, DEV)
    gy = ops.conv2d(g, wp, cc, 3, stats=dots0, dot_src=x0)
    ref = ops.in_bwd_fromrgb(gy, x0, (dots0, gms, musig, sc, sh, N), img, extra=extra, extra_pool=True, extra_scale=0.25)     # [4, cc]
    assert ops.conv_in_bwd_fromrgb_supported(B, H, W, cc, cc, ops.BF16)
    dw = ops.zeros((cc, cc, 3, 3), DEV)
    dots = ops.SlotStats(B, cc, DEV)
    assert ops.conv_wgrad_dots(g, x0, dw, sc, sh, w, dots)
    coef = ops.in_bwd_coef(dots, gms, musig, sc, sh, N)
    frh = ops.SlotStats(B, cc, DEV)
    ops.conv2d(g, wp, cc, 3, dot_src=x0, out=x0.new_empty((1, 1, 1, 1)), in_bwd=dict(coef=coef, fr=frh, img4=img4, extra=extra, extra_scale=0.25))
    assert last_kernel() == "conv_stream<bf16,16,16,dot_fromrgb>"
    got = frh.buf.sum((0, 1)).t()                      # [4, cc]
    # sums with heavy cancellation: judged against the sums of absolute terms, taken from the separate passes' operands
    coef_ = coef[:, None, None, :, :]
    gx = coef_[..., 0] * gy.float() + coef_[..., 1] * x0.float() + coef_[..., 2]
    if with_extra:
        gx = gx + 0.25 * extra.float().repeat_interleave(2, 1).repeat_interleave(2, 2)
    gp = gx * torch.where(x0.float() > 0, 1.0, 0.2)
    absum = torch.stack([(gp.abs() * img4[..., k:k + 1].abs()).sum((0, 1, 2)) for k in range(4)])
    want = torch.stack([(gp.double() * img4[..., k:k + 1].double()).sum((0, 1, 2)) for k in range(4)])
    e_ref = ((ref.double() - want).abs() / absum).max().item()
    e_got = ((got.double() - want).abs() / absum).max().item()
    assert e_ref < 1e-4 and e_got < 1e-4, (e_ref, e_got)


@pytest.mark.parametrize("B,H,W,cc,with_extra", [(8, 512, 512, 32, True), (8, 256, 256, 64, True), (3, 136, 200, 32, True), (5, 128, 160, 64, False)])
def test_block_input_form(B, H, W, cc, with_extra):
    """conv_1 of an encoder block (cc -> cc): data gradient + instance-norm backward of the block input + pooled skip gradient in one
    launch, against data gradient with dot statistics followed by dge_in_bwd_fused(act = 0)"""
    from dge_amd import ops
    from dge_amd._lib import last_kernel
    gen = torch.Generator(device=DEV).manual_seed(6200 + H + cc)
    g = torch.randn(B, H, W, cc, device=DEV, generator=gen).to(torch.bfloat16)
    x = (1.5 * torch.randn(B, H, W, cc, device=DEV, generator=gen) + 0.3).to(torch.bfloat16)
    w = torch.randn(cc, cc, 3, 3, device=DEV, generator=gen) / math.sqrt(9 * cc)
    sc = 0.5 + torch.rand(B, cc, device=DEV, generator=gen)
    sh = 0.3 * torch.randn(B, cc, device=DEV, generator=gen)
    musig = torch.cat([0.3 * torch.randn(B, cc, device=DEV, generator=gen), 0.5 + torch.rand(B, cc, device=DEV, generator=gen)], 1)
    gms = torch.randn(B, 2 * cc, device=DEV, generator=gen)
    extra = torch.randn(B, H // 2, W // 2, cc, device=DEV, generator=gen).to(torch.bfloat16) if with_extra else None
    wp = ops.pack_conv_weight(w, ops.PACK_DGRAD, ops.BF16, 1.0)
    N = H * W
    dots0 = ops.SlotStats(B, cc, DEV)
    gy = ops.conv2d(g, wp, cc, 3, stats=dots0, dot_src=x)
    ref = ops.in_bwd(gy, x, (dots0, gms, musig, sc, sh, N), extra=extra, extra_pool=True, extra_scale=0.25)
    assert ops.conv_in_bwd_x_supported(B, H, W, cc, cc, ops.BF16)
    dw = ops.zeros((cc, cc, 3, 3), DEV)
    dots = ops.SlotStats(B, cc, DEV)
    assert ops.conv_wgrad_dots(g, x, dw, sc, sh, w, dots)
    coef = ops.in_bwd_coef(dots, gms, musig, sc, sh, N)
    got = ops.conv2d(g, wp, cc, 3, dot_src=x, in_bwd=dict(coef=coef, extra=extra, extra_scale=0.25))
    assert last_kernel() == f"conv_stream<bf16,{cc},{cc},dot_inx>"
    scale = (sc.abs().amax() * gy.float().abs().amax()).item()
    assert ((got.float() - ref.float()).abs().max().item()) < 2.0 ** -7 * scale
    # exact arithmetic on the same operands, two samples
    wq = CR.bf16_round(w.cpu())
    coef_c = coef.cpu().double()
    for b in sorted({0, B - 1}):
        gb = g[b:b + 1].float().permute(0, 3, 1, 2).cpu().double()
        xb = x[b:b + 1].float().permute(0, 3, 1, 2).cpu().double()
        gx = CR.conv_dgrad(gb, wq.double())
        A, Bc, Cc = (coef_c[b, :, k][None, :, None, None] for k in range(3))
        want = A * gx + Bc * xb + Cc
        if with_extra:
            want = want + 0.25 * extra[b:b + 1].float().permute(0, 3, 1, 2).cpu().double().repeat_interleave(2, 2).repeat_interleave(2, 3)
        gotb = got[b:b + 1].float().permute(0, 3, 1, 2).cpu().double()
        viol = ((gotb - want).abs() - 2.0 ** -8 * want.abs() - 2.0 ** -9 * (A.abs() * gx.abs()).amax()).max().item()
        assert viol <= 0, (b, viol)
