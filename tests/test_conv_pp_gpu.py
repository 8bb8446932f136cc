"""Parity of the ping-pong implicit GEMM (csrc/conv_pp.hip) at the shapes the benchmark runs it on, asserted by kernel name, plus
ragged shapes (partial tiles in both directions, odd chunk counts) and the shared-weight (LPIPS) form.  Bounds as in
tests/test_fullsize_gpu.py: every element within ONE bf16 rounding of the exact-arithmetic oracle that places the storage
rounding where the kernel does (oracle/conv_ref.py:modconv_folded - the reference's fused modulation :858-875), and a storage
bound against the plain f32 oracle."""
import math

import pytest
import torch

from oracle import conv_ref as CR

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _nchw(x_nhwc, b):
    return x_nhwc[b:b + 1].float().permute(0, 3, 1, 2).contiguous().cpu()


def _one_rounding(got, ref, slack=1e-5):
    return ((got - ref).abs() - (2.0 ** -8) * ref.abs() - slack * ref.abs().max()).max().item()


def _case(B, H, W, cin, cout, seed, modulated=True, samples=None):
    from dge_amd import ops
    from dge_amd._lib import last_kernel
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(B, H, W, cin, device=DEV, generator=g).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=DEV, generator=g).to(torch.bfloat16).float()
    wscale = 1.0 / math.sqrt(9 * cin)
    noise = torch.randn(1, H, W, device=DEV, generator=g)
    ns = torch.tensor([0.37], device=DEV)
    bias = 0.2 * torch.randn(cout, device=DEV, generator=g)
    gain = math.sqrt(2.0)
    assert ops.conv_pp_supported(B, H, W, cin, cout, ops.BF16)
    if modulated:
        s = 1.0 + 0.3 * torch.randn(B, cin, device=DEV, generator=g)
        d = 0.5 + torch.rand(B, cout, device=DEV, generator=g)
        wpp = ops.pack_conv_pp(w, wscale, in_scale=s, out_scale=d, gain=gain)
        y = ops.conv_pp(x, wpp, cout, bias=bias, bias_scale=1.0, noise=noise, noise_w=ns, act=ops.ACT_LRELU, gain=gain)
    else:
        s = d = None
        wpp = ops.pack_conv_pp(w, wscale)
        y = ops.conv_pp(x, wpp, cout, bias=bias, bias_scale=1.0, act=ops.ACT_RELU, gain=1.0)
    assert last_kernel() == "conv_pp<bf16,16,32,128>"
    for b in (samples if samples is not None else sorted({0, B - 1})):
        if modulated:
            a = (_nchw(x, b), w.cpu(), s[b:b + 1].cpu(), d[b:b + 1].cpu(), noise.cpu(), 0.37, bias.cpu(), 1.0, wscale)
            ref_q, ref = CR.modconv_folded(*a, q=CR.bf16_round), CR.modconv(*a)
        else:
            a = (_nchw(x, b), w.cpu(), None, None, None, 0.0, bias.cpu(), 1.0, wscale)
            ref_q = CR.modconv_folded(*a, gain=1.0, slope=0.0, q=CR.bf16_round)
            ref = CR.modconv(*a, gain=1.0, slope=0.0)
        got = _nchw(y, b)
        assert _one_rounding(got, ref_q) <= 0, (b, _one_rounding(got, ref_q))
        e = ((got - ref).abs().max() / ref.abs().max()).item()
        assert e < 8e-3, (b, e)


@pytest.mark.parametrize("cin,cout,R,B", [(128, 128, 256, 8), (256, 256, 128, 8), (512, 512, 64, 8)])
def test_generator_layers_fullsize(cin, cout, R, B):
    """layers 12 / 10 / 8 of the StyleGAN2-1024 synthesis at batch 8 (stylegan2_generator.py:855-922, stride-1 branch)"""
    _case(B, R, R, cin, cout, 4000 + cin)


@pytest.mark.parametrize("B,H,W,cin,cout", [(40, 50, 70, 64, 128), (12, 33, 97, 96, 256), (48, 44, 44, 128, 128), (16, 129, 63, 160, 128)])
def test_ragged_shapes(B, H, W, cin, cout):
    """partial tiles in y and x, 2 / 3 / 4 / 5 K chunks, every sample checked"""
    _case(B, H, W, cin, cout, 4100 + H, samples=range(0, B, max(1, B // 6)))


@pytest.mark.parametrize("B,H,W,cin,cout", [(16, 128, 128, 64, 128), (16, 64, 48, 256, 256)])
def test_shared_weights_bias_relu(B, H, W, cin, cout):
    """one shared weight image, bias + ReLU epilogue: the LPIPS VGG16 convs (third-party lpips algorithm, call site training_utils.py:93)"""
    _case(B, H, W, cin, cout, 4200 + H, modulated=False)


# ---------------------------------------------------------------------------------------------------------------- data-gradient form
def _dg_case(B, R, cof, cif, seed, up=False, with_add=False, mode="prep", samples=None, Rw=None, t2d=False):
    """Data gradient of layer i (cif -> cof channels forward) on conv_pp: g_z of layer i in, the demodulation factor folded into the
    per-sample weight image; epilogue per `mode`: "prep" (synthesis chain: style-gradient sums, fused tail backward of layer i-1,
    stylegan2_generator.py:908-921 adjoint), "stats" (encoder: the two sums the instance-norm backward needs), "mask" (LPIPS: ReLU
    backward of the layer below).  Oracle: oracle/conv_ref.py:dgrad_folded (rounding where the kernel has it) + oracle/elem_ref.py."""
    from dge_amd import ops
    from dge_amd._lib import last_kernel
    from oracle import elem_ref as ER
    if ops.is_deterministic() and mode != "mask":
        pytest.skip("the statistics of conv_pp's data-gradient form are f32 atomics: refused in deterministic mode")
    H, W = R, (Rw or R)
    g = torch.Generator(device=DEV).manual_seed(seed)
    gain = math.sqrt(2.0)
    Hg, Wg = (2 * H, 2 * W) if up else (H, W)
    gz_in = torch.randn(B, Hg, Wg, cof, device=DEV, generator=g).to(torch.bfloat16)
    d_in = 0.5 + torch.rand(B, cof, device=DEV, generator=g)
    xin = (1.5 * torch.randn(B, H, W, cif, device=DEV, generator=g)).to(torch.bfloat16)
    add = torch.randn(B, H, W, cif, device=DEV, generator=g).to(torch.bfloat16) if with_add else None
    w = torch.randn(cof, cif, 3, 3, device=DEV, generator=g).to(torch.bfloat16).float()
    wscale = 1.0 / math.sqrt(9 * cif)
    s = 1.0 + 0.3 * torch.randn(B, cif, device=DEV, generator=g)
    noise = torch.randn(1, H, W, device=DEV, generator=g)
    ns = torch.tensor([0.37], device=DEV)
    assert ops.conv_pp_supported(B, H, W, 4 * cof if up else cof, cif, ops.BF16)
    if t2d:        # phase form: FIR^T (times the demodulation factor) to the t grid in a pass of its own, shared 4-tap weights
        rows = ops.pack_conv_weight(w, ops.PACK_UPT2D_DGRAD, ops.F32, wscale)
        wpp = ops.pack_conv_pp_rows(rows, cif, t2d=True)
    elif up:
        rows = ops.pack_conv_weight(w, ops.PACK_UPFOLD_DGRAD, ops.F32, wscale)
        wpp = ops.pack_conv_pp_rows(rows, cif, in_scale=d_in, in_period=cof)
    else:
        wpp = ops.pack_conv_pp(w, wscale, in_scale=d_in, dgrad=True)
    st, P = ops.SlotStats(B, cif, DEV), ops.SlotStats(B, cif, DEV)
    kw = dict(dgrad=True, in_s2d=up and not t2d, out_scale=s, addend=add, add_scale=1.0)
    if t2d:
        out = ops.conv_pp(ops.fir_t2d(gz_in, d_in), wpp, cif, in_t2d=True, stats=st, dot_src=xin, prep=dict(gain=gain, noise=noise, ns=ns, stats=P), **kw)
        want_kernel = "conv_pp<bf16,16,32,128>+dg+t2d+prep"
    elif mode == "prep":
        out = ops.conv_pp(gz_in, wpp, cif, stats=st, dot_src=xin, prep=dict(gain=gain, noise=noise, ns=ns, stats=P), **kw)
        want_kernel = "conv_pp<bf16,16,32,128>+dg" + ("+s2d" if up else "") + "+prep"
    elif mode == "stats":
        out = ops.conv_pp(gz_in, wpp, cif, stats=st, dot_src=xin, **kw)
        want_kernel = "conv_pp<bf16,16,32,128>+dg" + ("+s2d" if up else "")
    else:
        out = ops.conv_pp(gz_in, wpp, cif, relu_mask=xin, **kw)
        want_kernel = "conv_pp<bf16,16,32,128>+dg+mask"
    assert last_kernel() == want_kernel
    stt = st.buf.sum(0).cpu() if st.buf is not None else None
    Pt = P.buf.sum(0).cpu() if P.buf is not None else None
    for b in (samples if samples is not None else sorted({0, B - 1})):
        if t2d:    # exact arithmetic on the kernel's operands: Z = bf16(FIR^T(g_z * d)) (the pass of its own, tests/test_fullsize_gpu.py), bf16 weights
            xz = torch.zeros(1, cif, H, W, requires_grad=True)
            raw = torch.autograd.grad(CR.up_linear(xz, CR.bf16_round(w.cpu() * wscale), 1.0), xz, _nchw(gz_in, b) * d_in[b].cpu()[None, :, None, None])[0]
        else:
            raw = CR.dgrad_folded(_nchw(gz_in, b), w.cpu(), wscale, d_in[b].cpu(), up=up, q=CR.bf16_round)
        gref = raw * s[b].cpu()[None, :, None, None]
        if with_add:
            gref = gref + _nchw(add, b)
        xb = _nchw(xin, b)
        if mode == "prep":
            ref, ref_R = ER.modconv_tail_bwd(xb, gref, torch.ones(cif), noise[0].cpu(), gain)
            ref = ref.float()
        elif mode == "mask":
            ref = gref * (xb > 0)
        else:
            ref = gref
        # (phase form: Z is stored in bf16 - one more operand rounding than the oracle has; bound as in tests/test_fullsize_gpu.py)
        v = _one_rounding(_nchw(out, b), ref, slack=6e-3 if t2d else 1e-5)
        assert v <= 0, (b, v)
        stol = 7.9e-3 if t2d else 1e-5
        if mode in ("prep", "stats"):
            rd, xd = raw.double(), xb.double()
            for k, (want, absum) in enumerate((((rd * xd).sum((0, 2, 3)), (rd * xd).abs().sum((0, 2, 3))), (rd.sum((0, 2, 3)), rd.abs().sum((0, 2, 3))))):
                if mode == "prep" and k == 1:          # (the prep flavour leaves the plain sum out: the synthesis chain has no use for it)
                    continue
                e = ((stt[b, :, k].double() - want).abs() / absum).max().item()
                assert e < stol, (b, k, e)
        if mode == "prep":
            gzd = ref.double()
            zt = ER.lrelu_inverse(xb.double(), gain) - 0.37 * noise[0].cpu().double()[None, None]
            for k, (want, absum) in enumerate((((gzd * zt).sum((0, 2, 3)), (gzd * zt).abs().sum((0, 2, 3))), (gzd.sum((0, 2, 3)), gzd.abs().sum((0, 2, 3))))):
                e = ((Pt[b, :, k].double() - want).abs() / absum).max().item()
                assert e < stol, (b, k, e)


@pytest.mark.parametrize("cof,cif,R,B,up,with_add", [(128, 128, 256, 8, False, False), (256, 256, 128, 8, False, False), (512, 512, 64, 8, False, False),
                                                     (64, 128, 256, 8, True, True)])
def test_synthesis_data_gradients_fullsize(cof, cif, R, B, up, with_add):
    """layers 12 / 10 / 8 (stride 1) and layer 13 (up, space-to-depth read, toRGB addend) of the StyleGAN2-1024 synthesis backward at
    batch 8, with the fused tail backward of the layer below"""
    _dg_case(B, R, cof, cif, 4300 + cof + R, up=up, with_add=with_add)


@pytest.mark.parametrize("B,H,W,cof,cif,up,mode,with_add", [(40, 50, 70, 64, 128, False, "prep", True), (12, 33, 97, 96, 256, False, "stats", True),
                                                           (48, 44, 44, 128, 128, False, "mask", False), (40, 50, 70, 96, 128, False, "mask", True), (24, 70, 50, 32, 128, True, "prep", False),
                                                           (16, 129, 63, 160, 128, False, "stats", False)])
def test_data_gradient_ragged_and_modes(B, H, W, cof, cif, up, mode, with_add):
    """partial tiles in y and x, 2 - 5 K chunks (4 x 1 under space-to-depth), the three epilogue flavours, with and without addend"""
    _dg_case(B, H, cof, cif, 4400 + H, up=up, with_add=with_add, mode=mode, samples=range(0, B, max(1, B // 4)), Rw=W)


@pytest.mark.parametrize("cof,cif,R,B", [(128, 256, 128, 8), (256, 512, 64, 8)])
def test_phase_form_adjoints_fullsize(cof, cif, R, B):
    """layers 11 / 9 of the StyleGAN2-1024 synthesis backward in phase form (dge_fir_t2d + the 4-tap conv on the t grid) at batch 8"""
    _dg_case(B, R, cof, cif, 4500 + cof, up=True, with_add=True, t2d=True)


@pytest.mark.parametrize("B,H,W,cof,cif,with_add", [(24, 40, 70, 32, 128, True), (12, 33, 97, 64, 256, False)])
def test_phase_form_ragged(B, H, W, cof, cif, with_add):
    """partial tiles in y and x (the source has H + 1 rows / W + 1 columns), 4 and 8 K chunks"""
    _dg_case(B, H, cof, cif, 4600 + H, up=True, with_add=with_add, t2d=True, samples=range(0, B, max(1, B // 4)), Rw=W)
