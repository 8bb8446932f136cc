"""conv_pp data-gradient form against conv_igemm on the synthesis backward's MFMA-bound launches (dev tool): the data gradient of
layers 12 / 10 / 8 and the folded adjoint of layer 13, each with the fused tail backward of the layer below.  Interleaved rounds in
one process, median per variant.  python tools/perf_pp_dg.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
from dge_amd._lib import last_kernel

# (B, cof, cif, R, up, addend)
CASES = [(8, 128, 128, 256, False, False), (8, 64, 128, 256, True, True), (8, 256, 256, 128, False, False), (8, 512, 512, 64, False, False),
         (8, 128, 256, 128, "t2d", True), (8, 256, 512, 64, "t2d", True)]


def timed(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, cof, cif, R, up, with_add in CASES:
    t2d = up == "t2d"
    up = bool(up)
    g = torch.Generator(device="cuda").manual_seed(1)
    Rg = 2 * R if up else R
    gz = torch.randn(B, Rg, Rg, cof, device="cuda", generator=g).bfloat16()
    d = 0.5 + torch.rand(B, cof, device="cuda", generator=g)
    xin = (1.5 * torch.randn(B, R, R, cif, device="cuda", generator=g)).bfloat16()
    add = torch.randn(B, R, R, cif, device="cuda", generator=g).bfloat16() if with_add else None
    w = torch.randn(cof, cif, 3, 3, device="cuda", generator=g)
    wscale = 1.0 / (9 * cif) ** 0.5
    s = 1.0 + 0.3 * torch.randn(B, cif, device="cuda", generator=g)
    nz = torch.randn(1, R, R, device="cuda", generator=g)
    ns = torch.full((1,), 0.37, device="cuda")
    gain = 2 ** 0.5
    mode = ops.PACK_UPT2D_DGRAD if t2d else (ops.PACK_UPFOLD_DGRAD if up else ops.PACK_DGRAD)
    wp = ops.pack_conv_weight(w, mode, ops.BF16, wscale)
    rows = ops.pack_conv_weight(w, mode, ops.F32, wscale) if up else None
    K = 4 * cof if up else cof
    wpp = torch.empty((1 if t2d else B, (4 if t2d else 9) * cif * K), dtype=torch.bfloat16, device="cuda")

    def old():
        st, P = ops.SlotStats(B, cif, "cuda"), ops.SlotStats(B, cif, "cuda")
        if t2d:
            return ops.conv2d(z, wp, cif, 3, in_t2d=True, out_scale=s, addend=add, add_scale=1.0, stats=st, dot_src=xin,
                              prep=dict(gain=gain, noise=nz, ns=ns, stats=P)), st, P
        return ops.conv2d(gz, wp, cif, 3, in_s2d=up, in_scale=d, out_scale=s, addend=add, add_scale=1.0, stats=st, dot_src=xin,
                          prep=dict(gain=gain, noise=nz, ns=ns, stats=P)), st, P

    z = ops.fir_t2d(gz, d) if t2d else None

    def fold():
        if t2d:
            ops.pack_conv_pp_rows(rows, cif, t2d=True, out=wpp)
        elif up:
            ops.pack_conv_pp_rows(rows, cif, in_scale=d, in_period=cof, out=wpp)
        else:
            ops.pack_conv_pp(w, wscale, in_scale=d, dgrad=True, out=wpp)

    def new():
        st, P = ops.SlotStats(B, cif, "cuda"), ops.SlotStats(B, cif, "cuda")
        return ops.conv_pp(z if t2d else gz, wpp, cif, dgrad=True, in_s2d=up and not t2d, in_t2d=t2d, out_scale=s, addend=add, add_scale=1.0, stats=st, dot_src=xin,
                           prep=dict(gain=gain, noise=nz, ns=ns, stats=P)), st, P

    fold()
    y0, st0, P0 = old(); k0 = last_kernel()
    y1, st1, P1 = new(); k1 = last_kernel()
    err = ((y0.float() - y1.float()).abs().max() / y0.float().abs().max()).item()
    es = ((st0.buf.sum(0) - st1.buf.sum(0)).abs().max() / st0.buf.sum(0).abs().max()).item()
    ep = ((P0.buf.sum(0) - P1.buf.sum(0)).abs().max() / P0.buf.sum(0).abs().max()).item()
    for f in (old, new, fold):
        for _ in range(3):
            f()
    r = {"old": [], "new": [], "fold": []}
    for _ in range(5):
        r["old"].append(timed(old)); r["new"].append(timed(new)); r["fold"].append(timed(fold))
    fl = 2 * (4 if t2d else 9) * K * cif * R * R * B
    m = {k: statistics.median(v) for k, v in r.items()}
    print(f"B={B} dgrad of {cif}->{cof} @{R}^2 up={up}: {k0} {m['old']:.1f} us ({fl / m['old'] / 1e6:.0f} TF/s executed) | {k1} {m['new']:.1f} us "
          f"({fl / m['new'] / 1e6:.0f} TF/s) + fold {m['fold']:.1f} us | rel diff y {err:.2e} stats {es:.2e} prep {ep:.2e}  dbg={os.environ.get('DGE_CONV_DBG', '0')}", flush=True)
