"""conv_pp (ping-pong implicit GEMM) against conv_igemm on the MFMA-bound stride-1 layers (dev tool): interleaved rounds in one
process, median per variant.  python tools/perf_pp.py [quick]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
from dge_amd._lib import last_kernel

SHAPES = [(8, 128, 128, 256), (8, 256, 256, 128), (8, 512, 512, 64), (16, 128, 128, 128), (16, 256, 256, 64), (16, 64, 128, 128)]
if len(sys.argv) > 1 and sys.argv[1] == "small":
    SHAPES = [(16, 512, 512, 32), (8, 512, 512, 32), (16, 256, 256, 48), (16, 256, 256, 44), (16, 128, 128, 88), (8, 256, 128, 64), (8, 128, 128, 128), (8, 256, 256, 64)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    SHAPES = SHAPES[:3]


def timed(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, cin, cout, H in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, H, H, cin, device="cuda", generator=g).bfloat16()
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g)
    wscale = 1.0 / (9 * cin) ** 0.5
    s = 1.0 + 0.3 * torch.randn(B, cin, device="cuda", generator=g)
    d = 0.5 + torch.rand(B, cout, device="cuda", generator=g)
    bias = torch.randn(cout, device="cuda", generator=g)
    nz = torch.randn(1, H, H, device="cuda", generator=g)
    nw = torch.full((1,), 0.3, device="cuda")
    wp = ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16, wscale)
    old = lambda: ops.conv2d(x, wp, cout, 3, in_scale=s, out_scale=d, bias=bias, noise=nz, noise_w=nw, act=1, gain=2 ** 0.5)
    wpp = torch.empty((B, 9 * cin * cout), dtype=torch.bfloat16, device="cuda")
    fold = lambda: ops.pack_conv_pp(w, wscale, in_scale=s, out_scale=d, gain=2 ** 0.5, out=wpp)
    new = lambda: ops.conv_pp(x, wpp, cout, bias=bias, noise=nz, noise_w=nw, act=1, gain=2 ** 0.5)
    fold()
    y0 = old(); k0 = last_kernel()
    y1 = new(); k1 = last_kernel()
    err = ((y0.float() - y1.float()).abs().max() / y0.float().abs().max()).item()
    for f in (old, new, fold):
        for _ in range(3):
            f()
    r = {"old": [], "new": [], "fold": []}
    for _ in range(5):
        r["old"].append(timed(old)); r["new"].append(timed(new)); r["fold"].append(timed(fold))
    fl = 2 * 9 * cin * cout * H * H * B
    m = {k: statistics.median(v) for k, v in r.items()}
    print(f"B={B} {cin}->{cout} @{H}^2: {k0} {m['old']:.1f} us ({fl / m['old'] / 1e6:.0f} TF/s) | {k1} {m['new']:.1f} us "
          f"({fl / m['new'] / 1e6:.0f} TF/s) + fold {m['fold']:.1f} us | rel diff {err:.2e}  dbg={os.environ.get('DGE_CONV_DBG', '0')}", flush=True)
