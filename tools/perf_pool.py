"""conv_2 of the first encoder blocks: encoder flavour against its pooled epilogue (dev tool)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
from dge_amd._lib import last_kernel
DEV = "cuda"


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, R, cin, cout in [(8, 1024, 16, 32), (8, 512, 32, 64)]:
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, R, R, cin, device=DEV, generator=g).bfloat16()
    w = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / math.sqrt(9 * cin)
    wp = ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16)
    sc = 0.5 + torch.rand(B, cin, device=DEV); sh = torch.randn(B, cin, device=DEV)
    nz = torch.randn(B, R, R, device=DEV); nw = torch.randn(cout, device=DEV); bias = torch.randn(cout, device=DEV)
    for name, kw in (("enc", {}), ("pool", dict(pool_out=True)), ("pool+mask", dict(pool_out=True, pool_mask=True))):
        fn = lambda: ops.conv2d(x, wp, cout, 3, in_scale=sc, in_shift=sh, noise=nz, noise_w=nw, bias=bias, act=1, **kw)
        t = timeit(fn)
        print(f"B{B} {cin}->{cout} @{R} {name:10s}: {t:7.1f} us  [{last_kernel()}]  dbg={os.environ.get('DGE_CONV_DBG','0')}", flush=True)
