"""Low-resolution 3x3 layers: the general kernel's small-tile configuration vs csrc/conv_small.hip (dev tool, GPU box).
   python tools/perf_small.py   (DGE_SMALL_MAXHW=32 to include 32^2)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dge_amd
from dge_amd import ops
from dge_amd._lib import last_kernel

def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

C = 512
g = torch.Generator(device="cuda").manual_seed(0)
w = torch.randn(C, C, 3, 3, device="cuda", generator=g)
print(f"{'shape':>16} {'mode':>6} {'igemm us':>9} {'small us':>9}  kernels")
for B, H, W in [(8, 4, 4), (8, 8, 8), (8, 16, 16), (2, 16, 16), (16, 16, 16), (16, 16, 12), (16, 11, 11), (16, 22, 22), (8, 32, 32), (16, 32, 32), (16, 32, 24)]:
    x = torch.randn(B, H, W, C, device="cuda", generator=g).to(torch.bfloat16)
    s = 1 + 0.3 * torch.randn(B, C, device="cuda", generator=g)
    d = 0.5 + torch.rand(B, C, device="cuda", generator=g)
    noise = torch.randn(1, H, W, device="cuda", generator=g)
    ns = torch.tensor([0.3], device="cuda"); bias = torch.randn(C, device="cuda", generator=g)
    for mode, name in ((ops.PACK_FWD, "fwd"), (ops.PACK_DGRAD, "dgrad"), (ops.PACK_UPFOLD, "upfold")):
        if mode == ops.PACK_UPFOLD and H > 8: continue
        res = []
        for frag in (False, True):
            m = mode
            if frag:
                m = ops.pack_mode_for(w, mode, H, W, ops.BF16)
                if not (m & ops.PACK_FRAG):
                    res.append((float("nan"), "-")); continue
            pk = ops.pack_conv_weight(w, m, ops.BF16, 1 / math.sqrt(9 * C))
            if mode == ops.PACK_DGRAD:
                st = torch.zeros(64, B, C, 2, device="cuda")
                f = lambda: ops.conv2d(x, pk, C, 3, out_scale=s, stats=st[0], dot_src=x)
            else:
                nz = torch.randn(1, 2 * H, 2 * W, device="cuda") if mode == ops.PACK_UPFOLD else noise
                f = lambda: ops.conv2d(x, pk, C, 3, up=(mode == ops.PACK_UPFOLD), in_scale=s, out_scale=d, bias=bias, noise=nz, noise_w=ns, act=1, gain=1.414)
            t = timeit(f)
            res.append((t, last_kernel()))
        print(f"{str((B, H, W)):>16} {name:>6} {res[0][0]:9.1f} {res[1][0]:9.1f}  {res[0][1]} | {res[1][1]}")
