"""conv_2 weight gradient of the first encoder blocks: materialised gradient (act_bwd_mask + wgrad_dma) against the pooled form (dev tool)"""
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
    gup = torch.randn(B, R // 2, R // 2, cout, device=DEV, generator=g).bfloat16()
    mask = torch.randint(0, 2 ** 31 - 1, (B, (R // 2) ** 2, cout // 8), device=DEV, generator=g, dtype=torch.int64).to(torch.int32)
    noise = torch.randn(B, R, R, device=DEV, generator=g)
    x = torch.randn(B, R, R, cin, device=DEV, generator=g).bfloat16()
    w = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / math.sqrt(9 * cin)
    sc = 0.5 + torch.rand(B, cin, device=DEV); sh = torch.randn(B, cin, device=DEV)
    dw = ops.zeros((cout, cin, 3, 3), DEV)
    red = ops.zeros((2, cout), DEV)
    gfull = ops.act_bwd_mask(gup, mask, noise, scale=0.03, red=red, planar=True)
    t_ab = timeit(lambda: ops.act_bwd_mask(gup, mask, noise, scale=0.03, red=red, planar=True))
    t_ab0 = timeit(lambda: ops.act_bwd_mask(gup, mask, noise, scale=0.03, red=red, planar=True, store=False))
    t_w = timeit(lambda: ops.conv_wgrad_dots(gfull, x, dw, sc, sh, w, ops.SlotStats(B, cin, DEV)))
    k0 = last_kernel()
    t_wp = timeit(lambda: ops.conv_wgrad_dots_pool(ops.PooledGrad(gup, mask, 0.03), x, dw, sc, sh, w, ops.SlotStats(B, cin, DEV)))
    print(f"B{B} {cin}->{cout} @{R}: act_bwd_mask {t_ab:6.1f} us (no store {t_ab0:6.1f})   wgrad {t_w:6.1f} us [{k0}]   pooled wgrad {t_wp:6.1f} us [{last_kernel()}]", flush=True)
