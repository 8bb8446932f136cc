import sys, os
sys.path.insert(0, "/root/repo")
import torch
from dge_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, R, C in [(8, 1024, 32), (8, 512, 64), (8, 256, 128)]:
    gup = torch.randn(B, R // 2, R // 2, C, device="cuda").bfloat16()
    mask = torch.randint(0, 2 ** 31 - 1, (B, (R // 2) ** 2, C // 8), device="cuda", dtype=torch.int64).to(torch.int32)
    noise = torch.randn(B, R, R, device="cuda")
    red = ops.zeros((3, C), "cuda")
    print(B, R, C, round(timeit(lambda: ops.act_bwd_mask(gup, mask, noise, scale=0.03, red=red, planar=True)), 1), "us")
