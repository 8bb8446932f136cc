"""Ablation of the space-to-depth data gradient of the narrow up layers (layer 15: 64 -> 32 @ 512 -> 1024) by the DGE_CONV_DBG
bits of conv_igemm (1: no weight DMA, 2: no MFMA, 4: no epilogue, 8: no main loop) - dev tool."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dge_amd
from dge_amd import ops
from dge_amd._lib import last_kernel
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 8
g = torch.Generator(device="cuda").manual_seed(0)
for cof, cif, R in [(32, 64, 512), (64, 128, 256)]:
    gz = torch.randn(B, 2 * R, 2 * R, cof, device="cuda", generator=g).to(torch.bfloat16)
    x = torch.randn(B, R, R, cif, device="cuda", generator=g).to(torch.bfloat16)
    add = torch.randn(B, R, R, cif, device="cuda", generator=g).to(torch.bfloat16)
    s = 1 + 0.3 * torch.randn(B, cif, device="cuda", generator=g)
    w = torch.randn(cof, cif, 3, 3, device="cuda", generator=g)
    pk = ops.pack_conv_weight(w, ops.PACK_UPFOLD_DGRAD, ops.BF16, 1 / math.sqrt(9 * cif))
    st = ops.SlotStats(B, cif, "cuda")
    variants = {"full (dot+add+stats)": dict(out_scale=s, addend=add, stats=st, dot_src=x),
                "no addend": dict(out_scale=s, stats=st, dot_src=x),
                "plain (no dot/add/stats)": dict()}
    for name, kw in variants.items():
        t = timeit(lambda: ops.conv2d(gz, pk, cif, 3, in_s2d=True, **kw))
        print(f"{cof}->{cif}@{R} DBG={os.environ.get('DGE_CONV_DBG', '0'):>2} {name:28s} {t:8.1f} us  {last_kernel()}")
