"""conv_small on hot / memory-side-cache-resident / HBM-cold weights (dev probe): is a weight prefetch worth building?"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dge_amd import ops
from dge_amd._lib import last_kernel
C = 512
g = torch.Generator(device="cuda").manual_seed(0)
for B, H in ((8, 16), (8, 8), (8, 4), (16, 16)):
    x = torch.randn(B, H, H, C, device="cuda", generator=g).to(torch.bfloat16)
    s = 1 + 0.3 * torch.randn(B, C, device="cuda", generator=g); d = 0.5 + torch.rand(B, C, device="cuda", generator=g)
    noise = torch.randn(1, H, H, device="cuda", generator=g); ns = torch.tensor([0.3], device="cuda"); bias = torch.randn(C, device="cuda", generator=g)
    for nw in (1, 12, 80):
        ws = []
        for i in range(nw):
            w = torch.randn(C, C, 3, 3, device="cuda", generator=g)
            ws.append(ops.pack_conv_weight(w, ops.pack_mode_for(w, ops.PACK_FWD, H, H, ops.BF16), ops.BF16, 1 / math.sqrt(9 * C)))
        def run(i):
            return ops.conv2d(x, ws[i % nw], C, 3, in_scale=s, out_scale=d, bias=bias, noise=noise, noise_w=ns, act=1, gain=1.414)
        for i in range(nw + 3): run(i)
        torch.cuda.synchronize()
        n = max(40, nw * 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): run(i)
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} {H}^2 weights x{nw} ({nw * 4.7:.0f} MB): {e0.elapsed_time(e1) / n * 1e3:.1f} us  {last_kernel()}", flush=True)
