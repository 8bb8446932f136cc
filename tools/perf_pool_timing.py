"""Phase timestamps of one wave of the pooled encoder flavour of conv_stream (tuning build with -DDGE_SC_TIMING -DDGE_SC_ONLY=cin,cout):
   python tools/perf_pool_timing.py cin cout R"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
cin, cout, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = 8
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, R, R, cin, device="cuda", generator=g).bfloat16()
w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / math.sqrt(9 * cin)
wp = ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16)
sc = 0.5 + torch.rand(B, cin, device="cuda"); sh = torch.randn(B, cin, device="cuda")
nz = torch.randn(B, R, R, device="cuda"); nw = torch.randn(cout, device="cuda"); bias = torch.randn(cout, device="cuda")
mode = sys.argv[4] if len(sys.argv) > 4 else "pool"
kw = dict(pool_out=True, pool_mask=True) if mode == "pool" else {}
for _ in range(3):
    y = ops.conv2d(x, wp, cout, 3, in_scale=sc, in_shift=sh, noise=nz, noise_w=nw, bias=bias, act=1, **kw)
torch.cuda.synchronize()
if isinstance(y, tuple): y = y[0]
t = y.view(-1)[:80 * 4].view(torch.int64).cpu().view(16, 5)
prev = None
tot = 0
for i in range(16):
    row = t[i].tolist()
    d = [row[k + 1] - row[k] for k in range(4)]
    gap = (row[0] - prev) if prev is not None else 0
    prev = row[4]
    tot += row[4] - row[0]
    print(f"step {40+i}: wait {d[0]:6d}  issue {d[1]:6d}  kloop {d[2]:6d}  epi+store {d[3]:6d}   total {row[4]-row[0]:6d}  gap {gap}")
print(f"{mode} {cin}->{cout} @{R}: mean step {tot/16:.0f} cycles")
