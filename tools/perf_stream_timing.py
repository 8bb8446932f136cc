"""Phase timestamps of one wave of conv_stream (tuning build with -DDGE_SC_TIMING): cycles per phase over 16 steps."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
B, R, cin, cout = 8, 1024, 32, 32
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, R, R, cin, device="cuda", generator=g).bfloat16()
w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / math.sqrt(9 * cin)
wp = ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16)
s = 1 + 0.3 * torch.randn(B, cin, device="cuda"); d = 0.5 + torch.rand(B, cout, device="cuda")
nz = torch.randn(1, R, R, device="cuda"); nw = torch.ones(1, device="cuda"); bias = torch.randn(cout, device="cuda")
for _ in range(3):
    y = ops.conv2d(x, wp, cout, 3, in_scale=s, out_scale=d, bias=bias, noise=nz, noise_w=nw, act=1, gain=1.414)
torch.cuda.synchronize()
t = y.view(-1)[:80 * 4].view(torch.int64).cpu().view(16, 5)
names = ["wait", "issue", "kloop", "epi", "next"]
prev = None
for i in range(16):
    row = t[i].tolist()
    d = [row[k + 1] - row[k] for k in range(4)]
    gap = (row[0] - prev) if prev is not None else 0
    prev = row[4]
    print(f"step {40+i}: wait {d[0]:6d}  issue {d[1]:6d}  kloop {d[2]:6d}  epi+store {d[3]:6d}   total {row[4]-row[0]:6d}  gap {gap}")
