"""Shader-clock timeline of up_s4's K loop (dev tool; needs a -DDGE_UP_TIMING build): wave 0 of workgroups 0 and 8, cycles between stamps per chunk.
tags: 1 chunk top, 2 own requests landed (vmcnt 0), 3 barrier passed, 4 cluster A fragments in, 5 cluster A MFMAs + DMA issued / B reads issued,
6 B fragments in, 7 B MFMAs issued / C reads issued, 8 C fragments in"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math, torch
from dge_amd import ops
from dge_amd._lib import lib
B, H, cin, cout = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (8, 32, 512, 512))]
x = torch.randn(B, H, H, cin, device="cuda").bfloat16()
w = torch.randn(cout, cin, 3, 3, device="cuda")
s = 1.0 + 0.3 * torch.randn(B, cin, device="cuda"); d = 0.5 + torch.rand(B, cout, device="cuda")
bias = torch.randn(cout, device="cuda"); nz = torch.randn(1, 2 * H, 2 * H, device="cuda"); nw = torch.full((1,), 0.3, device="cuda")
wimg = ops.pack_up_pp(ops.pack_upconv_weight(w, ops.BF16, 1.0 / math.sqrt(9 * cin)), cout, cin, in_scale=s, out_scale=d, gain=math.sqrt(2.0))
for _ in range(3):
    y = ops.up_pp(x, wimg, cout, bias=bias, bias_scale=1.0, noise=nz, noise_w=nw, act=ops.ACT_LRELU, gain=math.sqrt(2.0))
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 2048)()
f = lib().dge_dbg_up_prof
f.argtypes = [ctypes.c_void_p]
assert f(buf) == 0
for g in range(2):
    n = buf[g * 1024 + 1023]
    ev = [(buf[g * 1024 + i] >> 48, buf[g * 1024 + i] & ((1 << 48) - 1)) for i in range(n)]
    print(f"== workgroup {0 if g == 0 else 8}: {n} stamps, shape B={B} H={H} {cin}->{cout}; first stamp at {ev[0][1]}")
    prev, line = ev[0][1], []
    for tag, t in ev:
        if tag == 1 and line:
            print("  ", " ".join(line)); line = []
        line.append(f"[{tag}]+{t - prev}")
        prev = t
    print("  ", " ".join(line))
