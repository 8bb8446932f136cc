"""Per-job clock stamps of a conv_stream launch (tuning build: tools/build_variant.sh NAME "-DDGE_SC_TIMING -DDGE_SC_ONLY=ci,co" main):
prologue / row-loop cycles per job and the spread of the jobs' end times (all jobs of a launch are resident at once: the slowest one
sets the launch time).   python tools/perf_stream_jobs.py cin cout R flavour      flavour: gen | enc | stats | pool | dot"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
cin, cout, R, fl = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
B = 8
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, R, R, cin, device="cuda", generator=g).bfloat16()
w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / math.sqrt(9 * cin)
wp = ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16)
sc = 0.5 + torch.rand(B, cin, device="cuda"); sh = torch.randn(B, cin, device="cuda")
nz = torch.randn(B, R, R, device="cuda"); nw = torch.randn(cout, device="cuda"); bias = torch.randn(cout, device="cuda")
if fl == "gen":
    fn = lambda: ops.conv2d(x, wp, cout, 3, in_scale=sc, out_scale=0.5 + torch.rand(B, cout, device="cuda"), bias=bias, noise=nz[:1], noise_w=torch.ones(1, device="cuda"), act=1, gain=1.414)
elif fl == "enc":
    fn = lambda: ops.conv2d(x, wp, cout, 3, in_scale=sc, in_shift=sh, noise=nz, noise_w=nw, bias=bias, act=1)
elif fl == "stats":
    fn = lambda: ops.conv2d(x, wp, cout, 3, in_scale=sc, in_shift=sh, noise=nz, noise_w=nw, bias=bias, act=1, stats=ops.SlotStats(B, cout, x.device))
elif fl == "pool":
    fn = lambda: ops.conv2d(x, wp, cout, 3, in_scale=sc, in_shift=sh, noise=nz, noise_w=nw, bias=bias, act=1, pool_out=True, pool_mask=True)
else:
    raise SystemExit("flavour")
for _ in range(3):
    y = fn()
torch.cuda.synchronize()
if isinstance(y, tuple): y = y[0]
n = 4096
t = y.view(-1)[:n * 16].view(torch.int64).cpu().view(n, 4)
rows = (t[:, 3] >> 40)
ok = (rows > 0) & (rows < 4096) & (t[:, 2] > t[:, 0])
t = t[ok]; rows = rows[ok]
pro = (t[:, 1] - t[:, 0]).float(); loop = (t[:, 2] - t[:, 1]).float(); tot = (t[:, 2] - t[:, 0]).float()
rt = (t[:, 3] & 0xffffffffff).float()
print(f"{fl} {cin}->{cout} @{R}: {len(t)} jobs, rows/job {rows.float().mean():.1f}")
print(f"  prologue cycles: mean {pro.mean():.0f}  max {pro.max():.0f}")
print(f"  loop cycles/row: mean {(loop / rows.float()).mean():.0f}  p90 {(loop / rows.float()).quantile(0.9):.0f}  max {(loop / rows.float()).max():.0f}")
print(f"  job total cycles: mean {tot.mean():.0f}  max {tot.max():.0f}   (max / mean {tot.max() / tot.mean():.2f})")
print(f"  end-time spread (100 MHz ticks): {(rt.max() - rt.min()):.0f} = {(rt.max() - rt.min()) / 100:.1f} us; job length at 100 MHz n/a")
