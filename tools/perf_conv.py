"""Time one conv configuration (dev tool): python tools/perf_conv.py B Cin Cout H k [up]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
B, cin, cout, H, k = [int(v) for v in sys.argv[1:6]]
up = len(sys.argv) > 6 and sys.argv[6] == "up"
x = torch.randn(B, H, H, cin, device="cuda").bfloat16()
w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
wp = ops.pack_conv_weight(w, ops.PACK_UPFOLD if up else ops.PACK_FWD, ops.BF16)
s = torch.randn(B, cin, device="cuda"); d = torch.rand(B, cout, device="cuda") + 0.5
bias = torch.randn(cout, device="cuda"); nz = torch.randn(1, H * (2 if up else 1), H * (2 if up else 1), device="cuda"); nw = torch.ones(1, device="cuda")
def run():
    return ops.conv2d(x, wp, cout, k, up=up, in_scale=s, out_scale=d, bias=bias, noise=nz, noise_w=nw, act=1, gain=1.414)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 20
for _ in range(N): run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / N
fl = 2 * k * k * cin * cout * H * H * B
print(f"DBG={os.environ.get('DGE_CONV_DBG','0')} B={B} {cin}->{cout} H={H} k={k} up={up}: {t*1e3:.1f} us  {fl/t/1e9:.1f} TF/s")
