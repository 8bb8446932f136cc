import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
B, cin, cout, H, k = [int(v) for v in sys.argv[1:6]]
x = torch.randn(B, H, H, cin, device="cuda").bfloat16()
g = torch.randn(B, H, H, cout, device="cuda").bfloat16()
sc = torch.rand(B, cin, device="cuda") + 0.5; sh = torch.randn(B, cin, device="cuda")
dw = torch.zeros(cout, cin, k, k, device="cuda")
for _ in range(3): ops.conv_wgrad(g, x, dw, sc, sh)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 10
for _ in range(N): ops.conv_wgrad(g, x, dw, sc, sh)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / N
fl = 2 * k * k * cin * cout * H * H * B
byts = (x.numel() + g.numel()) * 2
from dge_amd._lib import last_kernel
print(f"wgrad B={B} {cin}->{cout} H={H} k={k}: {t*1e3:.1f} us  {fl/t/1e9:.1f} TF/s  {byts/t/1e6:.0f} GB/s  {last_kernel()}")
