import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
B, cin, cout, H = [int(v) for v in sys.argv[1:5]]   # forward conv cin->cout; dgrad maps cout->cin
mode = sys.argv[5] if len(sys.argv) > 5 else "dot"
g = torch.randn(B, H, H, cout, device="cuda").bfloat16()
xin = torch.randn(B, H, H, cin, device="cuda").bfloat16()
w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
wp = ops.pack_conv_weight(w, ops.PACK_DGRAD, ops.BF16)
s = torch.rand(B, cin, device="cuda") + 0.5
add = torch.randn(B, H, H, cin, device="cuda").bfloat16()
def run():
    st = torch.zeros(B, cin, 2, device="cuda")
    if mode == "dot":
        return ops.conv2d(g, wp, cin, 3, out_scale=s, stats=st, dot_src=xin, addend=add)
    if mode == "plain":
        return ops.conv2d(g, wp, cin, 3)
    return ops.conv2d(g, wp, cin, 3, out_scale=s, stats=st)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); N = 10
for _ in range(N): run()
e1.record(); torch.cuda.synchronize()
print(f"dgrad[{mode}] B={B} {cout}->{cin} H={H}: {e0.elapsed_time(e1)/N*1e3:.1f} us")
