"""Phase timestamps of one wave of conv_igemm (tuning build -DDGE_IGEMM_TIMING): python tools/perf_igemm_timing.py B Cin Cout H"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
B, cin, cout, H = [int(v) for v in sys.argv[1:5]]
x = torch.randn(B, H, H, cin, device="cuda").bfloat16()
w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
wp = ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16)
s = torch.randn(B, cin, device="cuda"); d = torch.rand(B, cout, device="cuda") + 0.5
bias = torch.randn(cout, device="cuda"); nz = torch.randn(1, H, H, device="cuda"); nw = torch.ones(1, device="cuda")
for _ in range(3):
    y = ops.conv2d(x, wp, cout, 3, in_scale=s, out_scale=d, bias=bias, noise=nz, noise_w=nw, act=1, gain=1.414)
torch.cuda.synchronize()
t = y.view(-1)[:98 * 4].view(torch.int64).cpu()
st = t[:96].view(16, 6)
for i in range(16):
    r = st[i].tolist()
    if r[0] == 0: break
    print(f"stage {i:2d}: dma_issue {r[1]-r[0]:5d}  halo_load {r[2]-r[1]:5d}  mfma {r[3]-r[2]:6d}  vmwait {r[4]-r[3]:5d}  barrier(+store_a) {r[5]-r[4]:6d}   total {r[5]-r[0]:6d}")
print("main loop", st[11][5].item() - st[0][0].item() if st[11][0] else "?", " epilogue", (t[97] - t[96]).item())
