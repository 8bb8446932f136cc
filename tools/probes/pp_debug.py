import os, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import torch
from dge_amd import ops
B, cin, cout, H = 8, 128, 128, 256
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, H, H, cin, device="cuda", generator=g).bfloat16()
w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g)
wscale = 1.0 / (9 * cin) ** 0.5
s = 1.0 + 0.3 * torch.randn(B, cin, device="cuda", generator=g)
d = 0.5 + torch.rand(B, cout, device="cuda", generator=g)
bias = torch.randn(cout, device="cuda", generator=g)
nz = torch.randn(1, H, H, device="cuda", generator=g)
nw = torch.full((1,), 0.3, device="cuda")
wp = ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16, wscale)
y0 = ops.conv2d(x, wp, cout, 3, in_scale=s, out_scale=d, bias=bias, noise=nz, noise_w=nw, act=1, gain=2 ** 0.5).float()
wpp = ops.pack_conv_pp(w, wscale, in_scale=s, out_scale=d, gain=2 ** 0.5)
y1 = ops.conv_pp(x, wpp, cout, bias=bias, noise=nz, noise_w=nw, act=1, gain=2 ** 0.5).float()
torch.cuda.synchronize()
df = (y0 - y1).abs()
print("max", df.max().item(), "ref max", y0.abs().max().item())
bad = df > 0.05
print("bad frac", bad.float().mean().item())
print("by sample", bad.float().mean(dim=(1, 2, 3)).tolist())
print("by row%16", bad[0].float().mean(dim=(1, 2)).view(-1, 16).mean(0).tolist())
print("by col%32", bad[0].float().mean(dim=(0, 2)).view(-1, 32).mean(0).tolist())
print("by chan", [round(v, 2) for v in bad[0].float().mean(dim=(0, 1)).tolist()])
print(y0[0, 0, 0, :8].tolist()); print(y1[0, 0, 0, :8].tolist())
print(y0[0, 5, 17, 32:40].tolist()); print(y1[0, 5, 17, 32:40].tolist())
