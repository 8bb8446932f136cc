"""Where does the conv-epilogue statistics sum differ from the oracle?  (run on the GPU box)"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dge_amd
from dge_amd import ops
from oracle import conv_ref as CR
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(3000 + 32 + 32 + 512)
B, R, cin, cout = 8, 512, 32, 32
x = (torch.randn(B, R, R, cin, device=DEV, generator=g)).to(torch.bfloat16)
w = (torch.randn(cout, cin, 3, 3, device=DEV, generator=g).to(torch.bfloat16).float() * (1.0 / math.sqrt(9 * cin))).to(torch.bfloat16).float()
sc = 0.5 + torch.rand(B, cin, device=DEV, generator=g)
sh = 0.3 * torch.randn(B, cin, device=DEV, generator=g)
noise = torch.randn(B, R, R, device=DEV, generator=g)
nw = 0.1 * torch.randn(cout, device=DEV, generator=g)
bias = 0.1 * torch.randn(cout, device=DEV, generator=g)
for nslot_force in (None,):
    st = ops.SlotStats(B, cout, DEV)
    y = ops.conv2d(x, ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16, 1.0), cout, 3, in_scale=sc, in_shift=sh, noise=noise,
                   noise_w=nw, bias=bias, act=ops.ACT_LRELU, stats=st)
    tot = st.buf.double().sum(0).cpu()
b = 0
xb = x[b:b+1].float().permute(0, 3, 1, 2).cpu()
ref, rs, rq = CR.enc_conv(xb, w.cpu(), sc[b:b+1].cpu(), sh[b:b+1].cpu(), noise[b:b+1].cpu(), nw.cpu(), bias.cpu())
xn = (xb * sc[b].cpu()[None, :, None, None] + sh[b].cpu()[None, :, None, None]).bfloat16().double()
yy = torch.nn.functional.conv2d(xn, w.cpu().double(), padding=1) + (nw.cpu()[None, :, None, None] * noise[b:b+1].cpu()[:, None]).double() + bias.cpu().double()[None, :, None, None]
yy = torch.where(yy > 0, yy, 0.2 * yy)
emu = yy.sum((2, 3))[0]
ysum = y[b].double().sum((0, 1)).cpu()
print("kernel - emu(f64, bf16-rounded xn):", (tot[b, :, 0] - emu)[:8])
print("kernel - oracle f32              :", (tot[b, :, 0] - rs[0].double())[:8])
print("sum(y bf16) - emu                :", (ysum - emu)[:8])
print("emu - oracle                     :", (emu - rs[0].double())[:8])
print("max elem |y - emu|", (y[b].double().permute(2, 0, 1).cpu() - yy[0]).abs().max().item())
