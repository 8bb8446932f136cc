"""where does the conv_pp data-gradient output differ from conv_igemm's? (dev probe)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dge_amd import ops
B, cof, cif, R = 8, 128, 128, 256
g = torch.Generator(device="cuda").manual_seed(1)
gz = torch.randn(B, R, R, cof, device="cuda", generator=g).bfloat16()
d = 0.5 + torch.rand(B, cof, device="cuda", generator=g)
xin = (1.5 * torch.randn(B, R, R, cif, device="cuda", generator=g)).bfloat16()
w = torch.randn(cof, cif, 3, 3, device="cuda", generator=g)
wscale = 1.0 / (9 * cif) ** 0.5
s = 1.0 + 0.3 * torch.randn(B, cif, device="cuda", generator=g)
nz = torch.randn(1, R, R, device="cuda", generator=g); ns = torch.full((1,), 0.37, device="cuda")
wp = ops.pack_conv_weight(w, ops.PACK_DGRAD, ops.BF16, wscale)
wpp = ops.pack_conv_pp(w, wscale, in_scale=d, dgrad=True)
st, P = ops.SlotStats(B, cif, "cuda"), ops.SlotStats(B, cif, "cuda")
y0 = ops.conv2d(gz, wp, cif, 3, in_scale=d, out_scale=s, stats=st, dot_src=xin, prep=dict(gain=2 ** 0.5, noise=nz, ns=ns, stats=P)).float()
for it in range(3):
    st, P = ops.SlotStats(B, cif, "cuda"), ops.SlotStats(B, cif, "cuda")
    y1 = ops.conv_pp(gz, wpp, cif, dgrad=True, out_scale=s, stats=st, dot_src=xin, prep=dict(gain=2 ** 0.5, noise=nz, ns=ns, stats=P)).float()
    bad = ((y0 - y1).abs() > 0.05 * y0.abs().max())
    print("iteration", it, "bad fraction", bad.float().mean().item())
    if bad.any():
        idx = bad.nonzero()
        print("by sample", torch.bincount(idx[:, 0], minlength=B).tolist())
        print("by row%16", torch.bincount(idx[:, 1] % 16, minlength=16).tolist())
        print("by col%32", torch.bincount(idx[:, 2] % 32, minlength=32).tolist())
        print("by ch//8", torch.bincount(idx[:, 3] // 8, minlength=16).tolist())
        print("by tile row", torch.bincount(idx[:, 1] // 16, minlength=16).tolist())
        tid = (idx[:, 0] * 16 + idx[:, 1] // 16) * 8 + idx[:, 2] // 32
        print("by k (position in the workgroup's tile sequence)", torch.bincount((tid % 128) // 32, minlength=4).tolist())
        print("by wave q (row%16//4)", torch.bincount((idx[:, 1] % 16) // 4, minlength=4).tolist())
        k = idx[0].tolist(); print("first", k, y0[tuple(k)].item(), y1[tuple(k)].item())
        # is the wrong value the right value of some neighbour (same channel)?
        for kk in idx[:6].tolist():
            bb, yy, xx, cc = kk
            v = y1[bb, yy, xx, cc].item()
            hits = (y0[bb, max(0, yy - 16):yy + 17, max(0, xx - 32):xx + 33, :] == v).nonzero()
            print("  ", kk, "got", v, "want", y0[bb, yy, xx, cc].item(), "same value at (dy,dx,c):", [(h[0].item() - min(16, yy), h[1].item() - min(32, xx), h[2].item()) for h in hits[:4]])
