"""conv_pp data-gradient form with an EMPTY epilogue menu (no dot_src / scale / prep) against conv_igemm: isolates the staging (dev probe)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dge_amd import ops
B, cof, cif, R = 8, 128, 128, 256
g = torch.Generator(device="cuda").manual_seed(1)
gz = torch.randn(B, R, R, cof, device="cuda", generator=g).bfloat16()
w = torch.randn(cof, cif, 3, 3, device="cuda", generator=g)
wscale = 1.0 / (9 * cif) ** 0.5
xin = (1.5 * torch.randn(B, R, R, cif, device="cuda", generator=g)).bfloat16()
s = 1.0 + 0.3 * torch.randn(B, cif, device="cuda", generator=g)
wp = ops.pack_conv_weight(w, ops.PACK_DGRAD, ops.BF16, wscale)
wpp = ops.pack_conv_pp(w, wscale, dgrad=True)
for mode in ("plain", "scale", "dot"):
    kw = {} if mode == "plain" else (dict(out_scale=s) if mode == "scale" else dict(out_scale=s, stats=ops.SlotStats(B, cif, "cuda"), dot_src=xin))
    kw0 = dict(kw)
    if mode == "dot":
        kw0["stats"] = ops.SlotStats(B, cif, "cuda")
    y0 = ops.conv2d(gz, wp, cif, 3, **kw0).float()
    y1 = ops.conv_pp(gz, wpp, cif, dgrad=True, **kw).float()
    bad = ((y0 - y1).abs() > 0.02 * y0.abs().max())
    print(mode, "bad fraction", bad.float().mean().item())
    if bad.any():
        idx = bad.nonzero()
        print("  by row%16", torch.bincount(idx[:, 1] % 16, minlength=16).tolist())
        print("  by col%32", torch.bincount(idx[:, 2] % 32, minlength=32).tolist())
        print("  by ch//8", torch.bincount(idx[:, 3] // 8, minlength=16).tolist())
