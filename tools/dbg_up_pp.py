"""debug aid for csrc/up_pp.hip: raw t (DGE_UP_DBG=16) against torch's conv_transpose2d on a small shape"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DGE_NO_UPSTREAM"] = "1"; os.environ["DGE_UP_PP_MIN_TILES"] = "1"
os.environ["DGE_UP_DBG"] = sys.argv[1] if len(sys.argv) > 1 else "16"
import torch
import torch.nn.functional as F
import dge_amd
from dge_amd import ops
DEV = "cuda"
g = torch.Generator(device=DEV); g.manual_seed(1)
B, H, W, cin, cout = 1, 32, 32, 128, 32
def tref(x, w):
    xn = x.float().permute(0, 3, 1, 2)
    t = F.conv_transpose2d(xn, w.flip(2, 3).permute(1, 0, 2, 3).contiguous(), stride=2)     # [B, Cout, 2H+1, 2W+1]
    return t[:, :, :2 * H, :2 * W].permute(0, 2, 3, 1)
def run(tag, x, w):
    wu = ops.pack_upconv_weight(w, ops.BF16, 1.0)
    wimg = ops.pack_up_pp(wu, cout, cin, gain=1.0)
    y1 = ops.up_pp(x, wimg, cout).float()
    torch.cuda.synchronize()
    y0 = tref(x, w.to(torch.bfloat16).float())
    diff = (y1 - y0).abs(); mx = y0.abs().max().item()
    bad = diff > 0.03 * mx + 1e-3
    print(f"[{tag}] max|diff| {diff.max().item():.4f} of {mx:.3f}; frac bad {bad.float().mean().item():.4f}")
    if bad.any():
        b4 = bad[0]
        print("   by (row%2, col%2):", [[round(b4[r::2, c::2].float().mean().item(), 3) for c in range(2)] for r in range(2)])
        print("   by row:", [round(b4[r].float().mean().item(), 1) for r in range(2 * H)])
        print("   by col:", [round(b4[:, c].float().mean().item(), 1) for c in range(2 * W)])
        print("   by chan:", [round(b4[:, :, c].float().mean().item(), 2) for c in range(cout)])
        i = bad.nonzero()[0].tolist()
        print("   first bad", i, y0[tuple(i)].item(), y1[tuple(i)].item())
    return y0, y1
x = (torch.randn(B, H, W, cin, device=DEV, generator=g) * 0.7).to(torch.bfloat16)
w = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / math.sqrt(9 * cin)
y0, y1 = run("plain", x, w)
# x = delta in channel k0 at pixel (5, 7), w = identity-like: out channel o <- in channel o, single tap
for (wy, wx) in ((1, 1), (2, 2), (0, 0), (2, 1), (1, 0)):
    xd = torch.zeros(B, H, W, cin, device=DEV, dtype=torch.bfloat16); xd[0, 5, 7, :32] = 1.0
    wd = torch.zeros(cout, cin, 3, 3, device=DEV)
    for o in range(cout): wd[o, o, wy, wx] = 1.0 + o / 64.0
    y0, y1 = run(f"delta tap({wy},{wx})", xd, wd)
    nz0 = (y0[0].abs() > 0.1).nonzero(); nz1 = (y1[0].abs() > 0.1).nonzero()
    print("   ref nonzero (oy, ox, c):", nz0[:6].tolist(), "n", len(nz0)); print("   got nonzero:", nz1[:12].tolist(), "n", len(nz1))
    if len(nz1): print("   got values", [round(y1[0][tuple(i)].item(), 3) for i in nz1[:12].tolist()])
# chunk test: which K chunk contributes
for kc in range(cin // 32):
    w1 = torch.zeros_like(w); w1[:, kc * 32:(kc + 1) * 32] = w[:, kc * 32:(kc + 1) * 32]
    run(f"chunk{kc}", x, w1)
