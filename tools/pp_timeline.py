"""Shader-clock timeline of one workgroup of conv_pp (dev tool; DGE_CONV_DBG bit 256): per phase the cycles spent in the LOAD part,
at the waits, at the barriers and in the MFMA part, for wave 0 (group 0) and wave 4 (group 1)."""
import ctypes, os, sys
os.environ["DGE_CONV_DBG"] = str(256 | int(os.environ.get("PP_EXTRA", "0")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
from dge_amd._lib import lib
B, cin, cout, H = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (8, 128, 128, 256))]
x = torch.randn(B, H, H, cin, device="cuda").bfloat16()
w = torch.randn(cout, cin, 3, 3, device="cuda")
s = 1.0 + 0.3 * torch.randn(B, cin, device="cuda"); d = 0.5 + torch.rand(B, cout, device="cuda")
bias = torch.randn(cout, device="cuda"); nz = torch.randn(1, H, H, device="cuda"); nw = torch.full((1,), 0.3, device="cuda")
wpp = ops.pack_conv_pp(w, 1.0 / (9 * cin) ** 0.5, in_scale=s, out_scale=d, gain=2 ** 0.5)
DG = len(sys.argv) > 5 and sys.argv[5] == "dg"
if DG:      # the data-gradient form with the fused tail backward (x plays g_z; cin -> cout is the GEMM's K -> N)
    wpp = ops.pack_conv_pp(w, 1.0 / (9 * cout) ** 0.5, in_scale=s, dgrad=True)
    xin = torch.randn(B, H, H, cout, device="cuda").bfloat16(); so = 1.0 + 0.3 * torch.randn(B, cout, device="cuda")
    w = w.transpose(0, 1).contiguous()
    wpp = ops.pack_conv_pp(w, 1.0 / (9 * cout) ** 0.5, in_scale=s, dgrad=True)
for _ in range(3):
    if DG:
        st, P = ops.SlotStats(B, cout, "cuda"), ops.SlotStats(B, cout, "cuda")
        y = ops.conv_pp(x, wpp, cout, dgrad=True, out_scale=so, stats=st, dot_src=xin, prep=dict(gain=2 ** 0.5, noise=nz, ns=nw, stats=P))
    else:
        y = ops.conv_pp(x, wpp, cout, bias=bias, noise=nz, noise_w=nw, act=1, gain=2 ** 0.5)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 2048)()
f = lib().dge_dbg_pp_prof
f.argtypes = [ctypes.c_void_p]
assert f(buf) == 0
for g in range(2):
    n = buf[g * 1024 + 1023]
    ev = [(buf[g * 1024 + i] >> 48, buf[g * 1024 + i] & ((1 << 48) - 1)) for i in range(n)]
    print(f"== group {g}: {n} stamps")
    t0 = ev[0][1]
    line, prev = [], ev[0][1]
    names = {1: "L", 2: "w", 3: "b", 4: "M", 5: "E", 6: "e"}
    out = []
    for tag, t in ev:
        out.append(f"{names.get(tag, '?')}{t - prev}")
        prev = t
        if tag in (4, 6):
            line.append(" ".join(out)); out = []
    nch = cin // 32
    for i, l in enumerate(line[: 9 * nch * 3 + 4]):
        print(f"{i:3d} {l}")
    print("total cycles", ev[-1][1] - t0)
