"""Clock stamps of one workgroup's tile loop in wgrad_dma (tuning build: tools/build_variant_wg.sh NAME -DDGE_WG_TIMING):
   python tools/perf_wgrad_timing.py B cin cout H"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
from dge_amd._lib import last_kernel, LIB_PATH
B, cin, cout, H = [int(v) for v in sys.argv[1:5]]
x = torch.randn(B, H, H, cin, device="cuda").bfloat16()
g = torch.randn(B, H, H, cout, device="cuda").bfloat16()
sc = torch.rand(B, cin, device="cuda") + 0.5; sh = torch.randn(B, cin, device="cuda")
dw = torch.zeros(cout, cin, 3, 3, device="cuda")
for _ in range(3): ops.conv_wgrad(g, x, dw, sc, sh)
torch.cuda.synchronize()
raw = C.CDLL(LIB_PATH)
buf = (C.c_longlong * (32 * 6))()
raw.dge_wgrad_tlog(buf, 32 * 6)
names = ["wait", "barrier", "issue", "mfma"]
tot = [0] * 4; n = 0; prev = None; gaps = 0
for i in range(32):
    r = [buf[i * 6 + k] for k in range(5)]
    if r[4] <= r[0] or r[0] == 0: continue
    d = [r[k + 1] - r[k] for k in range(4)]
    if prev is not None: gaps += r[0] - prev
    prev = r[4]; n += 1
    for k in range(4): tot[k] += d[k]
    if i < 6: print("tile", i, dict(zip(names, d)), "total", r[4] - r[0])
print(last_kernel(), "mean cycles per tile:", {nm: tot[k] // max(n, 1) for k, nm in enumerate(names)}, "gap", gaps // max(n - 1, 1), "tiles", n)
