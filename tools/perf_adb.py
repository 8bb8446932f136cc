"""A/B of the 64-wide conv_igemm tiles with / without the double-buffered halo tile (dev tool):
   DGE_CONV_DBG=64 python tools/perf_adb.py   (old)   vs   python tools/perf_adb.py   (new)
Shapes: the deep-K launches of a step that run on the 64-wide tile (LPIPS conv3-5 on the crops, encoder data gradients
at <= 64^2 and 128->64 @256^2, the layer-15 adjoint of the generator)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops


def timeit(run, n=20):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def fwd(B, cin, cout, H):
    x = torch.randn(B, H, H, cin, device="cuda").bfloat16()
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    wp = ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16)
    bias = torch.randn(cout, device="cuda")
    t = timeit(lambda: ops.conv2d(x, wp, cout, 3, bias=bias, act=ops.ACT_RELU))
    return t


def dgrad(B, cin, cout, H):
    g = torch.randn(B, H, H, cout, device="cuda").bfloat16()
    xin = torch.randn(B, H, H, cin, device="cuda").bfloat16()
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    wp = ops.pack_conv_weight(w, ops.PACK_DGRAD, ops.BF16)

    def run():
        st = ops.SlotStats(B, cin, g.device)
        return ops.conv2d(g, wp, cin, 3, stats=st, dot_src=xin)
    return timeit(run)


print("DGE_CONV_DBG =", os.environ.get("DGE_CONV_DBG", "0"))
for (B, ci, co, H) in [(16, 256, 256, 48), (16, 512, 512, 32), (16, 256, 256, 32), (16, 512, 512, 16), (16, 128, 128, 64)]:
    t = fwd(B, ci, co, H)
    print(f"fwd   B={B} {ci}->{co} @{H}^2: {t:7.1f} us  {2*9*ci*co*H*H*B/t/1e6:7.1f} TF/s")
for (B, ci, co, H) in [(8, 64, 128, 256), (8, 256, 512, 64), (8, 256, 256, 64), (8, 512, 512, 32)]:
    t = dgrad(B, ci, co, H)
    print(f"dgrad B={B} {co}->{ci} @{H}^2: {t:7.1f} us  {2*9*ci*co*H*H*B/t/1e6:7.1f} TF/s")

import math
from dge_amd._lib import last_kernel
Bq = 8
for cof, cif, R in [(32, 64, 512)]:          # layer-15 adjoint: in_s2d, Cin = 128 logical, 64-wide tile, 8192 tiles of 16 x 16
    gz = torch.randn(Bq, 2 * R, 2 * R, cof, device="cuda").to(torch.bfloat16)
    x = torch.randn(Bq, R, R, cif, device="cuda").to(torch.bfloat16)
    add = torch.randn(Bq, R, R, cif, device="cuda").to(torch.bfloat16)
    s = 1 + 0.3 * torch.randn(Bq, cif, device="cuda")
    w = torch.randn(cof, cif, 3, 3, device="cuda")
    pk = ops.pack_conv_weight(w, ops.PACK_UPFOLD_DGRAD, ops.BF16, 1 / math.sqrt(9 * cif))
    st = ops.SlotStats(Bq, cif, "cuda")
    for name, kw in {"full": dict(out_scale=s, addend=add, stats=st, dot_src=x), "plain": dict()}.items():
        t = timeit(lambda: ops.conv2d(gz, pk, cif, 3, in_s2d=True, **kw), 10)
        print(f"s2d adjoint {cof}->{cif}@{R} {name:6s} {t:8.1f} us  {last_kernel()}")
for (B, ci, co, H) in [(16, 64, 64, 128), (16, 128, 64, 128), (8, 128, 64, 256)]:
    t = fwd(B, ci, co, H)
    print(f"fwd   B={B} {ci}->{co} @{H}^2: {t:7.1f} us  {2*9*ci*co*H*H*B/t/1e6:7.1f} TF/s  {last_kernel()}")
for (B, ci, co, H) in [(8, 128, 128, 256), (8, 256, 256, 128), (8, 512, 512, 64), (16, 128, 128, 128), (16, 256, 256, 64), (16, 512, 512, 32)]:
    t = fwd(B, ci, co, H)
    print(f"fwd   B={B} {ci}->{co} @{H}^2: {t:7.1f} us  {2*9*ci*co*H*H*B/t/1e6:7.1f} TF/s  {last_kernel()}")
