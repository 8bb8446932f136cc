"""Time the encoder's 1x1 launches (BEBlock conv_3 forward with addend + statistics, and its data gradient) at batch 8:
python tools/perf_pw.py   (DGE_NO_PW=1 keeps them on conv_igemm: run both for the A/B)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
from dge_amd._lib import last_kernel
B = 8


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, cout, R in ((16, 32, 512), (32, 64, 256), (64, 128, 128)):
    x = torch.randn(B, R, R, cin, device="cuda").bfloat16()
    a = torch.randn(B, R, R, cout, device="cuda").bfloat16()
    w = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5
    bias = torch.randn(cout, device="cuda")
    wp = ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16, 1.0)
    wd = ops.pack_conv_weight(w, ops.PACK_DGRAD, ops.BF16, 1.0)
    st = ops.SlotStats(B, cout, "cuda")
    t = timeit(lambda: ops.conv2d(x, wp, cout, 1, bias=bias, gain=0.889, addend=a, add_scale=0.111, stats=st))
    k = last_kernel()
    mb = B * R * R * (cin + 2 * cout) * 2 / 1e6
    print(f"fwd {cin}->{cout} @{R}: {t:.1f} us  {mb:.0f} MB  {mb / t / 1e3 * 1e3:.0f} GB/s  {k}")
    t1 = timeit(lambda: ops.conv2d(x, wp, cout, 1, bias=bias, gain=0.889, addend=a, add_scale=0.111))
    t2 = timeit(lambda: ops.conv2d(x, wp, cout, 1, bias=bias, gain=0.889))
    print(f"    without statistics {t1:.1f} us, without addend too {t2:.1f} us; slots {st.nslot}")
    t = timeit(lambda: ops.conv2d(a, wd, cin, 1, gain=0.889))
    k = last_kernel()
    mb = B * R * R * (cin + cout) * 2 / 1e6
    print(f"dgrad {cout}->{cin} @{R}: {t:.1f} us  {mb:.0f} MB  {mb / t / 1e3 * 1e3:.0f} GB/s  {k}")
