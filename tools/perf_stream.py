"""A/B timing of the streaming conv (csrc/conv_stream.hip) against conv_igemm on the HBM-bound launch families of the step
(dev tool, GPU box): python tools/perf_stream.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
from dge_amd._lib import last_kernel
DEV = "cuda"


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(kind, B, R, cin, cout):
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, R, R, cin, device=DEV, generator=g).bfloat16()
    w = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / math.sqrt(9 * cin)
    nbytes = x.numel() * 2 + B * R * R * cout * 2
    if kind == "g":
        wp = ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16)
        s = 1 + 0.3 * torch.randn(B, cin, device=DEV); d = 0.5 + torch.rand(B, cout, device=DEV)
        nz = torch.randn(1, R, R, device=DEV); nw = torch.ones(1, device=DEV); bias = torch.randn(cout, device=DEV)
        fn = lambda: ops.conv2d(x, wp, cout, 3, in_scale=s, out_scale=d, bias=bias, noise=nz, noise_w=nw, act=1, gain=1.414)
    elif kind in ("enc", "encs"):
        wp = ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16)
        sc = 0.5 + torch.rand(B, cin, device=DEV); sh = torch.randn(B, cin, device=DEV)
        nz = torch.randn(B, R, R, device=DEV); nw = torch.randn(cout, device=DEV); bias = torch.randn(cout, device=DEV)
        def fn():
            st = ops.SlotStats(B, cout, DEV) if kind == "encs" else None
            return ops.conv2d(x, wp, cout, 3, in_scale=sc, in_shift=sh, noise=nz, noise_w=nw, bias=bias, act=1, stats=st)
    else:   # dgrad with dot statistics
        wp = ops.pack_conv_weight(w.permute(1, 0, 2, 3).contiguous(), ops.PACK_DGRAD, ops.BF16)
        xin = torch.randn(B, R, R, cout, device=DEV).bfloat16()
        s = 1 + 0.3 * torch.randn(B, cout, device=DEV)
        nbytes += xin.numel() * 2
        def fn():
            st = ops.SlotStats(B, cout, DEV)
            return ops.conv2d(x, wp, cout, 3, stats=st, dot_src=xin, out_scale=s)
    os.environ.pop("DGE_NO_STREAM", None)
    t_new = timeit(fn); k_new = last_kernel()
    os.environ["DGE_NO_STREAM"] = "1"
    t_old = timeit(fn); k_old = last_kernel()
    os.environ.pop("DGE_NO_STREAM", None)
    fl = 2 * 9 * cin * cout * R * R * B
    print(f"{kind:5s} B{B} {cin:3d}->{cout:3d} @{R:4d}: stream {t_new:7.1f} us ({nbytes/t_new/1e6:6.2f} TB/s, {fl/t_new/1e6:6.0f} TF)  igemm {t_old:7.1f} us   "
          f"floor@8TB/s {nbytes/8e6:6.1f} us   [{k_new} | {k_old}]")


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for args in [("g", B, 1024, 32, 32), ("g", B, 512, 64, 64), ("dot", B, 1024, 32, 32), ("dot", B, 512, 64, 64),
                 ("encs", B, 1024, 16, 16), ("enc", B, 1024, 16, 32), ("dot", B, 1024, 16, 16), ("dot", B, 1024, 32, 16),
                 ("encs", B, 512, 32, 32), ("enc", B, 512, 32, 64), ("dot", B, 512, 32, 32), ("dot", B, 512, 64, 32),
                 ("encs", B, 256, 64, 64), ("dot", B, 256, 64, 64), ("g", 2 * B, 256, 64, 64), ("g", 2 * B, 256, 16, 64)]:
        case(*args)
