"""Up-layer timing, folded conv2d(up=True) vs upconv_fir (dev tool): python tools/perf_up.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dge_amd
from dge_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt = ops.BF16
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("  H   Cin Cout   folded_us    new_us   speedup   alg TF/s(new)")
for H, Cin, Cout in ((4, 512, 512), (8, 512, 512), (16, 512, 512), (32, 512, 512), (64, 512, 256), (128, 256, 128), (256, 128, 64), (512, 64, 32)):
    x = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda")
    s = torch.rand(B, Cin, device="cuda") + 0.5; d = torch.rand(B, Cout, device="cuda") + 0.5
    noise = torch.randn(1, 2 * H, 2 * H, device="cuda"); ns = torch.tensor([0.1], device="cuda"); bias = torch.randn(Cout, device="cuda")
    kw = dict(in_scale=s, out_scale=d, bias=bias, noise=noise, noise_w=ns, act=ops.ACT_LRELU, gain=1.414)
    wf = ops.pack_conv_weight(w, ops.PACK_UPFOLD, dt, 0.01); wu = ops.pack_upconv_weight(w, dt, 0.01)
    t_old = timeit(lambda: ops.conv2d(x, wf, Cout, 3, up=True, **kw))
    t_new = timeit(lambda: ops.upconv_fir(x, wu, Cout, **kw))
    print(f"{H:4d} {Cin:4d} {Cout:4d} {t_old:10.1f} {t_new:10.1f} {t_old / t_new:8.2f} {2 * 9 * Cin * Cout * H * H * B / t_new / 1e6:10.1f}")
