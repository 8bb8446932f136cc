import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dge_amd import ops
from dge_amd._lib import last_kernel
def timeit(run, n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, ci, co, H, W) in [(16, 512, 512, 48, 48), (16, 512, 512, 64, 48), (16, 256, 512, 48, 48)]:
    x = torch.randn(B, H, W, ci, device="cuda").bfloat16()
    w = torch.randn(co, ci, 3, 3, device="cuda") / (ci * 9) ** 0.5
    wp = ops.pack_conv_weight(w, ops.PACK_FWD, ops.BF16)
    bias = torch.randn(co, device="cuda")
    t = timeit(lambda: ops.conv2d(x, wp, co, 3, bias=bias, act=ops.ACT_RELU))
    print(f"BN={os.environ.get('DGE_CONV_BN','-')} fwd B={B} {ci}->{co} @{H}x{W}: {t:7.1f} us {2*9*ci*co*H*W*B/t/1e6:7.1f} TF/s {last_kernel()}")
