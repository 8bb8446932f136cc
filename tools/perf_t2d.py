"""Data gradient of the StyleGAN2 up layers at config 3 (batch 8): folded space-to-depth form against the phase form
(dge_fir_t2d + in_t2d), both with the fused tail backward the synthesis backward uses - dev tool."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dge_amd
from dge_amd import ops
from dge_amd._lib import last_kernel
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 8
g = torch.Generator(device="cuda").manual_seed(0)
print(f"{'layer (cout_up <- cin_up @ R)':32s} {'folded':>8s} {'fir':>7s} {'conv':>7s} {'t2d sum':>8s}  kernel")
for cof, cif, R in [(32, 64, 512), (64, 128, 256), (128, 256, 128), (256, 512, 64), (512, 512, 32)]:
    gz = torch.randn(B, 2 * R, 2 * R, cof, device="cuda", generator=g).to(torch.bfloat16)
    x = torch.randn(B, R, R, cif, device="cuda", generator=g).to(torch.bfloat16)
    add = torch.randn(B, R, R, cif, device="cuda", generator=g).to(torch.bfloat16)
    s = 1 + 0.3 * torch.randn(B, cif, device="cuda", generator=g)
    d = 0.5 + torch.rand(B, cof, device="cuda", generator=g)
    w = torch.randn(cof, cif, 3, 3, device="cuda", generator=g)
    ws = 1 / math.sqrt(9 * cif)
    noise = torch.randn(1, R, R, device="cuda", generator=g)
    ns = torch.tensor([0.3], device="cuda")
    pk3 = ops.pack_conv_weight(w, ops.PACK_UPFOLD_DGRAD, ops.BF16, ws)
    pk6 = ops.pack_conv_weight(w, ops.PACK_UPT2D_DGRAD, ops.BF16, ws)
    def kw():
        return dict(out_scale=s, addend=add, stats=ops.SlotStats(B, cif, "cuda"), dot_src=x,
                    prep=dict(gain=math.sqrt(2.0), noise=noise, ns=ns, stats=ops.SlotStats(B, cif, "cuda")))
    t_f = timeit(lambda: ops.conv2d(gz, pk3, cif, 3, in_s2d=True, in_scale=d, **kw()))
    t_fir = timeit(lambda: ops.fir_t2d(gz, d))
    z = ops.fir_t2d(gz, d)
    t_c = timeit(lambda: ops.conv2d(z, pk6, cif, 3, in_t2d=True, **kw()))
    print(f"{str((cof, cif, R)):32s} {t_f:8.1f} {t_fir:7.1f} {t_c:7.1f} {t_fir + t_c:8.1f}  {last_kernel()}")
