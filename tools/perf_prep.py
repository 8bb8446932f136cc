"""Fused tail backward (dge_conv_desc.prep) vs the separate pass, per layer of the synthesis backward (dev tool, GPU box)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dge_amd
from dge_amd import ops
from dge_amd._lib import last_kernel

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = torch.Generator(device="cuda").manual_seed(0)
gain = math.sqrt(2.0)
print(f"{'layer i -> i-1':>28} {'conv':>8} {'+prep pass':>10} {'sum':>8} {'fused':>8}  kernel")
# (cout_fwd, cin_fwd, R of layer i-1, up)
for cof, cif, R, up in [(32, 32, 1024, False), (32, 64, 512, True), (64, 64, 512, False), (64, 128, 256, True), (128, 128, 256, False),
                        (128, 256, 128, True), (256, 256, 128, False), (256, 512, 64, True), (512, 512, 64, False), (512, 512, 32, True),
                        (512, 512, 32, False), (512, 512, 16, False), (512, 512, 8, False)]:
    Rg = 2 * R if up else R
    gz = torch.randn(B, Rg, Rg, cof, device="cuda", generator=g).to(torch.bfloat16)
    x = torch.randn(B, R, R, cif, device="cuda", generator=g).to(torch.bfloat16)
    d_in = 0.5 + torch.rand(B, cof, device="cuda", generator=g)
    d_prev = 0.5 + torch.rand(B, cif, device="cuda", generator=g)
    s = 1 + 0.3 * torch.randn(B, cif, device="cuda", generator=g)
    noise = torch.randn(1, R, R, device="cuda", generator=g); ns = torch.tensor([0.3], device="cuda")
    w = torch.randn(cof, cif, 3, 3, device="cuda", generator=g)
    mode = ops.PACK_UPFOLD_DGRAD if up else ops.PACK_DGRAD
    pk = ops.pack_conv_weight(w, ops.pack_mode_for(w, mode, R, R, ops.BF16), ops.BF16, 1 / math.sqrt(9 * cif))
    st = torch.zeros(64, B, cif, 2, device="cuda")
    Rr = torch.zeros(B, cif, 3, device="cuda")
    holder = {}
    def plain():
        holder["g"] = ops.conv2d(gz, pk, cif, 3, in_s2d=up, out_scale=s, stats=st[0], dot_src=x)
    def prep_pass():
        ops.modconv_bwd_prep(holder["g"], x, d_prev, noise, gain, Rr)
    def fused():
        P = ops.SlotStats(B, cif, "cuda")
        ops.conv2d(gz, pk, cif, 3, in_s2d=up, in_scale=d_in, out_scale=s, stats=st[0], dot_src=x, prep=dict(gain=gain, noise=noise, ns=ns, stats=P))
    t0 = timeit(plain); t1 = timeit(prep_pass); t2 = timeit(fused)
    print(f"{str((cof, cif, R, 'up' if up else '')):>28} {t0:8.1f} {t1:10.1f} {t0 + t1:8.1f} {t2:8.1f}  {last_kernel()}")
