"""data gradient + instance-norm backward: separate passes against the fused epilogue (dev tool).  python tools/perf_in_bwd.py"""
import os, sys, statistics, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops


def timed(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, H, c2, cc in [(8, 1024, 32, 16), (8, 512, 64, 32)]:
    gen = torch.Generator(device="cuda").manual_seed(1)
    g = torch.randn(B, H, H, c2, device="cuda", generator=gen).bfloat16()
    x1 = (1.5 * torch.randn(B, H, H, cc, device="cuda", generator=gen) + 0.3).bfloat16()
    w = torch.randn(c2, cc, 3, 3, device="cuda", generator=gen) / math.sqrt(9 * cc)
    sc = 0.5 + torch.rand(B, cc, device="cuda", generator=gen); sh = 0.3 * torch.randn(B, cc, device="cuda", generator=gen)
    musig = torch.cat([0.3 * torch.randn(B, cc, device="cuda", generator=gen), 0.5 + torch.rand(B, cc, device="cuda", generator=gen)], 1)
    gms = torch.randn(B, 2 * cc, device="cuda", generator=gen); noise = torch.randn(B, H, H, device="cuda", generator=gen)
    wp = ops.pack_conv_weight(w, ops.PACK_DGRAD, ops.BF16, 1.0)
    N = H * H
    dw = ops.zeros((c2, cc, 3, 3), "cuda")

    def wg_old():
        ops.conv_wgrad(g, x1, dw, sc, sh)

    def wg_new():
        d = ops.SlotStats(B, cc, "cuda"); ops.conv_wgrad_dots(g, x1, dw, sc, sh, w, d); return d

    def dg_old():
        d = ops.SlotStats(B, cc, "cuda")
        return ops.conv2d(g, wp, cc, 3, stats=d, dot_src=x1), d

    gy, d0 = dg_old()

    def inb():
        red0 = ops.zeros((2, cc), "cuda")
        return ops.in_bwd(gy, x1, (d0, gms, musig, sc, sh, N), noise=noise, act=True, red=red0, planar=True)

    dn = wg_new()

    def dg_new():
        coef = ops.in_bwd_coef(dn, gms, musig, sc, sh, N)
        red = ops.SlotStats(B, cc, "cuda")
        return ops.conv2d(g, wp, cc, 3, dot_src=x1, in_bwd=dict(coef=coef, noise=noise, red=red))

    fs = dict(wg_old=wg_old, wg_new=wg_new, dg_old=dg_old, in_bwd=inb, dg_new=dg_new)
    for f in fs.values():
        for _ in range(3):
            f()
    r = {k: [] for k in fs}
    for _ in range(5):
        for k, f in fs.items():
            r[k].append(timed(f))
    m = {k: statistics.median(v) for k, v in r.items()}
    print(f"B={B} conv_2 {cc}->{c2} @{H}^2: wgrad {m['wg_old']:.1f} -> {m['wg_new']:.1f} us (+dots) | dgrad {m['dg_old']:.1f} + in_bwd {m['in_bwd']:.1f} = "
          f"{m['dg_old'] + m['in_bwd']:.1f} -> fused {m['dg_new']:.1f} us", flush=True)


# the last data gradient: conv_1 of block 0 + in_bwd_fromrgb against the fused reduction
B, H, cc = 8, 1024, 16
gen = torch.Generator(device="cuda").manual_seed(2)
g = torch.randn(B, H, H, cc, device="cuda", generator=gen).bfloat16()
img = torch.rand(B, 3, H, H, device="cuda", generator=gen) * 2 - 1
x0, img4 = ops.fromrgb(img, torch.randn(cc, 3, device="cuda", generator=gen), torch.zeros(cc, device="cuda"), ops.BF16, img4=True)
w = torch.randn(cc, cc, 3, 3, device="cuda", generator=gen) / 12
sc = 0.5 + torch.rand(B, cc, device="cuda", generator=gen); sh = 0.3 * torch.randn(B, cc, device="cuda", generator=gen)
musig = torch.cat([0.3 * torch.randn(B, cc, device="cuda", generator=gen), 0.5 + torch.rand(B, cc, device="cuda", generator=gen)], 1)
gms = torch.randn(B, 2 * cc, device="cuda", generator=gen)
extra = torch.randn(B, H // 2, H // 2, cc, device="cuda", generator=gen).bfloat16()
wp = ops.pack_conv_weight(w, ops.PACK_DGRAD, ops.BF16, 1.0)
N = H * H
d0 = ops.SlotStats(B, cc, "cuda")
gy = ops.conv2d(g, wp, cc, 3, stats=d0, dot_src=x0)
dw = ops.zeros((cc, cc, 3, 3), "cuda")
dn = ops.SlotStats(B, cc, "cuda"); ops.conv_wgrad_dots(g, x0, dw, sc, sh, w, dn)
tiny = x0.new_empty((1, 1, 1, 1))
fs = dict(
    fromrgb=lambda: ops.fromrgb(img, torch.zeros(cc, 3, device="cuda"), torch.zeros(cc, device="cuda"), ops.BF16),
    fromrgb4=lambda: ops.fromrgb(img, torch.zeros(cc, 3, device="cuda"), torch.zeros(cc, device="cuda"), ops.BF16, img4=True),
    dg_old=lambda: ops.conv2d(g, wp, cc, 3, stats=ops.SlotStats(B, cc, "cuda"), dot_src=x0),
    in_bwd_fr=lambda: ops.in_bwd_fromrgb(gy, x0, (d0, gms, musig, sc, sh, N), img, extra=extra, extra_pool=True, extra_scale=0.25),
    fused=lambda: ops.conv2d(g, wp, cc, 3, dot_src=x0, out=tiny, in_bwd=dict(coef=ops.in_bwd_coef(dn, gms, musig, sc, sh, N), fr=ops.SlotStats(B, cc, "cuda"), img4=img4, extra=extra, extra_scale=0.25)))
for f in fs.values():
    for _ in range(3):
        f()
r = {k: [] for k in fs}
for _ in range(5):
    for k, f in fs.items():
        r[k].append(timed(f))
m = {k: statistics.median(v) for k, v in r.items()}
print(f"B={B} conv_1 16->16 @{H}^2 (block 0): fromrgb {m['fromrgb']:.1f} -> {m['fromrgb4']:.1f} us (+img4) | dgrad {m['dg_old']:.1f} + in_bwd_fromrgb {m['in_bwd_fr']:.1f} = "
      f"{m['dg_old'] + m['in_bwd_fr']:.1f} -> fused {m['fused']:.1f} us", flush=True)


# block-input form: conv_1 of blocks 1 / 2
for B, H, cc in [(8, 512, 32), (8, 256, 64)]:
    gen = torch.Generator(device="cuda").manual_seed(3)
    g = torch.randn(B, H, H, cc, device="cuda", generator=gen).bfloat16()
    x = (1.5 * torch.randn(B, H, H, cc, device="cuda", generator=gen) + 0.3).bfloat16()
    w = torch.randn(cc, cc, 3, 3, device="cuda", generator=gen) / math.sqrt(9 * cc)
    sc = 0.5 + torch.rand(B, cc, device="cuda", generator=gen); sh = 0.3 * torch.randn(B, cc, device="cuda", generator=gen)
    musig = torch.cat([0.3 * torch.randn(B, cc, device="cuda", generator=gen), 0.5 + torch.rand(B, cc, device="cuda", generator=gen)], 1)
    gms = torch.randn(B, 2 * cc, device="cuda", generator=gen)
    extra = torch.randn(B, H // 2, H // 2, cc, device="cuda", generator=gen).bfloat16()
    wp = ops.pack_conv_weight(w, ops.PACK_DGRAD, ops.BF16, 1.0)
    N = H * H
    d0 = ops.SlotStats(B, cc, "cuda")
    gy = ops.conv2d(g, wp, cc, 3, stats=d0, dot_src=x)
    dw = ops.zeros((cc, cc, 3, 3), "cuda")
    dn = ops.SlotStats(B, cc, "cuda"); ops.conv_wgrad_dots(g, x, dw, sc, sh, w, dn)
    fs = dict(
        dg_old=lambda: ops.conv2d(g, wp, cc, 3, stats=ops.SlotStats(B, cc, "cuda"), dot_src=x),
        in_bwd=lambda: ops.in_bwd(gy, x, (d0, gms, musig, sc, sh, N), extra=extra, extra_pool=True, extra_scale=0.25),
        fused=lambda: ops.conv2d(g, wp, cc, 3, dot_src=x, in_bwd=dict(coef=ops.in_bwd_coef(dn, gms, musig, sc, sh, N), extra=extra, extra_scale=0.25)))
    for f in fs.values():
        for _ in range(3):
            f()
    r = {k: [] for k in fs}
    for _ in range(5):
        for k, f in fs.items():
            r[k].append(timed(f))
    m = {k: statistics.median(v) for k, v in r.items()}
    print(f"B={B} conv_1 {cc}->{cc} @{H}^2 (block input): dgrad {m['dg_old']:.1f} + in_bwd {m['in_bwd']:.1f} = {m['dg_old'] + m['in_bwd']:.1f} -> fused {m['fused']:.1f} us", flush=True)
