"""up_pp (csrc/up_pp.hip) against upconv_fir on the generator's up layers: agreement + isolated timing (dev tool).
python tools/perf_up_pp.py [quick]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dge_amd
import dge_amd._lib
from dge_amd import ops

DEV = "cuda"
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
shapes = [(2, 64, 64, 128, 64)] if quick else [(8, 16, 16, 512, 512), (8, 32, 32, 512, 512), (8, 64, 64, 512, 256), (8, 128, 128, 256, 128), (8, 256, 256, 128, 64), (2, 70, 66, 128, 96)]
g = torch.Generator(device=DEV); g.manual_seed(1)
for (B, H, W, cin, cout) in shapes:
    x = (torch.randn(B, H, W, cin, device=DEV, generator=g) * 0.7).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=DEV, generator=g)
    wscale = 1.0 / math.sqrt(9 * cin)
    s = 1.0 + 0.3 * torch.randn(B, cin, device=DEV, generator=g)
    d = 0.5 + torch.rand(B, cout, device=DEV, generator=g)
    noise = torch.randn(1, 2 * H, 2 * W, device=DEV, generator=g)
    ns = torch.tensor([0.37], device=DEV)
    bias = 0.2 * torch.randn(cout, device=DEV, generator=g)
    wu = ops.pack_upconv_weight(w, ops.BF16, wscale)
    kw = dict(bias=bias, bias_scale=1.0, noise=noise, noise_w=ns, act=ops.ACT_LRELU, gain=math.sqrt(2.0))
    os.environ["DGE_NO_UPSTREAM"] = "1"
    y0 = ops.upconv_fir(x, wu, cout, in_scale=s, out_scale=d, **kw)
    k0 = dge_amd._lib.last_kernel()
    if not ops.up_pp_supported(B, H, W, cin, cout, ops.BF16):
        print((B, H, W, cin, cout), "not supported by up_pp"); continue
    wimg = ops.pack_up_pp(wu, cout, cin, in_scale=s, out_scale=d, gain=math.sqrt(2.0))
    y1 = ops.up_pp(x, wimg, cout, **kw)
    torch.cuda.synchronize()
    k1 = dge_amd._lib.last_kernel()
    diff = (y1.float() - y0.float()).abs()
    mx = y0.float().abs().max().item()
    print(f"{(B, H, W, cin, cout)} {k0} vs {k1}: max|diff| {diff.max().item():.4f} of max {mx:.3f}  (rel {diff.max().item() / mx:.2e}), mean {diff.mean().item():.2e}")
    if diff.max().item() > 0.05 * mx:
        bad = diff > 0.05 * mx
        rows = bad.any(dim=3).any(dim=2).any(dim=0).nonzero().flatten().tolist()
        cols = bad.any(dim=3).any(dim=1).any(dim=0).nonzero().flatten().tolist()
        chs = bad.any(dim=2).any(dim=1).any(dim=0).nonzero().flatten().tolist()
        print("  bad rows", rows[:40], "... n", len(rows)); print("  bad cols", cols[:40], "... n", len(cols)); print("  bad chans", chs[:40], "n", len(chs))
        print("  frac bad", bad.float().mean().item())
    def t(fn, n=20):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    if os.environ.get("DGE_UP_DBG"):
        t1 = t(lambda: ops.up_pp(x, wimg, cout, **kw))
        print(f"    DGE_UP_DBG={os.environ['DGE_UP_DBG']}: up_pp {t1:.1f} us")
    elif not quick:
        t0 = t(lambda: ops.upconv_fir(x, wu, cout, in_scale=s, out_scale=d, **kw))
        t1 = t(lambda: ops.up_pp(x, wimg, cout, **kw))
        tp = t(lambda: ops.pack_up_pp(wu, cout, cin, in_scale=s, out_scale=d, gain=math.sqrt(2.0)))
        fl = 2.0 * 9 * cin * cout * H * W * B
        print(f"    upconv_fir {t0:.1f} us ({fl / t0 / 1e6:.0f} TF/s)   up_pp {t1:.1f} us ({fl / t1 / 1e6:.0f} TF/s) + pack {tp:.1f} us")
