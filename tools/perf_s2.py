"""Quick per-layer timing of the StyleGAN2-1024 synthesis forward (dev tool, GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dge_amd
from dge_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cd = sys.argv[2] if len(sys.argv) > 2 else "bf16"
torch.manual_seed(0)
G = dge_amd.StyleGAN2Generator(1024, compute_dtype=cd).cuda().eval()
wp = torch.randn(B, 18, 512, device="cuda")
with torch.no_grad():
    for _ in range(int(os.environ.get("WARM", "3"))):
        G.synthesis(wp)
    torch.cuda.synchronize()
    t0 = time.time()
    N = int(os.environ.get("NIT", "10"))
    for _ in range(N):
        G.synthesis(wp)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / N
print(f"synthesis B={B} {cd}: {dt*1e3:.3f} ms/step, {dt*1e3/B:.3f} ms/img, {150.76e9*B/dt/1e12:.1f} TFLOP/s algorithmic")

# per-layer conv timing
mmac = {}
syn = G.synthesis
x = ops.nchw_to_nhwc(syn.early_layer.const.detach(), B, ops.BF16 if cd == "bf16" else ops.F32)
with torch.no_grad():
    for i in range(syn.num_layers - 1):
        L = getattr(syn, f"layer{i}")
        L(x, wp[:, i]); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        packed, wsq = L._prepared(ops.dtype_of(x))
        s, d = L.styles(wp[:, i])
        nz = L.noise.reshape(1, L.res, L.res); nw = L.noise_strength.detach().reshape(1)
        e0.record()
        for _ in range(5):
            y = ops.conv2d(x, packed, L.out_c, 3, up=L.up, in_scale=s, out_scale=d, bias=L.bias, noise=nz, noise_w=nw, act=1, gain=1.414)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5
        hin = x.shape[1]
        fl = 2 * 9 * L.in_c * L.out_c * hin * hin * B
        byts = (x.numel() + y.numel()) * x.element_size()
        print(f"layer{i:2d} {L.in_c:3d}->{L.out_c:3d} out{L.res:4d} up={int(L.up)}: {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TF/s alg  {byts/t/1e6:7.1f} GB/s")
        x = y
        if i % 2 == 0:
            O_ = getattr(syn, f"output{i//2}")
            e0.record()
            for _ in range(5):
                O_(x, wp[:, i + 1])
            e1.record(); torch.cuda.synchronize()
            print(f"   torgb{i//2}: {e0.elapsed_time(e1)/5*1e3:8.1f} us")
