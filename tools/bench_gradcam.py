"""Attention-map path timing: Grad-CAM++ masks + guided back-propagation + mask2cam of one image
batch on the full VGG16 widths (stand-in weights), and the whole E_mis_align_cropping_s1 iteration - dev/bench tool.
    python tools/bench_gradcam.py [--img-size 256] [--batch 5] [--iters 10] [--step]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dge_amd
from dge_amd import grad_cam as G
ap = argparse.ArgumentParser()
ap.add_argument("--img-size", type=int, default=256); ap.add_argument("--batch", type=int, default=5)
ap.add_argument("--iters", type=int, default=10); ap.add_argument("--dtype", default="bf16")
ap.add_argument("--step", action="store_true", help="time the whole mis-align iteration (StyleGAN2 + E.BE + LPIPS + attention maps)")
ap.add_argument("--start-features", type=int, default=64)
a = ap.parse_args()
net = G.VGG16(compute_dtype=a.dtype).cuda()
if a.step:
    from dge_amd.e_align import build_models
    from dge_amd.mis_align import MisAlignStep
    Gen, E, LP = build_models(a.img_size, a.start_features, a.dtype)
    st = MisAlignStep(Gen, E, LP, net, batch_size=a.batch)
    it = [0]
    def run():
        st.step(it[0]); it[0] += 1
    what = f"E_mis_align_cropping_s1 iteration, StyleGAN2-{a.img_size} + E.BE(startf={a.start_features}) + VGG16 attention maps"
else:
    gcpp = G.GradCamPlusPlus(net, net.final_layer)
    gbp = G.GuidedBackPropagation(net)
    imgs = torch.randn(a.batch, 3, a.img_size, a.img_size, device="cuda").clamp(-1, 1)
    def run():
        mask, _ = gcpp.with_input_gradient(imgs)
        G.mask2cam(mask, imgs)
    what = f"Grad-CAM++ mask + guided back-propagation + mask2cam, VGG16 @ {a.img_size}^2"
for i in range(3):
    run()
torch.cuda.synchronize()
t0 = time.time()
for i in range(a.iters):
    run()
torch.cuda.synchronize()
dt = (time.time() - t0) / a.iters
print(f"{what}, batch {a.batch}, {a.dtype}: {dt*1e3:.2f} ms/iteration, {a.batch/dt:.1f} img/s")
