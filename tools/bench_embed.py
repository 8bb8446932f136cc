"""Iterations/s of the inversion loop (BASELINE config 5: StyleGAN1 FFHQ-1024 + E_Blur, batch 1) - dev/bench tool.
    python tools/bench_embed.py [--img-size 1024] [--start-features 16] [--iters 20]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dge_amd
from dge_amd.embedding import EmbedStep, build_models
ap = argparse.ArgumentParser()
ap.add_argument("--img-size", type=int, default=1024); ap.add_argument("--start-features", type=int, default=16)
ap.add_argument("--iters", type=int, default=20); ap.add_argument("--batch", type=int, default=1); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--graph", action="store_true")
a = ap.parse_args()
Gs, E, LP = build_models(a.img_size, a.start_features, a.dtype)
st = EmbedStep(Gs, E, LP)
st.begin_image()
with torch.no_grad():
    imgs1 = Gs.forward(torch.randn(a.batch, 2 * Gs.layer_count, 512, device="cuda"), Gs.layer_count - 1).detach()
if a.graph:
    st.capture(imgs1)
    run = st.replay
else:
    run = lambda: st.step(imgs1)
for i in range(3):
    run()
torch.cuda.synchronize()
t0 = time.time()
for i in range(a.iters):
    run()
torch.cuda.synchronize()
dt = (time.time() - t0) / a.iters
mode = ", hipGraph replay" if a.graph else ""
print(f"embedding_img loop, StyleGAN1-{a.img_size} + E_Blur(startf={a.start_features}), batch {a.batch}, {a.dtype}{mode}: "
      f"{dt*1e3:.1f} ms/iteration, {1/dt:.1f} it/s, 1500 iterations in {1500*dt:.0f} s")
