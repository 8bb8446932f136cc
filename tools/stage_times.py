"""HIP-event time of every stage of one E_align_s2 step at the benchmark configuration (config 3, batch 8, bf16) - dev tool.
Stages are bracketed by events on the launch stream (the step is GPU-bound, so event time = kernel time of the stage)."""
import os, sys, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dge_amd
from dge_amd import e_align, losses, autograd_s2, autograd_enc, autograd_enc_bwd
from dge_amd.e_align import EAlignStep, build_models

B = int(os.environ.get("B", "8"))
G, E, LP = build_models(1024, 16, "bf16", "cuda")
G.train()
st = EAlignStep(G, E, LP, batch_size=B)
for i in range(3): st.step(i)
torch.cuda.synchronize()
marks = []


def wrap(obj, name, tag):
    orig = getattr(obj, name)

    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(*a, **k)
        e1.record()
        marks.append((tag, e0, e1))
        return r
    setattr(obj, name, f)


wrap(st.gen, "sample", "G sample (mapping + synthesis, no grad)")
wrap(st.gen, "synth", "G synthesis (saved for backward)")
wrap(autograd_enc, "encoder_forward", "E forward")
wrap(e_align.losses, "image_loss_tsa", "image losses (value + gradient)")
wrap(e_align.losses, "space_loss", "latent loss")
wrap(autograd_s2, "synthesis_backward", "G synthesis backward")
wrap(autograd_enc_bwd, "encoder_backward", "E backward")
wrap(st.opt, "step", "LREQAdam step")
tot = collections.OrderedDict()
N = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(N): st.step(10 + i)
e1.record()
torch.cuda.synchronize()
for tag, a, b in marks:
    tot.setdefault(tag, []).append(a.elapsed_time(b))
step_ms = e0.elapsed_time(e1) / N
print(f"step {step_ms:.2f} ms (batch {B})")
acc = 0.0
for tag, v in tot.items():
    per_step = sum(v) / N
    acc += per_step
    print(f"  {per_step:7.3f} ms  {len(v) // N} x {sum(v) / len(v):7.3f}  {tag}")
print(f"  {step_ms - acc:7.3f} ms  outside the brackets")
