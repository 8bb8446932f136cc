"""Full-size determinism check (GPU box): the headline E_align_s2 step (StyleGAN2-1024, E.BE(16), LPIPS, batch 8, bf16) run
twice for two iterations from the same state, in the default mode and with ops.set_deterministic(True); prints whether the
encoder parameters are bit-identical and the step time of both modes.   python tools/check_determinism.py [batch]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dge_amd import ops
from dge_amd.e_align import EAlignStep, build_models

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def run(nit=2):
    G, E, LP = build_models(1024, 16, "bf16", seed=0)
    G.train()
    st = EAlignStep(G, E, LP, batch_size=B)
    for it in range(nit):
        r = st.step(it)
    torch.cuda.synchronize()
    t0 = time.time()
    for it in range(nit, nit + 3):
        r = st.step(it)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    h = hashlib.sha256()
    for k, v in sorted(E.state_dict().items()):
        h.update(v.detach().float().cpu().numpy().tobytes())
    return h.hexdigest()[:16], float(r["loss_tsa"]), float(r["loss_w"]), dt


for mode in (False, True):
    ops.set_deterministic(mode)
    a = run(); b = run()
    print(f"deterministic={mode}: run1 {a[0]} loss {a[1]:.6f} {a[2]:.6f} | run2 {b[0]} loss {b[1]:.6f} {b[2]:.6f} | "
          f"bit-identical parameters: {a[0] == b[0]} | {a[3]*1e3:.1f} ms/step")
ops.set_deterministic(False)
