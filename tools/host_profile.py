"""cProfile of the host side of one E_align step (where the Python time of the ~850 launches goes) - dev tool."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dge_amd
from dge_amd.e_align import EAlignStep, build_models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
G, E, LP = build_models(1024, 16, "bf16", "cuda")
G.train()
st = EAlignStep(G, E, LP, batch_size=B)
for i in range(3): st.step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(3): st.step(10 + i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:50]))
