import sys, os, warnings
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import dge_amd
from dge_amd.e_align import EAlignStep, build_models
G, E, LP = build_models(1024, 16, "bf16", "cuda")
G.train()
st = EAlignStep(G, E, LP, batch_size=8)
for i in range(3): st.step(i)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
warnings.simplefilter("always")
import traceback
orig = warnings.showwarning
def show(message, category, filename, lineno, file=None, line=None):
    if "synchron" in str(message).lower():
        frames = [f for f in traceback.extract_stack() if "deep-gan-encoders_amd" in f.filename]
        print("SYNC at", [(os.path.basename(f.filename), f.lineno) for f in frames[-3:]], str(message)[:60])
warnings.showwarning = show
st.step(7)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
