import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, dge_amd
from dge_amd.e_align import EAlignStep, build_models
dev = torch.device("cuda", 0)
G, E, LP = build_models(1024, 16, "bf16", dev, seed=0); G.train()
st = EAlignStep(G, E, LP, batch_size=8)
t0 = time.time()
for i in range(400):
    r = st.step(i)
    if i in (10, 100, 200, 399):
        torch.cuda.synchronize()
        print(i, f"reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB  allocated {torch.cuda.memory_allocated()/2**30:.2f} GiB  loss {float(r['loss_tsa']):.4f}  t {time.time()-t0:.1f}s", flush=True)
