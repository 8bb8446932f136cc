"""Host enqueue time vs GPU time of one E_align step (dev tool): python tools/cpu_bound.py [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dge_amd
from dge_amd.e_align import EAlignStep, build_models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G, E, LP = build_models(1024, 16, "bf16", "cuda")
G.train()
st = EAlignStep(G, E, LP, batch_size=B)
for i in range(3): st.step(i)
torch.cuda.synchronize()
enq, tot = [], []
for i in range(5):
    t0 = time.perf_counter()
    st.step(10 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(f"B={B}: host enqueue {sum(enq)/len(enq):.1f} ms/step, enqueue+drain {sum(tot)/len(tot):.1f} ms/step")
