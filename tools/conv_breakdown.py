"""Per-conv-launch time breakdown of one E_align step (dev tool): python tools/conv_breakdown.py [batch]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dge_amd
from dge_amd import ops
from dge_amd.e_align import EAlignStep, build_models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G, E, LP = build_models(1024, 16, "bf16", "cuda")
G.train()
st = EAlignStep(G, E, LP, batch_size=B)
for i in range(2): st.step(i)
torch.cuda.synchronize()
ops.PROFILE = []
st.step(5)
torch.cuda.synchronize()
rows = collections.OrderedDict()
for e0, e1, fl, tag, _ in ops.PROFILE:
    r = rows.setdefault(tag, [0, 0.0, 0.0]); r[0] += 1; r[1] += e0.elapsed_time(e1) * 1e3; r[2] += fl
tot = sum(r[1] for r in rows.values())
print(f"conv launches {len(ops.PROFILE)}, total {tot/1e3:.2f} ms")
print("  n   total_us   avg_us  TF/s   (B,H,W,Cin,Cout,k,up,s2d)")
for tag, r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{r[0]:3d} {r[1]:10.1f} {r[1]/r[0]:8.1f} {r[2]/r[1]/1e6:6.1f}   {tag}")
