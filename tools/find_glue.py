"""Which Python lines of a step launch torch's own fill / copy kernels (dev tool): counts per caller of the allocation-time fills and
device copies in one eager step of the headline configuration."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dge_amd
from dge_amd.e_align import EAlignStep, build_models
dev = torch.device("cuda")
G, E, LP = build_models(1024, 16, "bf16", dev, seed=0)
G.train()
st = EAlignStep(G, E, LP, batch_size=8)
for i in range(3):
    st.step(i, prefetch_next=True)
torch.cuda.synchronize()
cnt = collections.Counter()
def caller():
    for fr in traceback.extract_stack()[:-2][::-1]:
        if "deep-gan-encoders_amd" in fr.filename or "dge_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:90] if fr.line else ''}"
    return "?"
def wrap(mod, name, tag):
    f = getattr(mod, name)
    def g(*a, **k):
        r = f(*a, **k)
        t = a[0] if (a and torch.is_tensor(a[0])) else r
        if torch.is_tensor(t) and t.is_cuda:
            cnt[(tag, caller())] += 1
        return r
    setattr(mod, name, g)
for n in ("zeros", "zeros_like", "ones", "full", "ones_like", "full_like"):
    wrap(torch, n, "fill:" + n)
for n in ("zero_", "fill_", "copy_", "clone", "mul", "__mul__", "__rmul__", "add", "__add__", "sub", "__sub__", "div", "__truediv__", "mean", "sum", "float", "contiguous", "to"):
    try:
        wrap(torch.Tensor, n, "T." + n)
    except Exception as e:
        print("skip", n, e)
st.step(3, prefetch_next=True)
torch.cuda.synchronize()
for (tag, c), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{n:4d} {tag:14s} {c}")
