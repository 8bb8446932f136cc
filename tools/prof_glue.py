"""Which call sites issue torch glue launches (clone / contiguous-copy / zeros / fill) inside one E_align step - dev tool."""
import sys, os, collections, traceback
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dge_amd
from dge_amd.e_align import EAlignStep, build_models
G, E, LP = build_models(1024, 16, "bf16", "cuda")
G.train()
st = EAlignStep(G, E, LP, batch_size=8)
for i in range(3): st.step(i)
torch.cuda.synchronize()
cnt = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack(limit=8)[:-2]):
        if "deep-gan-encoders_amd" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"
def wrap(obj, name, cond=lambda self, *a, **k: True):
    orig = getattr(obj, name)
    def f(*a, **k):
        if cond(*a, **k): cnt[(name, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)
wrap(torch.Tensor, "clone")
wrap(torch.Tensor, "contiguous", lambda self, *a, **k: not self.is_contiguous())
wrap(torch.Tensor, "zero_")
wrap(torch.Tensor, "fill_")
wrap(torch.Tensor, "copy_")
wrap(torch.Tensor, "float", lambda self, *a, **k: self.dtype != torch.float32)
wrap(torch.Tensor, "to", lambda self, *a, **k: True)
wrap(torch.Tensor, "__mul__"); wrap(torch.Tensor, "__add__"); wrap(torch.Tensor, "__setitem__")
wrap(torch, "zeros"); wrap(torch, "full"); wrap(torch, "stack"); wrap(torch, "cat"); wrap(torch, "zeros_like"); wrap(torch, "ones")
wrap(torch, "ones_like"); wrap(torch, "full_like"); wrap(torch, "randn"); wrap(torch, "tensor"); wrap(torch, "as_tensor"); wrap(torch, "where")
wrap(torch.Tensor, "new_zeros"); wrap(torch.Tensor, "new_ones"); wrap(torch.Tensor, "new_full"); wrap(torch.Tensor, "__rmul__"); wrap(torch.Tensor, "__truediv__")
wrap(torch.Tensor, "__sub__"); wrap(torch.Tensor, "__neg__"); wrap(torch.Tensor, "sum"); wrap(torch.Tensor, "mean"); wrap(torch.Tensor, "mul_"); wrap(torch.Tensor, "add_")
wrap(torch.Tensor, "expand"); wrap(torch.Tensor, "repeat"); wrap(torch.Tensor, "cuda")
st.step(7)
torch.cuda.synchronize()
for (name, s), n in cnt.most_common(50):
    print(f"{n:4d} {name:14s} {s}")
