"""What un-batching the two LPIPS image sets would cost / save (dev probe): value_and_grad at B pairs against B/2 pairs (= the conv batch
of one image set alone) with and without the gradient, per attention-window size of the headline step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dge_amd.lpips import LPIPS
from oracle import lpips_ref as LR
LP = LPIPS(compute_dtype="bf16").cuda(); LP.load_state_dict(LR.seeded_params(0))
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 8
for (h, w) in ((256, 256), (256, 192), (176, 176)):
    a = torch.rand(B, 3, h, w, device="cuda") * 2 - 1; b = torch.rand(B, 3, h, w, device="cuda") * 2 - 1
    t_full = timeit(lambda: LP.value_and_grad(a, b, need_grad=True))
    t_half = timeit(lambda: LP.value_and_grad(a[:B // 2].contiguous(), b[:B // 2].contiguous(), need_grad=True))
    t_half_ng = timeit(lambda: LP.value_and_grad(a[:B // 2].contiguous(), b[:B // 2].contiguous(), need_grad=False))
    t_full_ng = timeit(lambda: LP.value_and_grad(a, b, need_grad=False))
    print(f"{h}x{w}: fwd(2B)+bwd(B) {t_full:7.0f} us | fwd(2B) {t_full_ng:7.0f} | fwd(B)+bwd(B/2) {t_half:7.0f} | fwd(B) {t_half_ng:7.0f}  "
          f"=> bwd(B) ~ {t_full - t_full_ng:6.0f}; split main path ~ fwd(B) + bwd(B) = {t_half_ng + t_full - t_full_ng:7.0f} (saves {t_full_ng - t_half_ng:5.0f}), side stream + {t_half_ng:6.0f}")
