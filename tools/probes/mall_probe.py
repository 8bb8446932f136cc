"""Does a producer -> consumer pair of streaming kernels run faster when the intermediate is small enough to sit in the 256 MB
memory-side cache?  y = 2x, z = y + 1 on bf16 tensors of 537 MB (one 1024^2 x 32-channel activation at batch 8): whole tensors
against interleaved chunks of 1/2 .. 1/16 (dev probe, GPU box)."""
import torch
N = 8 * 1024 * 1024 * 32
x = torch.randn(N, device="cuda").bfloat16()
y = torch.empty_like(x); z = torch.empty_like(x)
def run(chunks):
    n = N // chunks
    for k in range(chunks):
        torch.mul(x[k * n:(k + 1) * n], 2, out=y[k * n:(k + 1) * n])
        torch.add(y[k * n:(k + 1) * n], 1, out=z[k * n:(k + 1) * n])
for chunks in (1, 2, 4, 8, 16, 1):
    for _ in range(3): run(chunks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run(chunks)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"chunks {chunks:2d} ({N * 2 / chunks / 1e6:6.1f} MB each): {t * 1e3:7.1f} us per pair  ({4 * N * 2 / t / 1e9:.2f} TB/s of tensor traffic)")
