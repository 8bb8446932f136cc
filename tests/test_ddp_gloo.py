"""world_size-2 (gloo, CPU) test of the data-parallel host logic: the globally-reduced loss sums
(exact cosine / means) and the flat-bucket gradient all-reduce used by EAlignStep."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dge_amd  # noqa
    from dge_amd.losses import GlobalBatch
    from dge_amd.e_align import EAlignStep
    from oracle import ref_torch as O
    torch.manual_seed(0)
    # --- global batch of 4 latents, each rank owns 2 rows: exact cosine needs the global sums
    a = torch.randn(4, 6, 16); b = a * 0.8 + 0.3 * torch.randn(4, 6, 16)
    la, lb = a[rank * 2:(rank + 1) * 2], b[rank * 2:(rank + 1) * 2]
    sums = torch.stack([((la - lb) ** 2).sum(), (la * lb).sum(), (la * la).sum(), (lb * lb).sum()])
    GlobalBatch(world).reduce(sums)
    n = a.numel()
    loss_global = 5 * sums[0] / n + 3 * (1 - sums[1] / (sums[2].sqrt() * sums[3].sqrt()))
    ref, _ = O.space_loss(a, b, image_space=False)
    ok1 = abs(float(loss_global) - float(ref)) < 1e-5 * abs(float(ref))

    # --- flat-bucket gradient exchange of EAlignStep (sum over ranks, views installed as .grad)
    lin = torch.nn.Linear(5, 3)
    with torch.no_grad():
        for p in lin.parameters():
            p.fill_(0.5)
    st = EAlignStep.__new__(EAlignStep)
    st.E, st.world, st.rank, st.exact_ddp, st.dev, st._flat, st.dist_on = lin, world, rank, True, torch.device("cpu"), None, True
    for p in lin.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    gs = st._sync_grads()
    ok2 = gs is None and all(torch.allclose(p.grad, torch.full_like(p, 3.0)) for p in lin.parameters())
    st.exact_ddp = False
    for p in lin.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    gs = st._sync_grads()
    ok3 = abs(float(gs) - 0.5) < 1e-7
    # --- early bucket: part of the gradients is reduced asynchronously from inside the backward, the rest in _sync_grads
    st2 = EAlignStep.__new__(EAlignStep)
    net = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    st2.E, st2.world, st2.rank, st2.exact_ddp, st2.dev, st2._flat, st2.dist_on = net, world, rank, True, torch.device("cpu"), None, True
    ok5 = True
    for phase in range(2):                                   # two phases per step re-use the layout
        for i_, p in enumerate(net.parameters()):
            p.grad = torch.full_like(p, float((rank + 1) * (i_ + 1) + phase))
        early = {n: p.grad.clone() for n, p in net.named_parameters() if n.startswith("1.")}
        st2.early_reduce(early)
        st2._sync_grads()
        for i_, p in enumerate(net.parameters()):
            want = float(sum((r + 1) * (i_ + 1) + phase for r in range(world)))
            ok5 = ok5 and torch.allclose(p.grad, torch.full_like(p, want)) and p.grad.data_ptr() >= st2._flat.data_ptr()
    # every rank must draw the same global z and take its own slice
    from dge_amd.e_align import set_seed
    set_seed(7)
    zg = torch.randn(2 * world, 8)
    t = zg.clone(); dist.broadcast(t, 0)
    ok4 = torch.equal(t, zg)
    q.put((rank, ok1, ok2, ok3, ok4, ok5))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(all(r[1:]) for r in res), res
