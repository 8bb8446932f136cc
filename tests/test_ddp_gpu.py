"""Two data-parallel ranks x B images == one process at batch 2B, for the REAL two-phase step (the couplings the
reference has across a batch: training_utils.py:73-75 batch-flattened cosine, stylegan2_generator.py:177-191 w_avg EMA and
style mixing, E.py:60,73 per-sample noise).  Two processes share the one GPU of the box, torch.distributed over gloo (device
tensors staged through the host by e_align._all_reduce; RCCL refuses two ranks on one device), every collective of the step is
real: w_avg mean, the packed loss sums, early + remainder gradient buckets in both phases."""
import os
import socket
import sys

import pytest
import torch

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _models():
    import dge_amd
    from dge_amd.encoder import BE
    from dge_amd.lpips import LPIPS
    from tests.golden import recipe as R
    from tests.helpers import s2_shapes, enc_shapes
    from oracle import lpips_ref as LR
    G = dge_amd.StyleGAN2Generator(64, fmaps_base=2048, fmaps_max=128, compute_dtype="f32").cuda()
    G.load_state_dict(R.fill_s2(s2_shapes(64, fmaps_base=2048, fmaps_max=128), seed=11))
    G.train()
    for p in G.parameters():
        p.requires_grad_(False)
    E = BE(startf=16, maxf=64, layer_count=5, compute_dtype="f32").cuda()
    E.load_state_dict(R.fill_encoder(enc_shapes(16, 64, 5), seed=31))
    LP = LPIPS(compute_dtype="f32").cuda()
    LP.load_state_dict(LR.seeded_params(0))
    return G, E, LP


def _run_steps(B, nsteps=2, prefetch=False):
    from dge_amd.e_align import EAlignStep
    G, E, LP = _models()
    st = EAlignStep(G, E, LP, lr=0.0015, batch_size=B)
    out = {}
    for it in range(nsteps):
        # z, style-mixing latent and encoder noise are all drawn inside.  prefetch (the ranks of the two-process run): the training
        # loop's eager form - the generator pass of iteration it + 1, with its w_avg all-reduce, goes out on a side stream beside the
        # image losses of iteration it, between the collectives of the main stream
        r = st.step(it, prefetch_next=(prefetch and it + 1 < nsteps))
        out[f"it{it}_w2"] = r["w2"].detach().cpu()
        out[f"it{it}_losses"] = torch.stack([r["loss_tsa"].detach().cpu(), r["loss_w"].detach().cpu()])
        out[f"it{it}_info"] = r["info_img"].cpu()
    out["params"] = {k: v.detach().cpu() for k, v in E.state_dict().items()}
    out["w_avg"] = G.truncation.w_avg.cpu()
    out["world"] = st.world
    return out


def _worker(rank, world, port, q, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    if backend == "nccl":                    # RCCL: one device per rank
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = _run_steps(2, prefetch=True)
        # numpy (pickled by value): torch tensors would travel as shared-memory handles that die with this process
        out["params"] = {k: v.numpy() for k, v in out["params"].items()}
        q.put((rank, {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_ranks_times_b_equal_one_process_at_2b(backend):
    """gloo: both ranks on the one GPU of the box.  nccl: the same over RCCL with one device per rank - the path bench.py --gpus N
    runs (collectives on RCCL's stream, asynchronous early bucket); needs two devices, skipped on a single-GPU box."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one device per rank: fewer than two GPUs visible")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
    T = torch.from_numpy
    for r in res.values():
        r["params"] = {k: T(v) for k, v in r["params"].items()}
        for k in list(r):
            if k not in ("params", "world"):
                r[k] = T(r[k])
    one = _run_steps(4)
    assert res[0]["world"] == 2 and one["world"] == 1
    # both ranks end with the same parameters, bit for bit (identical summed gradients, identical updates)
    for k in res[0]["params"]:
        assert torch.equal(res[0]["params"][k], res[1]["params"][k]), k
    assert torch.equal(res[0]["w_avg"], res[1]["w_avg"])
    rel = lambda a, b: ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
    for it in range(2):
        # the ranks' samples are rows [0:2] and [2:4] of the single-process batch: same z, same mixing latent, same noise
        w2 = torch.cat([res[0][f"it{it}_w2"], res[1][f"it{it}_w2"]])
        assert rel(w2, one[f"it{it}_w2"]) < (2e-5 if it == 0 else 2e-3), it
        # global losses (cosine over the batch-flattened vector, means over the global batch)
        assert rel(res[0][f"it{it}_losses"], one[f"it{it}_losses"]) < (1e-4 if it == 0 else 2e-3), it
        assert rel(res[0][f"it{it}_info"], one[f"it{it}_info"]) < (1e-3 if it == 0 else 5e-3), it
    assert rel(res[0]["w_avg"], one["w_avg"]) < 1e-5
    from tests.golden import recipe as R
    from tests.helpers import enc_shapes
    before = R.fill_encoder(enc_shapes(16, 64, 5), seed=31)
    worst = 0.0
    for k, v in one["params"].items():
        assert rel(res[0]["params"][k], v) < 2e-3, k        # a few update sizes (lr 1.5e-3; sign-like first Adam steps)
        du_ref, du = v - before[k], res[0]["params"][k] - before[k]
        if du_ref.abs().max() > 0:
            worst = max(worst, ((du - du_ref).norm() / du_ref.norm()).item())
    print("worst update L2 difference, 2 ranks x 2 vs 1 x 4:", worst)
    assert worst < 0.1
