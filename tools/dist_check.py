"""Dev tool: the RCCL gradient-exchange path (early async bucket + remainder) on ONE GPU with a 1-rank group, against the
non-distributed step and against the run-to-run spread of the non-distributed step (atomics + sign-like first Adam steps)."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", DGE_FORCE_DIST="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from dge_amd.e_align import EAlignStep, build_models
res = []
for force in ("1", "0", "0"):
    os.environ["DGE_FORCE_DIST"] = force
    G, E, LP = build_models(256, 64, "bf16", "cuda", seed=0, fmaps_base=8 << 10)
    G.train()
    st = EAlignStep(G, E, LP, batch_size=2)
    assert st.dist_on == (force == "1")
    torch.manual_seed(5)
    for it in range(3):
        r = st.step(it)
    if force == "1":
        assert st._layout["n_early"] > 0.8 * st._flat.numel(), (st._layout["n_early"], st._flat.numel())
    res.append(({k: v.detach().clone() for k, v in E.state_dict().items()}, float(r["loss_tsa"]), float(r["loss_w"])))
d = max(float((res[0][0][k] - res[1][0][k]).abs().max()) for k in res[0][0])
d2 = max(float((res[1][0][k] - res[2][0][k]).abs().max()) for k in res[0][0])
print("non-dist vs non-dist:", d2)
print("max param diff dist vs non-dist after 3 steps:", d, "losses", res[0][1:], res[1][1:])
dist.destroy_process_group()
