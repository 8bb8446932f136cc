"""Per-stage kernel tables of one E_align_s2 step (config 3, batch 8, bf16) - dev tool.
  run:     rocprofv3 --kernel-trace -d DIR -o r -- python tools/stage_trace.py run
  report:  python tools/stage_trace.py report <results.db> [top]
`run` brackets every stage of the step with a count-coded run of a marker kernel (rgb_tanh_kernel: a BigGAN kernel the
StyleGAN2 step never launches); `report` cuts the kernel trace at the markers and prints one (kernel, grid) table per stage."""
import os, sys, collections, re
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
STAGES = ["G sample (no grad)", "E forward", "G synthesis (saved)", "image losses (value + gradient)", "latent loss",
          "G synthesis backward", "E backward", "LREQAdam step", "other"]


def run():
    import torch
    import dge_amd
    from dge_amd import e_align, ops, autograd_s2, autograd_enc, autograd_enc_bwd
    from dge_amd.e_align import EAlignStep, build_models
    B = int(os.environ.get("B", "8"))
    G, E, LP = build_models(1024, 16, "bf16", "cuda")
    G.train()
    st = EAlignStep(G, E, LP, batch_size=B)
    for i in range(3): st.step(i)
    torch.cuda.synchronize()
    tiny = torch.zeros(1, 1, 1, 8, device="cuda", dtype=torch.bfloat16)

    def mark(n):
        for _ in range(n + 1): ops.rgb_tanh(tiny)

    def wrap(obj, name, sid):
        orig = getattr(obj, name)

        def f(*a, **k):
            mark(sid)
            r = orig(*a, **k)
            mark(len(STAGES) - 1)
            return r
        setattr(obj, name, f)
    wrap(st.gen, "sample", 0); wrap(autograd_enc, "encoder_forward", 1); wrap(st.gen, "synth", 2)
    wrap(e_align.losses, "image_loss_tsa", 3); wrap(e_align.losses, "space_loss", 4)
    wrap(autograd_s2, "synthesis_backward", 5); wrap(autograd_enc_bwd, "encoder_backward", 6); wrap(st.opt, "step", 7)
    for i in range(int(os.environ.get("STEPS", "3"))): st.step(10 + i)
    torch.cuda.synchronize()


def report(db, top=40):
    import sqlite3
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gcols = [k for k in ("grid_x", "grid_y", "workgroup_x") if k in cols] or ["0"]
    rows = c.execute(f"select name, start, end, {', '.join(gcols)} from kernels order by start").fetchall()
    first = next(i for i, r in enumerate(rows) if "rgb_tanh" in r[0])
    rows = rows[first:]
    stage, run_len, nsteps = len(STAGES) - 1, 0, 0
    agg = [collections.OrderedDict() for _ in STAGES]
    for r in rows:
        if "rgb_tanh" in r[0]:
            run_len += 1
            continue
        if run_len:
            if run_len > len(STAGES): run_len -= len(STAGES)      # an end marker directly followed by the next start marker
            stage = run_len - 1
            nsteps += stage == 0
            run_len = 0
        name = re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", "")).replace("void ", "").replace("unsigned short", "bf16")
        a = agg[stage].setdefault((name[:90],) + tuple(r[3:]), [0, 0.0])
        a[0] += 1; a[1] += (r[2] - r[1]) / 1e3
    nsteps = max(nsteps, 1)
    print(f"# {nsteps} steps; per-step figures")
    for sid, tab in enumerate(agg):
        tot = sum(a[1] for a in tab.values()) / nsteps
        n = sum(a[0] for a in tab.values()) / nsteps
        print(f"\n== {STAGES[sid]}: {tot / 1e3:.3f} ms, {n:.0f} launches")
        for k, a in sorted(tab.items(), key=lambda kv: -kv[1][1])[:top]:
            print(f"{a[0] / nsteps:6.1f} {a[1] / nsteps:9.1f} {a[1] / a[0]:8.1f}  {str(k[1:]):>20}  {k[0]}")


if __name__ == "__main__":
    if sys.argv[1] == "run": run()
    else: report(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40)
