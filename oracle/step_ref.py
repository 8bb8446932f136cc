"""ORACLE (test infrastructure only) - one E_align_s2 iteration (mtype 2) on the CPU in plain
torch fp32, composed from oracle/ref_torch.py + oracle/lpips_ref.py (reference
E_align_s2.py:102-221).  Used by tests and by the `cpu_baseline` leg of bench.py; never by the
product path.  `train` selects the reference's train-mode generator pass (quirk Q1: w_avg EMA + style mixing,
model/stylegan2_generator.py:177-191); in that form the WHOLE composition is pinned on the CPU against the reference's own
two-iteration run, tests/golden/step_s2.npz (tests/test_oracle_golden.py::test_step_ref_reproduces_the_reference_run).
Without it G runs in eval mode (the cpu_baseline leg times arithmetic only)."""
import torch

from . import ref_torch as O
from . import lpips_ref as LR


def e_align_step(PG, PE, PL, z, noises, lr=0.0015, state=None, record=None, train=None):
    """PE: dict of leaf tensors with requires_grad=True (updated in place through .data, like
    LREQAdam).  Returns dict with losses.  `state`: optimiser state dict (exp_avg_sq, step); `record`: optional dict that
    receives the encoder gradients of both phases ("grad1", "grad2") for the full-size gradient parity test.
    `train`: dict(new_z, u, cutoff) - the three random draws of the reference's train-mode forward (:185-188, torch.randn_like,
    np.random.uniform, np.random.randint); PG["truncation.w_avg"] is updated in place like the reference's buffer."""
    state = {} if state is None else state
    with torch.no_grad():
        if train is not None:
            wp, w_avg = O.s2_generator_train(PG, z, train["new_z"], train["u"], train["cutoff"])
            PG["truncation.w_avg"].copy_(w_avg)
            imgs1 = O.s2_synthesis(PG, wp)
        else:
            w, wp, imgs1 = O.s2_generator_eval(PG, z)
    const2, w2 = O.enc_forward(PE, imgs1, noises)
    imgs2 = O.s2_synthesis(PG, w2)
    lp = (lambda a, b: LR.lpips(PL, a, b)) if PL is not None else (lambda a, b: torch.zeros(a.shape[0], 1, 1, 1))
    tot = 0
    a_crops = [imgs1, *O.attention_crops(imgs1)]
    b_crops = [imgs2, *O.attention_crops(imgs2)]
    parts = []
    for wgt, x1, x2 in zip((1, 5, 9), a_crops, b_crops):
        l, _ = O.space_loss(x1, x2, lpips_fn=lp)
        parts.append(float(l))
        tot = tot + wgt * l

    def adam(coefs):
        for k, p in PE.items():
            if p.grad is None:
                continue
            st = state.setdefault(k, {"v": torch.zeros_like(p), "t": 0})
            st["t"] += 1
            newp, st["v"] = O.lreq_adam_step(p.data, p.grad, st["v"], st["t"], lr, coef=coefs.get(k))
            p.data.copy_(newp)

    for p in PE.values():
        p.grad = None
    tot.backward(retain_graph=True)
    if record is not None:          # encoder gradients of the image phase (before the optimiser touches the weights)
        record["grad1"] = {k: p.grad.clone() for k, p in PE.items() if p.grad is not None}
    adam(state.get("_coef", {}))
    lw, _ = O.space_loss(wp, w2, image_space=False)
    for p in PE.values():
        p.grad = None
    (lw * 0.01).backward()
    if record is not None:          # latent phase: weights already updated once, activations from before (quirk Q3)
        record["grad2"] = {k: p.grad.clone() for k, p in PE.items() if p.grad is not None}
        record["wp"] = wp
    adam(state.get("_coef", {}))
    return dict(loss_tsa=float(tot), loss_w=float(lw), loss_parts=parts, imgs1=imgs1, imgs2=imgs2.detach(), w2=w2.detach(), wp=wp)
