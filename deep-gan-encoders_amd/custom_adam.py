"""LREQAdam (reference model/utils/custom_adam.py:6-76): Adam with beta1 == 0 and a per-parameter
`lr_equalization_coef`, executed as one multi-tensor HIP launch per step."""
import ctypes as C
import math

import torch
from torch.optim.optimizer import Optimizer

from ._lib import lib, check
from .ops import _stream


class LREQAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.0, 0.99), eps=1e-8, weight_decay=0):
        beta_2 = betas[1]
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 == betas[0]:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= beta_2 < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(beta_2))
        if weight_decay != 0:
            raise ValueError("weight_decay != 0 is not used on the E_align path (the reference's branch reads the "
                             "non-existent attribute p.coef, custom_adam.py:56)")
        super().__init__(params, dict(lr=lr, beta_2=beta_2, eps=eps, weight_decay=weight_decay))
        # hipGraph mode (graph_begin): the sqrt(1 - beta2^t) factor of the k-th step() call inside the captured region is
        # read from a device scalar that the host refreshes before every replay
        self._graph_corr = None
        self._graph_call = 0

    def graph_begin(self, calls_per_replay, device):
        """Call once before capturing a region that contains `calls_per_replay` step() calls."""
        self._graph_corr = torch.ones(calls_per_replay, dtype=torch.float32, device=device)
        self._graph_call = 0
        self._graph_t = None            # a second capture starts from the optimizer's current step count again

    def graph_advance(self):
        """Before every capture / replay: advance all step counters by the region's step() calls and upload the factors.
        Parameters that never receive gradients are not counted (they have no state), as in the reference (:35-36)."""
        n = self._graph_corr.numel()
        group = self.param_groups[0]
        t0 = max([self.state[p]["step"] for p in group["params"] if p in self.state and len(self.state[p])] or [0])
        if getattr(self, "_graph_t", None) is None:
            self._graph_t = t0
        vals = [math.sqrt(1 - group["beta_2"] ** (self._graph_t + k + 1)) for k in range(n)]
        self._graph_t += n
        # a fresh pageable host tensor per call: the copy is staged before it returns, so the host may run several
        # replays ahead of the device without overwriting factors that have not been consumed yet
        self._graph_corr.copy_(torch.tensor(vals, dtype=torch.float32))
        self._graph_call = 0

    def graph_snapshot(self):
        """Host-side counters before a stream capture ..."""
        return (getattr(self, "_graph_t", None), {p: st["step"] for p, st in self.state.items() if len(st)})

    def graph_restore(self, snap):
        """... and their rollback after it: step() calls made while a hipGraph is being CAPTURED advance the step counters on the
        host although no update runs on the device (the captured iteration is recorded, not executed); left in place, every
        replay would apply sqrt(1 - beta2^t) one iteration ahead of the eager sequence.  Parameters that got their state inside
        the capture go back to step 0."""
        self._graph_t, steps = snap
        for p, st in self.state.items():
            if len(st):
                st["step"] = steps.get(p, 0)

    def graph_count_replay(self):
        """After every graph replay: the region's step() calls did not run on the host, so advance the per-parameter step
        counters here (optimizer.state_dict() and a later eager step() then see the true t)."""
        n = self._graph_corr.numel()
        for st in self.state.values():
            if len(st):
                st["step"] += n

    def graph_reset(self):
        """Fresh optimizer state without changing any device address (embedding_img.py:83 between images)."""
        for st in self.state.values():
            if len(st):
                st["exp_avg_sq"].zero_()
                st["step"] = 0
        self._graph_t = 0

    @torch.no_grad()
    def tick(self):
        """One step() with ZERO gradients on every parameter that already has optimizer state: t += 1, v *= beta_2, p unchanged
        (custom_adam.py:35-62 with grad == 0).  This is what the reference's first optimizer step of a stage-1 iteration does under
        torch < 2.0, whose `zero_grad()` zero-fills gradient tensors instead of dropping them (E_align_cropping_s1.py:203-205: the
        image-space loss reaches no encoder parameter, so every .grad is the zero tensor left by zero_grad)."""
        for group in self.param_groups:
            live = [p for p in group["params"] if len(self.state.get(p, {}))]
            if not live:
                continue
            n_max = max(p.numel() for p in live)
            z = getattr(self, "_zero_grad_buf", None)
            if z is None or z.numel() < n_max or z.device != live[0].device:
                z = self._zero_grad_buf = torch.zeros(n_max, dtype=torch.float32, device=live[0].device)
            saved = [(p, p.grad) for p in group["params"]]
            try:
                for p in group["params"]:
                    if p.dtype != torch.float32:
                        raise RuntimeError("LREQAdam.tick: float32 parameters only")
                    p.grad = z[:p.numel()].view_as(p) if len(self.state.get(p, {})) else None
                gens = [(p, getattr(p, "_dge_gen", 0)) for p in group["params"]]
                self.step()
                for p, gen in gens:          # no value changed: the packed weight copies stay valid (step() marks them stale)
                    p._dge_gen = gen
            finally:
                for p, g in saved:
                    p.grad = g

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """grad_scale: optional device scalar multiplied into every gradient (DDP mean)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():         # the closure re-evaluates the model (torch.optim convention)
                loss = closure()
        for group in self.param_groups:
            ps, gs, vs, ns, steps = [], [], [], [], []
            keep = []
            for p in group["params"]:
                if p.grad is None:            # skipped without advancing its step counter (:35-36)
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                if not p.is_cuda:
                    raise RuntimeError("LREQAdam runs on the HIP device only")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg_sq"] = torch.zeros_like(p.data, memory_format=torch.contiguous_format)
                state["step"] += 1
                if self._graph_corr is None:
                    step_size = group["lr"] * math.sqrt(1 - group["beta_2"] ** state["step"])
                else:
                    step_size = group["lr"]                       # x device factor of this call
                if hasattr(p, "lr_equalization_coef"):
                    step_size *= p.lr_equalization_coef
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(g)
                # the kernel writes through the raw pointer (like the reference's p.data update, it does not touch
                # autograd's version counter): bump our own generation so packed weight copies are rebuilt
                p._dge_gen = getattr(p, "_dge_gen", 0) + 1
                ps.append(p.data_ptr()); gs.append(g.data_ptr()); vs.append(state["exp_avg_sq"].data_ptr())
                ns.append(p.numel()); steps.append(step_size)
            n = len(ps)
            if n == 0:
                continue
            PA = (C.c_void_p * n)(*ps); GA = (C.c_void_p * n)(*gs); VA = (C.c_void_p * n)(*vs)
            NA = (C.c_long * n)(*ns); SA = (C.c_float * n)(*steps)
            gsc = C.c_void_p(grad_scale.data_ptr()) if grad_scale is not None else None
            smul = None
            if self._graph_corr is not None:
                smul = C.c_void_p(self._graph_corr.data_ptr() + 4 * self._graph_call)
                self._graph_call += 1
            check(lib().dge_lreq_adam_multi(n, PA, GA, VA, NA, SA, group["beta_2"], group["eps"], gsc, smul, _stream()),
                  "dge_lreq_adam_multi")
        return loss
