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

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """grad_scale: optional device scalar multiplied into every gradient (DDP mean)."""
        loss = closure() if closure is not None else None
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
                step_size = group["lr"] * math.sqrt(1 - group["beta_2"] ** state["step"])
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
            check(lib().dge_lreq_adam_multi(n, PA, GA, VA, NA, SA, group["beta_2"], group["eps"], gsc, _stream()),
                  "dge_lreq_adam_multi")
        return loss
