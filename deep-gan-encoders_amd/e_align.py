"""E_align stage-2 encoder training step (reference E_align_s2.py:23-299) on the HIP path.

`EAlignStep.step()` is one iteration of the reference's hot loop (:102-221) for mtype 2
(StyleGAN2): seed -> z -> G (no grad, train-mode quirks kept) -> E -> G.synthesis (grad) ->
loss_imgs + 5*loss_medium + 9*loss_small -> backward(retain_graph) -> LREQAdam.step ->
0.01*loss_w -> backward -> step.  One process per GPU; with torch.distributed initialised the
encoder gradients are all-reduced over RCCL once per phase (flat 97 MB bucket at FFHQ-1024) and
the batch-coupled loss terms (cosine, means) use globally reduced sums so that N ranks x B
images reproduce a single-process run at batch N*B.
"""
import math

import numpy as np
import os

import torch
import torch.distributed as dist

from . import losses
from .custom_adam import LREQAdam


def set_seed(seed):
    """training_utils.py:46-52"""
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    from . import ops
    ops.noise_seed(seed)            # the step's device noise is counter-based: (seed, draw number, global element index)


class _StyleGAN2Adapter:
    """mtype 2: generator(z, trunc...) -> dict, generator.synthesis(wp) -> dict (E_align_s2.py:110-115,160)"""

    def __init__(self, generator):
        self.G = generator

    mix_mask = None          # device [L] mask: set by EAlignStep in hipGraph mode (static kernel sequence)

    new_z = None             # parity runs: the reference's own second latent of the style mixing (stylegan2_generator.py:187)

    def sample(self, z, noises=None):
        r = self.G(z, trunc_psi=0.7, trunc_layers=8, randomize_noise=False, mix_mask=self.mix_mask, new_z=self.new_z)
        return r["image"], r["wp"]

    def synth(self, w, noises=None):
        return self.G.synthesis(w)["image"]


class _StyleGAN1Adapter:
    """mtype 1: w1 = Gm(z, coefs_m=coefs); imgs = Gs.forward(w, lod) with lod = log2(img_size)-2 (E_align_s2.py:27-41,105-108,158)"""

    def __init__(self, Gs, Gm):
        self.G, self.Gm = Gs, Gm
        n = 2 * Gs.layer_count
        layer_idx = torch.arange(n)[None, :, None]
        ones = torch.ones(layer_idx.shape, dtype=torch.float32)
        self.coefs = torch.where(layer_idx < n // 2, 0.7 * ones, ones)      # truncation psi on the first half of the layers
        self.lod = Gs.layer_count - 1

    def sample(self, z, noises=None):
        w1 = self.Gm(z, coefs_m=self.coefs)
        return self.G.forward(w1, self.lod, noises=noises), w1

    def synth(self, w, noises=None):
        return self.G.forward(w, self.lod, noises=noises)


class _PGGANAdapter:
    """mtype 3: w1 = z; imgs1 = generator(w1)['image'] (E_align_s2.py:134-138).  The script's second pass calls
    `generator.synthesis(w2)` (:160), which PGGANGenerator does not have; the evident intent
    `generator(w2)['image']` is what runs here."""

    def __init__(self, generator):
        self.G = generator

    def sample(self, z, noises=None):
        return self.G(z)["image"], z

    def synth(self, w, noises=None):
        return self.G(w)["image"]


def truncated_noise_sample(batch_size=1, dim_z=128, truncation=1.0, seed=None):
    """training_utils.py:32-44 (scipy truncnorm on a seeded RandomState)"""
    from scipy.stats import truncnorm
    state = None if seed is None else np.random.RandomState(seed)
    return truncation * truncnorm.rvs(-2, 2, size=(batch_size, dim_z), random_state=state).astype(np.float32)


class _BigGANAdapter:
    """mtype 4 (E_align_s2.py:139-150,155,162): z = 0.4 * truncnorm(seed), one class id per batch drawn with
    np.random.randint(1000) after set_seed, truncation = float32 tensor 0.4 (kept on the host: its BN-row arithmetic is the
    reference's float32 division); the encoder is conditioned on the generator's condition vector."""

    def __init__(self, generator):
        self.G = generator
        self.truncation = torch.tensor(0.4, dtype=torch.float)
        self.conditions = self.const1 = None

    def draw(self, iteration, n, dev):
        z = truncated_noise_sample(truncation=0.4, batch_size=n, dim_z=self.G.config.z_dim, seed=iteration % 30000)
        self.set_label(int(np.random.randint(1000)), dev)
        return torch.tensor(z, dtype=torch.float)

    def set_label(self, flag, dev):
        self.flag = flag

    def sample(self, z, noises=None):
        B = z.shape[0]
        self.conditions = torch.zeros(B, self.G.config.num_classes, device=z.device)
        self.conditions[:, self.flag] = 1.0
        imgs1, self.const1 = self.G(z, self.conditions, self.truncation)
        return imgs1, z

    def synth(self, w, noises=None):
        return self.G(w, self.conditions, self.truncation)[0]


class _StagedWork:
    """all-reduce of a device tensor through a host copy (gloo builds without device support): wait() writes the result back"""

    def __init__(self, t, host, work):
        self.t, self.host, self.work = t, host, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.t.copy_(self.host)


def _all_reduce(t, async_op=False):
    """Sum over ranks, in place.  RCCL ("nccl") reduces device tensors directly; with the gloo backend (the 2-process parity
    test on one GPU, CPU-only debugging) device tensors are staged through the host."""
    if t.is_cuda and dist.get_backend() == "gloo":
        host = t.detach().cpu()
        work = dist.all_reduce(host, op=dist.ReduceOp.SUM, async_op=async_op)
        st = _StagedWork(t, host, work if async_op else None)
        if async_op:
            return st
        st.wait()
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


class EAlignStep:
    def __init__(self, generator, E, lpips_model, lr=0.0015, beta_1=0.0, batch_size=2, z_dim=512,
                 reference_noise=False, exact_ddp=True, mapping=None, stage=2, zero_grad_to_none=True):
        """`generator`: StyleGAN2Generator (mtype 2), the StyleGAN1 synthesis network Gs together with
        `mapping` = Gm (mtype 1), a PGGANGenerator (mtype 3) or a BigGAN (mtype 4; z_dim is taken from its config).
        `stage`: 2 = E_align_s2.py (image phase 1/5/9-weighted with gradient, then the latent phase); 1 = the stage-1 variant
        E_align_cropping_s1.py:185-218: the image-space losses are evaluated on detached inputs and summed unweighted (they are
        reported, not trained on: no gradient reaches E, the script's first optimizer step changes nothing) and only the
        latent phase updates the encoder.
        `zero_grad_to_none` (stage 1 only): True = `optimizer.zero_grad()` of torch >= 2.0 drops the gradients, so the script's
        first optimizer step finds none and changes nothing (tests/golden/step_s1.npz).  False = the torch < 2.0 default the
        reference's pinned environment has (python 3.7, torch 1.8 .. 1.13): gradients are zero-FILLED, so from the second
        iteration on that step runs LREQAdam with zero gradients - every step counter advances and every second moment decays
        by beta_2 (custom_adam.py:35-62), which makes the latent-phase updates ~1.4x larger in steady state
        (tests/golden/step_s1_legacy.npz).  Stage 2 is unaffected: both of its phases give every trained parameter a gradient."""
        self.zero_grad_to_none = bool(zero_grad_to_none)
        if stage not in (1, 2):
            raise ValueError("EAlignStep: stage must be 1 or 2")
        self.stage = stage
        from .pggan_generator import PGGANGenerator
        from .biggan_generator import BigGAN
        self.G, self.E, self.lpips = generator, E, lpips_model
        if mapping is not None:
            self.gen = _StyleGAN1Adapter(generator, mapping)
        elif isinstance(generator, PGGANGenerator):
            self.gen = _PGGANAdapter(generator)
        elif isinstance(generator, BigGAN):
            self.gen = _BigGANAdapter(generator)
            z_dim = generator.config.z_dim
        else:
            self.gen = _StyleGAN2Adapter(generator)
        self.opt = LREQAdam([{"params": E.parameters()}], lr=lr, betas=(beta_1, 0.99), weight_decay=0)
        self.batch_size, self.z_dim = batch_size, z_dim
        self.reference_noise = reference_noise      # True: CPU-generated noise in the reference's order (Q6)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        # DGE_FORCE_DIST=1 exercises the collective code path on a 1-rank group (single-GPU validation of the DDP wiring)
        import os
        self.dist_on = self.world > 1 or (os.environ.get("DGE_FORCE_DIST") == "1" and dist.is_initialized())
        self.exact_ddp = exact_ddp
        self.dev = next(E.parameters()).device
        self._flat = None
        self.last = {}
        if self.dist_on:
            E.__dict__["_early_grad_hook"] = self.early_reduce        # autograd_enc_bwd calls it after the deep blocks
        from . import ops
        ops.noise_dp(self.rank, self.world)      # device noise = this rank's rows of the global-batch draw

    # ------------------------------------------------------------------ DDP gradient exchange
    def _flat_views(self, early_names=()):
        """One flat f32 bucket for all encoder gradients: the parameters named in `early_names` first (their gradients exist
        long before the backward ends), the rest after; returns {name: view}."""
        lay = getattr(self, "_layout", None)
        if lay is None or lay["early"] != tuple(early_names):
            named = dict(self.E.named_parameters())
            order = [n for n in early_names if n in named] + [n for n in named if n not in set(early_names)]
            n_early = sum(named[n].numel() for n in early_names if n in named)
            total = sum(p.numel() for p in named.values())
            self._flat = torch.empty(total, dtype=torch.float32, device=self.dev)
            views, off = {}, 0
            for n in order:
                views[n] = self._flat[off:off + named[n].numel()].view_as(named[n])
                off += named[n].numel()
            lay = self._layout = dict(early=tuple(early_names), views=views, n_early=n_early, named=named)
        return lay

    def early_reduce(self, grads):
        """Called from inside the encoder backward as soon as the gradients of the deep (512-channel) blocks exist: > 90 % of
        the 97 MB bucket.  Their all-reduce is issued asynchronously (RCCL's own stream) and runs under the backward of the
        high-resolution blocks, which is most of the backward's time; `_sync_grads` exchanges the remainder and joins."""
        if not getattr(self, "dist_on", self.world > 1) or not grads:
            return
        lay = self._flat_views(tuple(grads.keys()))
        names = [n for n in lay["early"] if grads.get(n) is not None]
        if len(names) != len(lay["early"]):
            return                                          # a different set than the layout was built for: leave it to _sync_grads
        torch._foreach_copy_([lay["views"][n] for n in names], [grads[n] for n in names])
        # Stream order: the copies above are queued on the CURRENT (compute) stream; ProcessGroupNCCL enqueues every collective on
        # its own stream behind an event it records on the current stream at call time (ProcessGroupNCCL::collective ->
        # syncStream), so the all-reduce reads the bucket after the copies without an explicit wait_stream here.  The rest of
        # the backward never touches [0, n_early) of the bucket (disjoint views), the bucket itself is owned by `self` (no
        # allocator reuse while the collective runs), and `_sync_grads` joins with work.wait(), which makes the compute stream wait
        # for RCCL's before the optimizer reads the sums.
        self._early_work = _all_reduce(self._flat[:lay["n_early"]], async_op=True)

    def _sync_grads(self):
        """All-reduce (sum) of every encoder gradient through the flat bucket; p.grad become views of it."""
        if not getattr(self, "dist_on", self.world > 1):
            return None
        work = self.__dict__.pop("_early_work", None)
        lay = self._flat_views(self._layout["early"] if (work is not None) else getattr(self, "_layout", {"early": ()})["early"])
        early = set(lay["early"]) if work is not None else set()
        rest = [(n, p) for n, p in lay["named"].items() if p.grad is not None and n not in early]
        # one multi-tensor copy instead of ~100 small ones (they sit on the critical path in front of the collective)
        if rest:
            torch._foreach_copy_([lay["views"][n] for n, _ in rest], [p.grad for _, p in rest])
        missing = [n for n, p in lay["named"].items() if p.grad is None and n not in early]
        for n in missing:                                   # parameters without a gradient this phase contribute zeros
            lay["views"][n].zero_()
        # comm_stats (bench.py --gpus N): events on the compute stream around the part of the exchange the step waits for - the
        # remainder bucket plus whatever of the early bucket the backward did not cover
        cs = getattr(self, "comm_stats", None)
        if cs is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if work is not None:
            _all_reduce(self._flat[lay["n_early"]:])
            work.wait()
        else:
            _all_reduce(self._flat)
        if cs is not None:
            e1.record()
            cs["events"].append((e0, e1))
            cs["early_bytes"] = 4 * lay["n_early"] if work is not None else 0
            cs["remainder_bytes"] = 4 * (self._flat.numel() - (lay["n_early"] if work is not None else 0))
        for n, p in lay["named"].items():
            if p.grad is not None:
                p.grad = lay["views"][n]
        # exact mode: every rank differentiated the GLOBAL loss w.r.t. its own samples -> sum.
        # plain mode: local losses -> mean.
        return None if self.exact_ddp else torch.full((1,), 1.0 / self.world, device=self.dev)

    # ------------------------------------------------------------------ hipGraph replay of the iteration
    def capture(self, warmup=2, start=0):
        """Captures one iteration into a hipGraph (single-GPU runs; the ≈1300 launches of a step cost ≈18 ms of host time,
        which bounds the step at the reference's default batch of 2).  Host-side decisions of an iteration become device
        inputs: z (static buffer), the style-mixing mask (StyleGAN2 train mode, same np.random draw order as the
        reference) and Adam's sqrt(1 - beta2^t) factors.  Encoder / StyleGAN1 noise comes from torch's graph-safe
        device generator.  `warmup` real iterations run inside this call (plus one eager iteration in front of them in the legacy
        stage-1 form); the captured iteration itself is only recorded.  The real iterations are numbered `start`, `start` + 1, ...
        (the number seeds z and the mixing mask, training_utils.py:46-52); `self._g_iter` is the number of the NEXT iteration when
        this returns - a training loop continues there (train() below) instead of repeating the warm-up's iterations."""
        if self.dist_on:
            raise RuntimeError("hipGraph capture is offered for single-process runs only (collectives are not captured)")
        # Capturing after EAGER steps of the same encoder used to end in a segmentation fault inside capture_end (round 2:
        # "crashed the runtime once"; reproduced with faulthandler in round 3).  Cause: the eager step's results (imgs2, w2) kept
        # their autograd graph alive, and with it the AccumulateGrad nodes of E's parameters, created on the default stream.
        # The captured iteration runs on a side stream; autograd re-uses the live nodes and inserts its cross-stream event
        # record / wait between the default stream and the capturing one - an illegal dependency for a capture.  step() now
        # returns detached results; dead graphs of earlier iterations are collected here before the warm-up, so that the
        # warm-up iterations create fresh accumulator nodes on the capture's side stream.
        import gc
        self.last = {}
        gc.collect()
        if isinstance(self.gen, _BigGANAdapter):
            raise RuntimeError("hipGraph capture is not offered for --mtype 4: z is a scipy truncnorm draw and the class id a host "
                               "decision of every iteration (E_align_s2.py:139-150); run the eager step")
        from . import ops
        ops.noise_graph_begin(self.dev)          # noise kernels read their seed from a device scalar from here on
        B = self.batch_size
        self._g_z = torch.zeros(B, self.z_dim, device=self.dev)
        if isinstance(self.gen, _StyleGAN2Adapter):
            self.gen.mix_mask = torch.zeros(self.G.num_layers, device=self.dev)
        # optimizer calls of one iteration: two in stage 2 (image phase, latent phase), one in stage 1 - two again in its legacy
        # zero_grad form (tick() + step()), where the tick is skipped while no parameter has state yet: one eager iteration
        # first gives every parameter its state, so that every captured / replayed iteration makes the same number of calls
        calls = 2 if (self.stage == 2 or not self.zero_grad_to_none) else 1
        self._g_iter = int(start)
        if self.stage == 1 and not self.zero_grad_to_none and not any(len(st) for st in self.opt.state.values()):
            self.step(self._g_iter)
            self._g_iter += 1
        self.opt.graph_begin(calls, self.dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._graph_inputs(self._g_iter); self._g_iter += 1
                self.step(0, z=self._g_z)
        torch.cuda.current_stream().wait_stream(side)
        from .autograd_enc import prime_pack_tables
        prime_pack_tables(self.E)
        self._graph = torch.cuda.CUDAGraph()
        # the captured iteration is RECORDED, not executed: the host-side counters it advances (Adam's t, the iteration number that
        # seeds z and the mixing mask) are rolled back, so that the first replay is iteration `warmup` of the sequence
        snap = self.opt.graph_snapshot()
        self._graph_inputs(self._g_iter)
        with torch.cuda.graph(self._graph):
            self._g_out = self.step(0, z=self._g_z)
        self.opt.graph_restore(snap)
        return self._g_out

    def _graph_inputs(self, iteration):
        set_seed(iteration % 30000)
        zg = torch.randn(self.batch_size * self.world, self.z_dim)
        self._g_z.copy_(zg[self.rank * self.batch_size:(self.rank + 1) * self.batch_size])
        if getattr(self.gen, "mix_mask", None) is not None:
            from .stylegan2_generator import mixing_mask
            self.gen.mix_mask.copy_(mixing_mask(self.G.num_layers))
        self.opt.graph_advance()

    def replay(self, iteration=None):
        it = self._g_iter if iteration is None else iteration
        self._graph_inputs(it)
        self._g_iter = it + 1
        self._graph.replay()
        self.opt.graph_count_replay()
        return self._g_out

    # ------------------------------------------------------------------ one iteration
    def _upload(self, t):
        """Host tensor -> device without stalling the host: `t.to(device)` from pageable memory waits for the stream to drain
        (the whole previous step), after which the GPU idles until the host has queued work again.  z (drawn on the CPU after
        set_seed, like the reference, E_align_s2.py:103-104) goes through a small ring of pinned staging buffers instead; a
        buffer is reused only after the copy that read it has completed."""
        if t.is_cuda:
            return t.to(self.dev)
        ring = self.__dict__.setdefault("_pin_ring", {"i": 0, "slots": [None] * 4})
        k = ring["i"] = (ring["i"] + 1) % len(ring["slots"])
        slot = ring["slots"][k]
        if slot is None or slot[0].shape != t.shape or slot[0].dtype != t.dtype:
            slot = ring["slots"][k] = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True), None]
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0].copy_(t)
        out = slot[0].to(self.dev, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return out

    def _draw_and_sample(self, iteration):
        """The head of an iteration with default inputs (E_align_s2.py:102-115): set_seed, z on the host, generator pass under
        no_grad.  Runs at the start of step(iteration) - or, with prefetch_next, beside the second backward of step(iteration - 1)."""
        B = self.batch_size
        set_seed(iteration % 30000)
        # every rank draws the same global z and takes its slice
        zg = torch.randn(B * self.world, self.z_dim)
        z = self._upload(zg[self.rank * B:(self.rank + 1) * B])
        with torch.no_grad():
            imgs1, w1 = self.gen.sample(z, None)
        return z, imgs1, w1

    def cancel_prefetch(self):
        """Drops a generator pass that step(..., prefetch_next=True) issued for an iteration that will not run (end of a loop that
        could not know it was at its end).  The generator's own state has seen that pass (StyleGAN2 train mode: one w_avg update) and
        so have the random generators; returns the iteration number it belonged to, or None."""
        pref = self.__dict__.pop("_pref", None)
        return None if pref is None else pref[0]

    def _prefetch_ok(self):
        # (data parallel: the prefetched pass's only collective is the w_avg mean - 512 floats behind the mapping network, the first
        #  ~0.1 ms of the pass - issued from the side stream in the same program order on every rank.  ProcessGroupNCCL runs a group's
        #  collectives in call order on its own stream, so this one sits in front of the iteration's loss sums and gradient buckets and is
        #  long done when they are issued, several ms later: no exposed serialisation is expected.  Measured only with a one-rank RCCL
        #  group and two gloo ranks (tests/test_ddp_gpu.py); `--no-prefetch` / prefetch_next=False is the serial form.)
        from . import ops
        return (_SIDE_STREAMS and self.stage == 2 and self.dev.type == "cuda" and not isinstance(self.gen, _BigGANAdapter) and not self.reference_noise
                and not ops.is_deterministic() and not torch.cuda.is_current_stream_capturing())

    def step(self, iteration, z=None, noises=None, gen_noises=(None, None), new_z=None, prefetch_next=False):
        """`noises`: optional encoder noise tensors; `gen_noises`: optional (first, second) generator noise lists for
        generators that draw noise per call (StyleGAN1); `new_z`: the style-mixing latent of StyleGAN2's train mode -- all only
        for parity runs against captured reference noise.
        `prefetch_next`: the caller's promise that its next call is step(iteration + 1) with default inputs (a training loop).  The
        generator pass that opens that iteration - set_seed(iteration + 1), z, G(z) under no_grad: the training DATA of the encoder, which
        depends on nothing the encoder does (E_align_s2.py:102-115) - is then issued on a side stream beside this iteration's image
        losses and backward passes, whose low-resolution launches leave most of the chip idle (measured at batch 8, same box: 22.07 ->
        21.56 ms per step).  Same work per iteration, same numbers: nothing between that point (behind E(imgs1) and synthesis(w2), the
        last consumers of random numbers and of the generator's state in an iteration) and the next iteration's start draws a random
        number or reads w_avg, so every draw and the w_avg update happen in the order of the serial loop.  A step that finds a
        prefetched pass it was not promised (another iteration number, explicit inputs) raises instead of silently using or dropping it."""
        G, E = self.G, self.E
        B = self.batch_size
        from . import ops
        # the promise is checked before this call has any side effect: a mismatch leaves the prefetched pass where it is
        # (cancel_prefetch() drops it) and the step object as it was
        pref = self.__dict__.get("_pref")
        if pref is not None and (pref[0] != iteration or z is not None or noises is not None or new_z is not None or gen_noises != (None, None)):
            raise RuntimeError(f"step({iteration}): the previous step prefetched the generator pass of iteration {pref[0]} with default "
                               "inputs (prefetch_next=True is a promise about the next call)")
        self.__dict__.pop("_pref", None)
        if isinstance(self.gen, _StyleGAN2Adapter):
            self.gen.new_z = new_z
        ops.zero_arena_begin(self.dev)       # one memset for all of this step's accumulation buffers
        big = isinstance(self.gen, _BigGANAdapter)
        if pref is not None:
            # the pass was issued on the side stream: the consumer orders itself behind the event recorded there (the issuing step's
            # own join at its end may not have run if that step raised in between)
            torch.cuda.current_stream(self.dev).wait_event(pref[4])
            z, imgs1, w1 = pref[1:4]
        else:
            if z is None or not z.is_cuda:
                set_seed(iteration % 30000)
            if z is None:
                # every rank draws the same global z and takes its slice
                zg = self.gen.draw(iteration, B * self.world, self.dev) if big else torch.randn(B * self.world, self.z_dim)
                z = zg[self.rank * B:(self.rank + 1) * B]
            z = self._upload(z)
        # the re-pack of the encoder's conv weights (stale since the last optimizer step) beside the generator's first pass:
        # an HBM-bound copy next to small-grid low-resolution layers; joined in front of the encoder
        pack_side = None
        if (self.dev.type == "cuda" and _SIDE_STREAMS and _PACK_STREAM and not ops.is_deterministic() and E.__dict__.get("_pack_cache")
                and (B * imgs_px(G) >= (4 << 20) or torch.cuda.is_current_stream_capturing())):
            from .autograd_enc import refresh_packs
            if getattr(self, "_pack_stream", None) is None:
                self._pack_stream = torch.cuda.Stream(device=self.dev)
            pack_side, main = self._pack_stream, torch.cuda.current_stream(self.dev)
            pack_side.wait_stream(main)
            with torch.cuda.stream(pack_side):
                refresh_packs(E)
        if pref is None:
            with torch.no_grad():
                imgs1, w1 = self.gen.sample(z, gen_noises[0])
        if pack_side is not None:
            torch.cuda.current_stream(self.dev).wait_stream(pack_side)
        if noises is None and self.reference_noise:
            from .autograd_enc import draw_noises
            noises = [n.to(self.dev) for n in draw_noises(E, B, imgs1.shape[2], "cpu")]
        const2, w2 = E(imgs1, self.gen.const1, noises=noises) if big else E(imgs1, noises=noises)
        imgs2 = self.gen.synth(w2, gen_noises[1])

        gctx = losses.GlobalBatch(self.world) if (self.dist_on and self.exact_ddp) else None
        pf_side = None

        def issue_prefetch():
            # the next iteration's generator pass on a side stream (see the docstring).  The side stream starts behind everything
            # queued so far; its results are handed to the next call after the join at the end of this one.
            nonlocal pf_side
            if getattr(self, "_pf_stream", None) is None:
                self._pf_stream = torch.cuda.Stream(device=self.dev)
            pf_side, main = self._pf_stream, torch.cuda.current_stream(self.dev)
            pf_side.wait_stream(main)
            with torch.cuda.stream(pf_side):
                nxt = self._draw_and_sample(iteration + 1)
                done = torch.cuda.Event()
                done.record(pf_side)
            for t in nxt:
                t.record_stream(main)
            self._pref = (iteration + 1,) + tuple(nxt) + (done,)
        # (a parity run that injects its own style-mixing latent leaves it on the adapter: the next iteration's pass must not see it)
        do_pf = prefetch_next and self._prefetch_ok() and getattr(self.gen, "new_z", None) is None
        if do_pf and _PREFETCH_AT == "loss":
            issue_prefetch()
        if self.stage == 1:
            # E_align_cropping_s1.py:185-203: .detach().clone() on every loss input, loss_tsa = imgs + medium + small
            with torch.no_grad():
                loss_tsa, info_img = losses.image_loss_tsa(imgs1, imgs2.detach(), self.lpips, weights=(1.0, 1.0, 1.0), global_batch=gctx)
            if not self.zero_grad_to_none:
                self.opt.tick()          # E_align_cropping_s1.py:203-205 under torch < 2.0: optimizer step on zero-filled gradients
        else:
            loss_tsa, info_img = losses.image_loss_tsa(imgs1, imgs2, self.lpips, global_batch=gctx)
            self.opt.zero_grad()
            if do_pf and _PREFETCH_AT == "bwd1":
                issue_prefetch()
            loss_tsa.backward(retain_graph=True)
            gs = self._sync_grads()
            self.opt.step(grad_scale=gs)

        loss_w, info_w = losses.space_loss(w1, w2, image_space=False, global_batch=gctx)
        loss_mtv = loss_w * 0.01
        self.opt.zero_grad()
        if do_pf and pf_side is None:
            issue_prefetch()
        loss_mtv.backward()
        gs = self._sync_grads()
        self.opt.step(grad_scale=gs)
        if pf_side is not None:
            torch.cuda.current_stream(self.dev).wait_stream(pf_side)
        ops.zero_arena_end()
        # (detached: a result that kept its grad_fn would keep this iteration's autograd graph alive - and with it the
        #  AccumulateGrad nodes of E's parameters, bound to the stream they were created on; see capture())
        det = lambda t: t.detach() if torch.is_tensor(t) else t
        self.last = dict(imgs1=imgs1, imgs2=det(imgs2), w1=det(w1), w2=det(w2), const2=det(const2), loss_tsa=loss_tsa.detach(),
                         info_img=info_img, loss_w=loss_w.detach(), info_w=info_w)
        return self.last


_SIDE_STREAMS = os.environ.get("DGE_SIDE_STREAMS", "1") != "0"
# where step(prefetch_next=True) issues the next iteration's generator pass: beside the image losses ("loss"), the first backward
# ("bwd1") or the second backward ("bwd2").  Same box, batch 8, two rounds: serial 22.07 / 22.23 ms, bwd2 21.70 / 21.81, bwd1 21.71 / 21.78, loss 21.56 / 21.67
_PREFETCH_AT = os.environ.get("DGE_PREFETCH_AT", "loss")
# the early weight re-pack beside the generator's first pass: - 0.13 ms in round 3, + 0.08 ms against this round's kernels (three
# same-box pairs, 24.08 vs 24.17 ms): opt-in
_PACK_STREAM = os.environ.get("DGE_PACK_STREAM", "0") == "1"


def imgs_px(G):
    """pixels of one generated image (resolution attribute of the generator families; 0 when unknown)"""
    r = getattr(G, "resolution", None) or getattr(G, "img_size", None) or 0
    return int(r) * int(r)


def build_models(img_size=1024, start_features=16, compute_dtype="bf16", device="cuda", lpips=True, seed=0,
                 fmaps_base=32 << 10, fmaps_max=512, enc_maxf=512):
    """Models of BASELINE config 3 with seeded random-init weights (no checkpoints ship)."""
    from .stylegan2_generator import StyleGAN2Generator
    from .encoder import BE
    from .lpips import LPIPS
    torch.manual_seed(seed)
    G = StyleGAN2Generator(img_size, fmaps_base=fmaps_base, fmaps_max=fmaps_max, compute_dtype=compute_dtype).to(device)
    for p in G.parameters():
        p.requires_grad_(False)          # G weight gradients are never used
    with torch.no_grad():
        for name, p in G.named_parameters():
            if name.endswith("noise_strength"):
                p.fill_(0.05)
    E = BE(startf=start_features, maxf=enc_maxf, layer_count=int(math.log2(img_size) - 1), compute_dtype=compute_dtype).to(device)
    LP = LPIPS(compute_dtype=compute_dtype).to(device) if lpips else None
    return G, E, LP


def build_models_sg1(img_size=256, start_features=64, compute_dtype="bf16", device="cuda", lpips=True, seed=0):
    """Models of BASELINE config 2 (StyleGAN1, E_align_s2.py:27-46) with seeded random-init weights."""
    from .stylegan1 import Generator, Mapping
    from .encoder import BE
    from .lpips import LPIPS
    torch.manual_seed(seed)
    L = int(math.log2(img_size) - 1)
    Gs = Generator(startf=start_features, maxf=512, layer_count=L, latent_size=512, channels=3, compute_dtype=compute_dtype).to(device)
    Gm = Mapping(num_layers=2 * L, mapping_layers=8, latent_size=512, dlatent_size=512, mapping_fmaps=512).to(device)
    for p in list(Gs.parameters()) + list(Gm.parameters()):
        p.requires_grad_(False)
    with torch.no_grad():
        for name, p in Gs.named_parameters():
            if "noise_weight" in name:
                p.fill_(0.05)
    Gm.buffer1 = torch.randn(2 * L, 512) * 0.1
    E = BE(startf=start_features, maxf=512, layer_count=L, compute_dtype=compute_dtype).to(device)
    LP = LPIPS(compute_dtype=compute_dtype).to(device) if lpips else None
    return Gs, Gm, E, LP


def build_models_pg(img_size=256, start_features=64, compute_dtype="bf16", device="cuda", lpips=True, seed=0):
    """Models of BASELINE config 1 (PGGAN, E_align_s2.py:67-77) with seeded random-init weights."""
    from .pggan_generator import PGGANGenerator
    from .encoder_variants import PGBE
    from .lpips import LPIPS
    torch.manual_seed(seed)
    G = PGGANGenerator(resolution=img_size, compute_dtype=compute_dtype).to(device)
    for p in G.parameters():
        p.requires_grad_(False)
    E = PGBE(startf=start_features, maxf=512, layer_count=int(math.log2(img_size) - 1), pggan=True, compute_dtype=compute_dtype).to(device)
    LP = LPIPS(compute_dtype=compute_dtype).to(device) if lpips else None
    return G, E, LP


def build_models_big(config, img_size=256, start_features=64, compute_dtype="bf16", device="cuda", lpips=True, seed=0):
    """Models of BASELINE config 4 (BigGAN-deep, E_align_s2.py:79-86) with seeded random-init weights; `config`: BigGANConfig."""
    from .biggan_generator import BigGAN
    from .encoder_variants import BigBE
    from .lpips import LPIPS
    torch.manual_seed(seed)
    G = BigGAN(config, compute_dtype=compute_dtype).to(device)
    for p in G.parameters():
        p.requires_grad_(False)
    E = BigBE(startf=start_features, maxf=512, layer_count=int(math.log2(img_size) - 1), biggan=True, compute_dtype=compute_dtype).to(device)
    LP = LPIPS(compute_dtype=compute_dtype).to(device) if lpips else None
    return G, E, LP


def load_lpips_weights(LP, vgg_weights=None, lin_weights=None, allow_standin=False):
    """The `2*lpips` term of every image loss (training_utils.py:93) is only the reference's objective with the real
    LPIPS-VGG16 weights.  With the two files (see LPIPS.load_pretrained / INTEGRATION.md) they are loaded; without them
    training is refused unless `allow_standin` (benchmarks, smoke runs), and then says so loudly."""
    if LP is None:
        return None
    if vgg_weights:
        LP.load_pretrained(vgg_weights, lin_weights)
        return LP
    msg = ("LPIPS runs on SEEDED STAND-IN weights (no --vgg_weights / --lpips_weights given): the 2*lpips term of the image "
           "losses is a random-feature distance, NOT the reference's objective")
    if not allow_standin:
        raise RuntimeError(msg + "; pass --vgg_weights vgg16-397923af.pth --lpips_weights <lpips>/weights/v0.1/vgg.pth, "
                           "or --allow_standin_lpips for throughput / plumbing runs")
    import sys
    import warnings
    warnings.warn(msg)
    print("WARNING: " + msg, file=sys.stderr)
    return LP


def load_models(args, device="cuda", lpips=True):
    """Models + checkpoints of one --mtype, shared by `train` and the inference entry points (infer.main).  The three
    checkpoint containers of the reference: mtype 2 / 3 a dict holding `generator_smooth` (or `generator`)
    (E_align_s2.py:51-55, :67-77); mtype 1 a DIRECTORY with Gs_dict.pth, Gm_dict.pth and center_tensor.pt (:30-35);
    mtype 4 a bare state_dict next to --config_dir (:79-86); the encoder is a bare state_dict (--checkpoint_dir_E).
    Everything is read with map_location='cpu' and moved by load_state_dict.  Returns (G, Gm | None, E, LP | None)."""
    cd = getattr(args, "compute_dtype", "bf16")
    small = {k: getattr(args, k) for k in ("fmaps_base", "fmaps_max", "enc_maxf") if getattr(args, k, None) is not None}
    if args.mtype == 2:
        G, E, LP = build_models(args.img_size, args.start_features, cd, device=device, lpips=lpips, **small)
        Gm = None
        if args.checkpoint_dir_GAN:
            ckpt = torch.load(args.checkpoint_dir_GAN, map_location="cpu")
            G.load_state_dict(ckpt["generator_smooth"] if "generator_smooth" in ckpt else ckpt["generator"])
    elif args.mtype == 1:
        G, Gm, E, LP = build_models_sg1(args.img_size, args.start_features, cd, device=device, lpips=lpips)
        if args.checkpoint_dir_GAN:                 # E_align_s2.py:30-35: a directory holding the three files
            G.load_state_dict(torch.load(args.checkpoint_dir_GAN + "Gs_dict.pth", map_location="cpu"))
            Gm.load_state_dict(torch.load(args.checkpoint_dir_GAN + "Gm_dict.pth", map_location="cpu"))
            Gm.buffer1 = torch.load(args.checkpoint_dir_GAN + "./center_tensor.pt", map_location="cpu")
    elif args.mtype == 3:
        G, E, LP = build_models_pg(args.img_size, args.start_features, cd, device=device, lpips=lpips)
        Gm = None
        if args.checkpoint_dir_GAN:
            ckpt = torch.load(args.checkpoint_dir_GAN, map_location="cpu")
            G.load_state_dict(ckpt["generator_smooth"] if "generator_smooth" in ckpt else ckpt["generator"])
    elif args.mtype == 4:
        from .biggan_generator import BigGANConfig
        G, E, LP = build_models_big(BigGANConfig.from_json_file(args.config_dir), args.img_size, args.start_features, cd,
                                    device=device, lpips=lpips)
        Gm = None
        if args.checkpoint_dir_GAN:
            G.load_state_dict(torch.load(args.checkpoint_dir_GAN, map_location="cpu"))
    else:
        raise ValueError("--mtype must be 1 (StyleGAN1), 2 (StyleGAN2), 3 (PGGAN) or 4 (BigGAN)")
    if getattr(args, "checkpoint_dir_E", None) is not None:
        E.load_state_dict(torch.load(args.checkpoint_dir_E, map_location="cpu"))
    return G, Gm, E, LP


def add_model_args(parser):
    """The reference's model flags (E_align_s2.py:304-318), shared by the training and inference parsers."""
    parser.add_argument("--checkpoint_dir_GAN", default=None)
    parser.add_argument("--config_dir", default=None)
    parser.add_argument("--checkpoint_dir_E", default=None)
    parser.add_argument("--img_size", type=int, default=1024)
    parser.add_argument("--img_channels", type=int, default=3)
    parser.add_argument("--z_dim", type=int, default=512)
    parser.add_argument("--mtype", type=int, default=2)
    parser.add_argument("--start_features", type=int, default=16)
    parser.add_argument("--compute_dtype", default="bf16")
    # not in the reference: reduced StyleGAN2 / encoder widths (tests, smoke runs)
    parser.add_argument("--fmaps_base", type=int, default=None)
    parser.add_argument("--fmaps_max", type=int, default=None)
    parser.add_argument("--enc_maxf", type=int, default=None)
    return parser


def train(tensor_writer=None, args=None):
    """Reference E_align_s2.train() (flags: E_align_s2.py:304-318)."""
    if getattr(args, "deterministic", False):
        from . import ops
        ops.set_deterministic(True)
    G, Gm, E, LP = load_models(args)
    load_lpips_weights(LP, getattr(args, "vgg_weights", None), getattr(args, "lpips_weights", None),
                       allow_standin=getattr(args, "allow_standin_lpips", False))
    st = EAlignStep(G, E, LP, lr=args.lr, beta_1=args.beta_1, batch_size=args.batch_size, z_dim=args.z_dim, mapping=Gm,
                    stage=getattr(args, "stage", 2), zero_grad_to_none=not getattr(args, "legacy_zero_grad", False))
    # Launch mode.  At the reference's default batch (2, E_align_s2.py:308) the eager step is bound by the host's launch rate
    # (~560 launches): single-process runs at batch <= 2 therefore replay the iteration from a captured hipGraph by default
    # (EAlignStep.capture: z, style-mixing mask and Adam factors become device inputs; same numbers, see
    # tests/test_step_gpu.py::test_graph_replay_*).  --launch eager / graph overrides; --mtype 4 and the deterministic mode stay eager.
    mode = getattr(args, "launch", "auto")
    use_graph = mode == "graph" or (mode == "auto" and not st.dist_on and args.batch_size <= 2 and args.mtype != 4
                                    and not getattr(args, "deterministic", False))
    first = 0
    prefetch = not getattr(args, "no_prefetch", False)
    # capture() runs its warm-up iterations for real (2, one more in the legacy stage-1 form): a run shorter than that stays eager
    if use_graph and args.iterations <= 4:
        use_graph = False
    if use_graph:
        # iteration 0 runs eagerly: the reference logs its losses and dumps E_model_ep0_iter0.pth right after it
        # (E_align_s2.py: iteration % 100 == 0, iteration % 5000 == 0) - the dump holds the encoder after ONE iteration, as its name says
        r = st.step(0)
        print("ep_0_iter_0", "loss_tsa", float(r["loss_tsa"]), "loss_w", float(r["loss_w"]))
        if getattr(args, "experiment_dir", None):
            torch.save(E.state_dict(), "%s/E_model_ep0_iter0.pth" % args.experiment_dir)
        st.capture(start=1)
        first = st._g_iter           # capture() ran iterations 1 .. first - 1 for real (its warm-up): the loop continues behind them,
        # so that `--launch graph` and `--launch eager` make the same number of encoder updates on the same z / mask sequence
        print("ep_0_iter_1 .. %d ran inside the graph capture (warm-up)" % (first - 1))
    for iteration in range(first, args.iterations):
        # (eager launches: the next iteration's generator pass goes out beside this iteration's second backward, EAlignStep.step)
        r = st.replay(iteration) if use_graph else st.step(iteration, prefetch_next=(prefetch and iteration + 1 < args.iterations))
        if iteration % 100 == 0:
            print("ep_%d_iter_%d" % (iteration // 30000, iteration % 30000), "loss_tsa", float(r["loss_tsa"]),
                  "loss_w", float(r["loss_w"]))
        if iteration % 5000 == 0 and getattr(args, "experiment_dir", None):
            torch.save(E.state_dict(), "%s/E_model_ep%d_iter%d.pth" % (args.experiment_dir, iteration // 30000, iteration % 30000))
    return st


def main(argv=None):
    import argparse
    parser = argparse.ArgumentParser(description="the training args")
    parser.add_argument("--iterations", type=int, default=210000)
    parser.add_argument("--lr", type=float, default=0.0015)
    parser.add_argument("--beta_1", type=float, default=0.0)
    parser.add_argument("--batch_size", type=int, default=2)
    parser.add_argument("--experiment_dir", default=None)
    add_model_args(parser)
    parser.add_argument("--launch", choices=("auto", "eager", "graph"), default="auto",
                        help="auto: hipGraph replay of the iteration for single-process runs at batch <= 2, eager otherwise")
    parser.add_argument("--no_prefetch", action="store_true", help="eager launches: run the generator pass of iteration n + 1 at the start of that "
                        "iteration instead of beside the second backward of iteration n (same numbers either way)")
    parser.add_argument("--stage", type=int, default=2, help="2: E_align_s2.py; 1: E_align_cropping_s1.py (latent phase only trains E)")
    parser.add_argument("--legacy_zero_grad", action="store_true", help="stage 1: optimizer.zero_grad() as torch < 2.0 (zero-filled gradients, the "
                        "reference's pinned environment): the first optimizer step of an iteration ticks every Adam state")
    parser.add_argument("--vgg_weights", default=None, help="torchvision vgg16 checkpoint (features.*) or an lpips.LPIPS state_dict")
    parser.add_argument("--lpips_weights", default=None, help="the lpips package's weights/v0.1/vgg.pth (lin{k}.model.1.weight)")
    parser.add_argument("--deterministic", action="store_true", help="bit-reproducible reductions (training_utils.py:51 cudnn.deterministic): ops.set_deterministic")
    parser.add_argument("--allow_standin_lpips", action="store_true", help="train on seeded stand-in LPIPS weights (NOT the reference objective)")
    return train(None, parser.parse_args(argv))


if __name__ == "__main__":
    main()
