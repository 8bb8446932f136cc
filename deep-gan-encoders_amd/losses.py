"""space_loss (reference training_utils.py:54-99) and the three-scale image loss of
E_align_s2.py:185-203 on the HIP kernels, as autograd Functions whose forward also produces the
analytic gradient w.r.t. the second argument (the only one that carries grad in E_align)."""
import contextlib
import os

import torch

from . import ops
from ._lib import lib, check
from .ops import _f32, _p, _stream


class GlobalBatch:
    """Makes the batch-coupled terms of space_loss (cosine over the batch-flattened vector, means,
    the 1/N of mse / ssim / lpips) refer to the GLOBAL batch of a data-parallel run: the per-rank
    partial sums are all-reduced before the loss and its gradient are formed."""

    def __init__(self, world):
        self.world = world

    def reduce(self, t):
        from .e_align import _all_reduce
        _all_reduce(t)
        return t


def _pool_factor(h):
    k = 1
    while h > 256:          # training_utils.py:81 tests shape[2] only
        h //= 2
        k *= 2
    return k


def attention_windows(H, W):
    """(y0, x0, h, w) of the full image, AT1 and AT2 (E_align_s2.py:188-199)."""
    oy, ox = H // 8 + H // 32, W // 8 + W // 32
    return [(0, 0, H, W), (0, W // 8, H, W - 2 * (W // 8)), (oy, ox, H - 2 * oy, W - 2 * ox)]


_PK = 48          # floats per window in the packed reduction buffer: [0:8] the 8 sums, [8:40] SSIM slot copies, [40] LPIPS mean


def _window_reduce(a, b, win, image_space, lpips_model, need_grad, pk, world):
    """Pass 1 of one space_loss window: every per-rank partial sum goes into the packed row `pk` ([_PK] view, pre-zeroed), so
    that a data-parallel run exchanges ALL windows of a phase in one all-reduce.  Returns the state pass 2 needs."""
    B, Cc, H, W = a.shape
    y0, x0, h, w = win
    dev = a.device
    L = lib()
    slots = ops.zeros((16, 8), dev)         # 16 slot copies of the 8 sums (atomics contention), added up below
    check(L.dge_loss_reduce(_f32(a), _f32(b), _p(slots), B, Cc, H, W, y0, x0, h, w, _stream()), "dge_loss_reduce")
    ops._sum_over_batch(slots, pk[0:8])
    st = dict(win=win, pk=pk, k=1, npool=1.0, n=float(B * Cc * h * w) * world, lp=None, ap=None)
    if image_space:
        k = _pool_factor(h)
        hp, wp = h // k, w // k
        ap = torch.empty((B, Cc, hp, wp), dtype=torch.float32, device=dev)
        bp = torch.empty_like(ap)
        check(L.dge_crop_pool(_f32(a), _p(ap), B * Cc, H, W, y0, x0, h, w, k, _stream()), "dge_crop_pool")
        check(L.dge_crop_pool(_f32(b), _p(bp), B * Cc, H, W, y0, x0, h, w, k, _stream()), "dge_crop_pool")
        dmap = torch.empty((3, B, Cc, hp, wp), dtype=torch.float32, device=dev) if need_grad else None
        # pk[8:40]: 32 slot copies of the SSIM sum (atomics contention), added up by the finaliser
        check(L.dge_ssim_fwd(_p(ap), _p(bp), _p(pk[8:40]), _p(dmap), B * Cc, hp, wp, _stream()), "dge_ssim_fwd")
        st.update(k=k, npool=float(B * Cc * hp * wp) * world, ap=ap, bp=bp, dmap=dmap, g_lp=None)
        if lpips_model is not None:
            lp, st["g_lp"] = lpips_model.value_and_grad(ap, bp, need_grad=need_grad)       # mean over the rank's batch, d/dbp
            if world > 1:      # global mean = sum over ranks of (rank mean / world): joins the packed exchange
                check(L.dge_axpy_scalar(_p(lp), None, _p(pk[40:41]), 1, 1.0 / world, 0, _stream()), "dge_axpy_scalar")
                lp = pk[40:41]
            st["lp"] = lp
    return st


def _window_finish(a, b, st, image_space, weight, g_out, accumulate, world):
    """Pass 2: loss terms from the (globally reduced) sums and, when g_out is given, weight * dloss/db added into it."""
    B, Cc, H, W = a.shape
    y0, x0, h, w = st["win"]
    dev = a.device
    L = lib()
    pk = st["pk"]
    out8 = torch.empty(8, dtype=torch.float32, device=dev)
    gp = None
    if image_space and g_out is not None:
        bp = st["bp"]
        gp = torch.empty_like(bp)
        check(L.dge_ssim_bwd(_p(st["ap"]), _p(bp), _p(st["dmap"]), _p(gp), B * Cc, bp.shape[2], bp.shape[3], -1.0 / st["npool"], 0,
                             _stream()), "dge_ssim_bwd")
        if st["g_lp"] is not None:
            check(L.dge_axpy_scalar(_p(st["g_lp"]), None, _p(gp), gp.numel(), 2.0 / world, 1, _stream()), "dge_axpy_scalar")
    check(L.dge_space_loss_finalize(_p(pk[0:8]), _p(pk[8:40]) if image_space else None, _p(st["lp"]), _p(out8), st["n"], st["npool"],
                                    1 if image_space else 0, _stream()), "dge_space_loss_finalize")
    if g_out is not None:
        check(L.dge_space_loss_bwd(_f32(a), _f32(b), _p(pk[0:8]), _p(gp), _p(g_out), B * Cc, H, W, y0, x0, h, w, st["k"], st["n"],
                                   float(weight), 1 if accumulate else 0, _stream()), "dge_space_loss_bwd")
    return out8


def _space_loss_windows(a, b, wins, image_space, lpips_model, weights, g_outs, accumulate, gb=None):
    """space_loss on several windows of a, b [B,C,H,W] (training_utils.py:54-99 each): all reductions first, ONE exchange of the
    packed partial sums in a data-parallel run (`gb`), then the loss terms and gradients.  Returns the list of out8 tensors."""
    world = gb.world if gb is not None else 1
    pack = ops.zeros((len(wins), _PK), a.device)
    sts = [_window_reduce(a, b, win, image_space, lpips_model, g_outs[i] is not None, pack[i], world) for i, win in enumerate(wins)]
    if gb is not None:
        gb.reduce(pack)
    return [_window_finish(a, b, st, image_space, weights[i], g_outs[i], accumulate, world) for i, st in enumerate(sts)]


# DGE_SIDE_STREAMS=0: everything on the caller's stream (PMC counter collection serialises kernels; profiles of single stages)
_WINDOW_STREAMS = os.environ.get("DGE_WINDOW_STREAMS", "1") != "0" and os.environ.get("DGE_SIDE_STREAMS", "1") != "0"
_SIDE = {}


def _side_streams(dev, n):
    """n side streams of the device, created once (stream creation inside a step would be a host synchronisation)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    have = _SIDE.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=dev))
    return have[:n]


def _space_loss_windows3(a, b, wins, lpips_model, weights, g, need, gb=None):
    """The three nested attention windows of image_loss_tsa with every image pass merged (dge_loss_reduce3, dge_crop_pool_multi,
    dge_space_loss_bwd3): `g` (or None) is WRITTEN with the weighted sum of the windows' gradients; need[i] False leaves window i
    out of the gradient.  Same arithmetic as _window_reduce / _window_finish per window."""
    import ctypes as C
    B, Cc, H, W = a.shape
    dev = a.device
    L = lib()
    world = gb.world if gb is not None else 1
    nw = len(wins)
    pack = ops.zeros((nw, _PK), dev)
    slots = ops.zeros((nw, 16, 8), dev)
    wflat = (C.c_int * (4 * nw))(*[int(v) for win in wins for v in win])
    check(L.dge_loss_reduce3(_f32(a), _f32(b), _p(slots), B, Cc, H, W, wflat, nw, _stream()), "dge_loss_reduce3")
    sums = ops.DeferredSums()
    for i in range(nw):
        sums.add(slots[i].view(16, 8, 1), pack[i, 0:8])      # [nslot, C = 8, NS = 1] -> the 8 sums of window i
    sums.flush()
    ks = [_pool_factor(win[2]) for win in wins]
    aps = [torch.empty((B, Cc, win[2] // k, win[3] // k), dtype=torch.float32, device=dev) for win, k in zip(wins, ks)]
    bps = [torch.empty_like(t) for t in aps]
    srcs = (C.c_void_p * (2 * nw))(*([a.data_ptr()] * nw + [b.data_ptr()] * nw))
    dsts = (C.c_void_p * (2 * nw))(*([t.data_ptr() for t in aps] + [t.data_ptr() for t in bps]))
    w2 = (C.c_int * (8 * nw))(*([int(v) for win in wins for v in win] * 2))
    k2 = (C.c_int * (2 * nw))(*(ks * 2))
    check(L.dge_crop_pool_multi(srcs, dsts, w2, k2, 2 * nw, B * Cc, H, W, _stream()), "dge_crop_pool_multi")
    sts = []
    # The windows are independent from here to the join below, and their LPIPS launches are small (conv3 - conv5 on 16^2 .. 64^2
    # features: 256 - 1152 workgroups, one or two per CU, each a serial weight stream): on one stream they run one after the
    # other on a half-empty chip.  Each window gets its own stream (forked off the caller's, joined before the results are
    # read); the first call stays on one stream (it fills the LPIPS weight-pack cache), so does the deterministic mode (its
    # slot workspace belongs to one stream at a time).
    main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
    # (the stream switches cost ~1 ms of host time per step: taken where the GPU work hides them - >= 4 Mpixel per call - or
    #  where the host is out of the picture, i.e. while a hipGraph is being captured)
    fork = (main is not None and lpips_model is not None and nw > 1 and _WINDOW_STREAMS and not ops.is_deterministic()
            and getattr(lpips_model, "_streams_warm", False)
            and (B * H * W >= (4 << 20) or torch.cuda.is_current_stream_capturing()))
    side = _side_streams(dev, nw - 1) if fork else []
    for i, win in enumerate(wins):
        y0, x0, h, w = win
        ap, bp, k = aps[i], bps[i], ks[i]
        hp, wp = h // k, w // k
        ng = g is not None and need[i]
        strm = side[i - 1] if (fork and i > 0) else None
        if strm is not None:
            strm.wait_stream(main)
        with (torch.cuda.stream(strm) if strm is not None else contextlib.nullcontext()):
            dmap = torch.empty((3, B, Cc, hp, wp), dtype=torch.float32, device=dev) if ng else None
            check(L.dge_ssim_fwd(_p(ap), _p(bp), _p(pack[i, 8:40]), _p(dmap), B * Cc, hp, wp, _stream()), "dge_ssim_fwd")
            st = dict(win=win, pk=pack[i], k=k, npool=float(B * Cc * hp * wp) * world, n=float(B * Cc * h * w) * world, ap=ap, bp=bp,
                      dmap=dmap, g_lp=None, lp=None, ng=ng)
            if lpips_model is not None:
                lp, st["g_lp"] = lpips_model.value_and_grad(ap, bp, need_grad=ng)
                if world > 1:
                    check(L.dge_axpy_scalar(_p(lp), None, _p(pack[i, 40:41]), 1, 1.0 / world, 0, _stream()), "dge_axpy_scalar")
                    lp = pack[i, 40:41]
                st["lp"] = lp
            # the window's pooled-image gradient (SSIM + LPIPS; neither needs the global sums) in the window's stream too
            st["gp"] = None
            if ng:
                gp = torch.empty_like(bp)
                check(L.dge_ssim_bwd(_p(ap), _p(bp), _p(dmap), _p(gp), B * Cc, hp, wp, -1.0 / st["npool"], 0, _stream()), "dge_ssim_bwd")
                if st["g_lp"] is not None:
                    check(L.dge_axpy_scalar(_p(st["g_lp"]), None, _p(gp), gp.numel(), 2.0 / world, 1, _stream()), "dge_axpy_scalar")
                st["gp"] = gp
        if strm is not None:            # results allocated on the side stream are read (and freed) under the caller's stream
            for t in (dmap, st["g_lp"], st["lp"], st["gp"]):
                if t is not None:
                    t.record_stream(main)
        sts.append(st)
    for strm in side:
        main.wait_stream(strm)
    if lpips_model is not None:
        lpips_model._streams_warm = True
    if gb is not None:
        gb.reduce(pack)
    outs, gps = [], []
    for i, st in enumerate(sts):
        gps.append(st["gp"])
        out8 = torch.empty(8, dtype=torch.float32, device=dev)
        check(L.dge_space_loss_finalize(_p(pack[i, 0:8]), _p(pack[i, 8:40]), _p(st["lp"]), _p(out8), st["n"], st["npool"], 1, _stream()),
              "dge_space_loss_finalize")
        outs.append(out8)
    if g is not None:
        sp = (C.c_void_p * nw)(*[pack[i, 0:8].data_ptr() for i in range(nw)])
        gpp = (C.c_void_p * nw)(*[(t.data_ptr() if t is not None else None) for t in gps])
        nn = (C.c_float * nw)(*[st["n"] for st in sts])
        ww = (C.c_float * nw)(*[float(weights[i]) if sts[i]["ng"] else 0.0 for i in range(nw)])
        kk = (C.c_int * nw)(*ks)
        check(L.dge_space_loss_bwd3(_f32(a), _f32(b), sp, gpp, _p(g), B * Cc, H, W, wflat, kk, nn, ww, nw, _stream()), "dge_space_loss_bwd3")
    return outs


def _space_loss_window(a, b, win, image_space, lpips_model, weight, g_out, accumulate, gb=None):
    return _space_loss_windows(a, b, [win], image_space, lpips_model, [weight], [g_out], accumulate, gb)[0]


class _ScaledGrad(torch.autograd.Function):
    """loss tensor whose gradient w.r.t. `b` was computed analytically in the forward."""

    @staticmethod
    def forward(ctx, b, loss, g):
        ctx.save_for_backward(g)
        return loss.clone()

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        out = torch.empty_like(g)
        check(lib().dge_axpy_scalar(_p(g), _p(go.contiguous().float()), _p(out), g.numel(), 1.0, 0, _stream()), "dge_axpy_scalar")
        return out, None, None


class _ScaledGrad2(torch.autograd.Function):
    """as _ScaledGrad, for a loss whose BOTH arguments carry a gradient (latent losses of embedding_img.py:118-124)."""

    @staticmethod
    def forward(ctx, a, b, loss, ga, gb):
        ctx.save_for_backward(ga, gb)
        return loss.clone()

    @staticmethod
    def backward(ctx, go):
        outs = []
        gof = go.contiguous().float()
        for g in ctx.saved_tensors:
            out = torch.empty_like(g)
            check(lib().dge_axpy_scalar(_p(g), _p(gof), _p(out), g.numel(), 1.0, 0, _stream()), "dge_axpy_scalar")
            outs.append(out)
        return outs[0], outs[1], None, None, None


def image_loss_tsa(imgs1, imgs2, lpips_model=None, weights=(1.0, 5.0, 9.0), global_batch=None, grad_windows=(True, True, True)):
    """loss_tsa = loss_imgs + 5*loss_medium + 9*loss_small (E_align_s2.py:185-203).
    Returns (loss [] on device, info [3,8] on device: rows full/AT1/AT2, columns
    loss, mse, mse_mean, mse_std, kl, cos, ssim, lpips).  No host synchronisation."""
    a = imgs1.detach().float().contiguous()
    b = imgs2.detach().float().contiguous()
    need = imgs2.requires_grad and torch.is_grad_enabled()
    wins = attention_windows(a.shape[2], a.shape[3])
    # grad_windows[i] False: the window enters the loss VALUE only (embedding_img.py:95-107 detaches both crops)
    if not ops.is_deterministic():       # every pass over the two images merged over the three windows
        g = torch.empty_like(b) if need else None
        infos = _space_loss_windows3(a, b, wins, lpips_model, weights, g, [bool(need and grad_windows[i]) for i in range(3)], gb=global_batch)
    else:
        g = torch.zeros_like(b) if need else None
        infos = _space_loss_windows(a, b, wins, True, lpips_model, weights,
                                    [g if (need and grad_windows[i]) else None for i in range(3)], accumulate=True, gb=global_batch)
    info = torch.stack(infos)
    loss = info[0, 0] * float(weights[0]) + info[1, 0] * float(weights[1]) + info[2, 0] * float(weights[2])   # no host->device copy
    if need:
        loss = _ScaledGrad.apply(imgs2, loss, g)
    return loss, info


def space_loss(imgs1, imgs2, image_space=True, lpips_model=None, global_batch=None):
    """Drop-in for training_utils.space_loss; returns (loss tensor, info tensor[8] on device)
    instead of Python floats (the reference's 7 .item() syncs per call are deferred)."""
    a = imgs1.detach().float().contiguous()
    b = imgs2.detach().float().contiguous()
    need = imgs2.requires_grad and torch.is_grad_enabled()
    g = torch.empty_like(b) if need else None
    if image_space:
        B, Cc, H, W = a.shape
        out8 = _space_loss_window(a, b, (0, 0, H, W), True, lpips_model, 1.0, g, accumulate=False, gb=global_batch)
    else:
        # 3-D latents: the implicit softmax dim is 0 (the batch) -> planes = batch (training_utils.py:67)
        Bt = a.shape[0]
        n_in = a.numel() // Bt
        if a.dim() == 2:
            # 2-D latents (PGGAN / BigGAN z): the implicit softmax dim is 1 -> one softmax per sample over the features
            out8 = _space_loss_window(a.view(Bt, n_in, 1, 1), b.view(Bt, n_in, 1, 1), (0, 0, 1, 1), False, None, 1.0,
                                      g.view(Bt, n_in, 1, 1) if need else None, accumulate=False, gb=global_batch)
            loss = out8[0]
            return (_ScaledGrad.apply(imgs2, loss, g) if need else loss), out8
        out8 = _space_loss_window(a.view(1, Bt, 1, n_in), b.view(1, Bt, 1, n_in), (0, 0, 1, n_in), False, None, 1.0,
                                  g.view(1, Bt, 1, n_in) if need else None, accumulate=False, gb=global_batch)
        if imgs1.requires_grad and torch.is_grad_enabled():
            # the first argument carries a gradient too: 5*mse + 3*cos is symmetric, so d/da is the same kernel with
            # the arguments exchanged (the logged-only KL term is not, and is taken from the first evaluation)
            ga = torch.empty_like(a)
            _space_loss_window(b.view(1, Bt, 1, n_in), a.view(1, Bt, 1, n_in), (0, 0, 1, n_in), False, None, 1.0,
                               ga.view(1, Bt, 1, n_in), accumulate=False, gb=global_batch)
            gbt = g if need else torch.zeros_like(b)
            return _ScaledGrad2.apply(imgs1, imgs2, out8[0], ga, gbt), out8
    loss = out8[0]
    if need:
        loss = _ScaledGrad.apply(imgs2, loss, g)
    return loss, out8
