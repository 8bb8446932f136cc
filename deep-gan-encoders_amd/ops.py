"""Tensor-level wrappers over the C ABI (include/dge_hip.h).  PyTorch supplies device memory
and the stream; every computation happens in libdge_hip.so."""
import ctypes as C
import math

import torch

from ._lib import lib, check, ConvDesc, ConvPPDesc, DgeError, last_kernel

F32, BF16 = 0, 1
ACT_NONE, ACT_LRELU, ACT_RELU, LIN_RSQRT = 0, 1, 2, 3
PACK_FWD, PACK_UPFOLD, PACK_DGRAD, PACK_UPFOLD_DGRAD, PACK_SG1_UP, PACK_SG1_UP_DGRAD, PACK_UPT2D_DGRAD = 0, 1, 2, 3, 4, 5, 6
PACK_FRAG = 0x100     # OR-ed into a pack mode: MFMA-fragment order for the low-resolution kernel (csrc/conv_small.hip)
import os as _os
import weakref as _weakref
_PF = {"on": not _os.environ.get("DGE_NO_PREFETCH"), "prev": {}, "next": {}}     # low-resolution weight prefetch chain (conv2d)
KERNEL_LOG = None   # tests set this to a list: (kernel instantiation name, stream handle) per conv_pp / up_pp launch (which streams ran them)
PROFILE = None      # bench.py sets this to a list: (start_event, stop_event, algorithmic_flops, tag, algorithmic_bytes) per conv launch


class _ZeroArena:
    """Pre-zeroed float32 scratch for the many small accumulation targets of one training step
    (statistics slots, reduction outputs, weight-gradient buffers).  One memset per step replaces
    a fill launch per buffer.  Tensors handed out are views: they stay valid until the arena is
    re-armed by the next `zero_arena_begin()`."""

    def __init__(self):
        self.buf, self.off, self.high, self.active = None, 0, 0, False


_ARENA = _ZeroArena()


def zero_arena_begin(device, nbytes=512 << 20):
    a = _ARENA
    if a.buf is None or a.buf.device != torch.device(device) or a.buf.numel() * 4 < nbytes:
        a.buf = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
    else:
        a.buf[:a.high].zero_()
    a.off, a.high, a.active = 0, 0, True


def zero_arena_end():
    _ARENA.active = False


def zeros(shape, device):
    """float32 zeros; served from the step arena when one is armed, else a plain allocation."""
    a = _ARENA
    n = int(math.prod(shape))
    if a.active and a.buf.device == torch.device(device) and a.off + n <= a.buf.numel():
        t = a.buf[a.off:a.off + n].view(shape)
        a.off = (a.off + n + 63) & ~63
        a.high = max(a.high, a.off)
        return t
    return torch.zeros(shape, dtype=torch.float32, device=device)


class SlotStats:
    """Per-(b,c) statistics target [B,C,2] whose slot copies (dge_conv2d spreads its statistics atomics over up to 64 copies,
    same-address contention) are added by the CONSUMING kernel (stats_finalize / in_bwd_coef) instead of a dge_sum_slots launch."""
    __slots__ = ("B", "C", "device", "buf", "nslot")

    def __init__(self, B, C, device):
        self.B, self.C, self.device, self.buf, self.nslot = B, C, device, None, 1

    def alloc(self, nslot):
        self.nslot = nslot
        self.buf = zeros((nslot, self.B, self.C, 2), self.device)
        return self.buf

    def plain(self):
        """a [B,C,2] tensor for producers that do not use slots (fromrgb, blend)"""
        return self.alloc(1)[0]


def _sum_over_batch(partial, out=None):
    """partial [B, ...] per-sample sums -> [...] (dge_sum_slots); accumulates into `out` when given."""
    n = partial[0].numel()
    acc = out is not None
    if out is None:
        out = torch.empty(partial.shape[1:], dtype=torch.float32, device=partial.device)
    check(lib().dge_sum_slots(_p(partial), _p(out), partial.shape[0], n, 1 if acc else 0, _stream()), "dge_sum_slots")
    return out


# ------------------------------------------------------------------ counter-based noise (dge_randn)
class _Noise:
    """State of the step's noise draws: `seed` (set by e_align.set_seed next to torch / numpy), a draw counter that names the
    Philox subsequence of each tensor, and the data-parallel position (rank, world): a draw of per-sample rows [B, ...] is the
    rank's slice of the global-batch tensor [world*B, ...], so N ranks x B reproduce one process at batch N*B exactly."""

    def __init__(self):
        self.seed, self.counter, self.rank, self.world, self.seed_dev = None, 0, 0, 1, None


NOISE = _Noise()


def noise_seed(seed):
    NOISE.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    NOISE.counter = 0
    if NOISE.seed_dev is not None:
        # a fresh pageable host tensor per call: staged before copy_ returns, so the host may run replays ahead
        NOISE.seed_dev.copy_(torch.tensor([NOISE.seed - (1 << 64) if NOISE.seed >= (1 << 63) else NOISE.seed], dtype=torch.int64))


def noise_dp(rank, world):
    NOISE.rank, NOISE.world = int(rank), int(world)


def noise_graph_begin(device):
    """hipGraph mode: from now on the kernels read the seed from a device scalar that noise_seed() refreshes before each replay."""
    NOISE.seed_dev = torch.zeros(1, dtype=torch.int64, device=device)
    if NOISE.seed is not None:
        noise_seed(NOISE.seed)


def randn_rows(shapes, device):
    """Normal noise tensors of per-sample shapes [(B, ...), ...] in ONE launch (contiguous slices of one buffer).  Tensor k
    is draw number `counter + k`; its elements are those of rows [rank*B, (rank+1)*B) of the global [world*B, ...] tensor."""
    if NOISE.seed is None:
        noise_seed(torch.initial_seed())
    n = len(shapes)
    sizes = [int(math.prod(sh)) for sh in shapes]
    starts, off = [], 0
    for sz in sizes:
        starts.append(off)
        off += (sz + 3) & ~3                       # keep every tensor 16-byte aligned
    flat = torch.empty(max(off, 1), dtype=torch.float32, device=device)
    LL, ULL, UI = C.c_longlong * n, C.c_ulonglong * n, C.c_uint * n
    goff = [NOISE.rank * sz for sz in sizes]
    sub = [(NOISE.counter + k) & 0xFFFFFFFF for k in range(n)]
    NOISE.counter += n
    sd = C.c_void_p(NOISE.seed_dev.data_ptr()) if NOISE.seed_dev is not None else None
    check(lib().dge_randn(_p(flat), n, LL(*starts), LL(*sizes), ULL(*goff), UI(*sub), C.c_ulonglong(NOISE.seed), sd, _stream()),
          "dge_randn")
    return [flat[st:st + sz].view(sh) for st, sz, sh in zip(starts, sizes, shapes)]


def randn(shape, device):
    return randn_rows([tuple(shape)], device)[0]


def tdtype(dtype):
    return torch.bfloat16 if dtype == BF16 else torch.float32


def dtype_of(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float32:
        return F32
    raise DgeError(f"unsupported activation dtype {t.dtype}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current stream of the current device as a hipStream_t.  torch.cuda.current_stream() builds a Stream object through
    several Python layers (~4 us); with ~850 launches per step that was a fifth of the host time at batch 2, so the raw
    handle is fetched directly when this torch exposes it."""
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise DgeError("tensor is not on the GPU: the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise DgeError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def _f32(t):
    if t is not None and t.dtype != torch.float32:
        raise DgeError(f"expected float32, got {t.dtype}")
    return _p(t)


def packed_n(n):
    return lib().dge_packed_n(int(n))


def small_conv(H, W, kdim, nvalid, ksize=3, in_s2d=False, in_up2=False, dtype=BF16):
    """True when dge_conv2d runs a launch of this shape (input grid H x W, packed K = kdim, N = nvalid before padding) on the
    low-resolution kernel: its weights must then be packed with `mode | PACK_FRAG`."""
    return bool(lib().dge_conv_small_supported(int(H), int(W), int(kdim), packed_n(nvalid), int(ksize), int(bool(in_s2d)),
                                               int(bool(in_up2)), int(dtype)))


def pack_dims(w, mode):
    """(nvalid, kdim) of the packed copy of w [Cout,Cin,k,k] in pack mode `mode`"""
    cout, cin = w.shape[0], w.shape[1]
    mode &= 0xff
    if mode in (PACK_SG1_UP, PACK_SG1_UP_DGRAD):
        cin, cout = cout, cin
    nvalid = 4 * cout if mode in (PACK_UPFOLD, PACK_SG1_UP) else (cin if mode in (PACK_DGRAD, PACK_UPFOLD_DGRAD, PACK_SG1_UP_DGRAD, PACK_UPT2D_DGRAD) else cout)
    kdim = cout if mode == PACK_DGRAD else (4 * cout if mode in (PACK_UPFOLD_DGRAD, PACK_SG1_UP_DGRAD, PACK_UPT2D_DGRAD) else cin)
    return nvalid, kdim


def pack_mode_for(w, mode, H, W, dtype):
    """`mode`, with PACK_FRAG added when the conv that reads this copy at input resolution H x W runs on the low-resolution kernel"""
    nvalid, kdim = pack_dims(w, mode)
    in_s2d = (mode & 0xff) in (PACK_UPFOLD_DGRAD, PACK_SG1_UP_DGRAD)
    return mode | PACK_FRAG if small_conv(H, W, kdim, nvalid, w.shape[-1], in_s2d, False, dtype) else mode


def pack_conv_weight(w, mode=PACK_FWD, dtype=BF16, scale=1.0):
    """w: [Cout,Cin,k,k] f32 (reference layout) -> packed [k*k, Npad, K] tensor of `dtype`; with `mode | PACK_FRAG` the same values in
    MFMA-fragment order (the tensor carries `_dge_frag = True`, which conv2d hands on as dge_conv_desc.w_layout)."""
    frag = bool(mode & PACK_FRAG)
    mode_full, mode = mode, mode & 0xff
    cout, cin, k, _ = w.shape
    if mode in (PACK_SG1_UP, PACK_SG1_UP_DGRAD):          # ConvTranspose2d parameter layout [Cin, Cout, k, k]
        cin, cout = cout, cin
    nvalid = 4 * cout if mode in (PACK_UPFOLD, PACK_SG1_UP) else (cin if mode in (PACK_DGRAD, PACK_UPFOLD_DGRAD, PACK_SG1_UP_DGRAD, PACK_UPT2D_DGRAD) else cout)
    kdim = cout if mode == PACK_DGRAD else (4 * cout if mode in (PACK_UPFOLD_DGRAD, PACK_SG1_UP_DGRAD, PACK_UPT2D_DGRAD) else cin)
    out = torch.empty((k * k, packed_n(nvalid), kdim), dtype=tdtype(dtype), device=w.device)
    check(lib().dge_pack_conv_weight(_f32(w.detach().contiguous()), _p(out), cout, cin, k, mode_full, dtype, float(scale),
                                     _stream()), "dge_pack_conv_weight")
    if frag:
        out._dge_frag = True
    return out


def set_deterministic(on=True):
    """Deterministic mode of the library (the reference pins torch.backends.cudnn.deterministic = True, training_utils.py:51):
    every reduction that ends in same-address f32 atomics - conv statistics, weight-gradient flush, the per-channel sums of
    the streaming backward kernels, loss sums - goes through per-contributor slots and an ordered sum by the last contributor
    instead (csrc/common.h).  Run to run the results are then bit-identical.  Synchronises the device."""
    check(lib().dge_set_deterministic(1 if on else 0), "dge_set_deterministic")


def is_deterministic():
    return bool(lib().dge_get_deterministic())


def pack_conv_weights_multi(entries, scratch=None):
    """entries: [(w [Cout,Cin,k,k] f32, mode, dtype, scale, out)] with `out` the tensors pack_conv_weight returned for the same
    arguments: refreshes all of them in ONE launch.  Returns the device scratch (pass it back to reuse it)."""
    import struct
    n = len(entries)
    if n == 0:
        return scratch
    rows = []
    for (w, mode, dtype, scale, out) in entries:
        cout, cin, k, _ = w.shape
        if (mode & 0xff) in (PACK_SG1_UP, PACK_SG1_UP_DGRAD):
            cin, cout = cout, cin
        if not (w.is_cuda and w.is_contiguous() and w.dtype == torch.float32):
            raise DgeError("pack_conv_weights_multi: weights must be contiguous f32 device tensors")
        rows += [w.data_ptr(), out.data_ptr(), cout | (cin << 32), k | (mode << 32), dtype,
                 struct.unpack("<I", struct.pack("<f", float(scale)))[0], 0, 0]
    tab = (C.c_longlong * (8 * n))(*rows)
    need = n * lib().dge_pack_desc_bytes()
    key = tuple(rows)
    if scratch is None or scratch[0].numel() < need:
        scratch = [torch.empty(max(need, 4096), dtype=torch.uint8, device=entries[0][0].device), None]
    upload = 0 if scratch[1] == key else 1          # same table as the last call: the descriptors are already on the device
    check(lib().dge_pack_conv_weights_multi(tab, _p(scratch[0]), n, upload, _stream()), "dge_pack_conv_weights_multi")
    scratch[1] = key
    return scratch


def weight_sumsq(w, scale=1.0):
    cout, cin, k, _ = w.shape
    out = torch.empty((cout, cin), dtype=torch.float32, device=w.device)
    check(lib().dge_weight_sumsq(_f32(w.detach().contiguous()), _p(out), cout, cin, k, float(scale), _stream()),
          "dge_weight_sumsq")
    return out


def linear(x, w, bias=None, wscale=1.0, bscale=1.0, add=0.0, act=ACT_NONE, gain=1.0, square_input=False, out=None):
    """x: [B, I] f32 view with unit inner stride (row stride arbitrary); w: [O, I]."""
    B, I = x.shape
    O = w.shape[0]
    if x.stride(1) != 1:
        raise DgeError("linear: inner stride must be 1")
    if out is None:
        out = torch.empty((B, O), dtype=torch.float32, device=x.device)
    if not x.is_cuda:
        raise DgeError("tensor is not on the GPU: the HIP path has no CPU fallback")
    check(lib().dge_linear(C.c_void_p(x.data_ptr()), x.stride(0), _f32(w), _f32(bias), C.c_void_p(out.data_ptr()),
                           out.stride(0), B, I, O, float(wscale), float(bscale), float(add), act, float(gain),
                           1 if square_input else 0, _stream()), "dge_linear")
    return out


def dense_chain(x, layers, pixelnorm=False, eps=1e-8):
    """x [B, I] f32 through a chain of up to 8 dense layers in one launch (dge_dense_chain; bit-identical to the per-layer linear()
    calls).  layers: objects with weight [O, I], bias, wscale, bscale, additional_bias, act, gain (DenseBlock)."""
    from ._lib import DenseLayer
    B = x.shape[0]
    arr = (DenseLayer * len(layers))()
    for e, L in zip(arr, layers):
        e.w, e.bias = _p(L.weight.detach()), _p(L.bias.detach())
        e.O, e.I = L.weight.shape
        e.wscale, e.bscale, e.add, e.act, e.gain = float(L.wscale), float(L.bscale), float(L.additional_bias), int(L.act), float(L.gain)
    y = torch.empty((B, layers[-1].weight.shape[0]), dtype=torch.float32, device=x.device)
    if not x.is_cuda or x.stride(1) != 1:
        raise DgeError("dense_chain: x must be a CUDA tensor with unit inner stride")
    check(lib().dge_dense_chain(_f32(x), x.stride(0), arr, len(layers), _p(y), y.stride(0), B, 1 if pixelnorm else 0, float(eps), _stream()),
          "dge_dense_chain")
    return y


def pixelnorm(x, eps=1e-8):
    y = torch.empty_like(x)
    check(lib().dge_pixelnorm(_f32(x), _p(y), x.shape[0], x.shape[1], eps, _stream()), "dge_pixelnorm")
    return y


def truncation(w, w_avg, num_layers, psi, layers):
    w_is_wp = (w.ndim == 3)
    B, D = w.shape[0], w.shape[-1]
    wp = torch.empty((B, num_layers, D), dtype=torch.float32, device=w.device)
    check(lib().dge_truncation(_f32(w.contiguous()), _f32(w_avg), _p(wp), B, num_layers, D, float(psi), int(layers),
                               1 if w_is_wp else 0, _stream()), "dge_truncation")
    return wp


def conv2d(x, w_packed, cout, ksize=3, up=False, in_scale=None, in_shift=None, out_scale=None, bias=None,
           bias_scale=1.0, noise=None, noise_w=None, act=ACT_NONE, gain=1.0, addend=None, add_scale=1.0, stats=None,
           out=None, in_s2d=False, dot_src=None, in_up2=False, in_relu=False, prep=None, relu_mask=None, in_t2d=False, rgb=None, pool_out=False, pool_mask=False,
           in_bwd=None):
    """x: [B,H,W,Cin] NHWC (bf16 or f32).  Returns y [B,OH,OW,cout].
    `prep`: dict(gain, noise [1|B,OH,OW] or None, ns (device scalar) or None, stats=SlotStats(B, cout)) - the fused tail backward of
    the layer that produced `dot_src` (dge_conv_desc.prep): y is then g_z and prep['stats'] receives (sum g_z*(z - ns*noise), sum g_z).
    `relu_mask`: stored activation a = relu(pre) of the layer below: the result is multiplied by [a > 0] (dge_conv_desc.mask_relu).
    `rgb`: dict(w [3,cout] f32, style [B,cout], bias [3], wscale, out [B,3,H,W] f32, skip_y=False) - the toRGB of the result written
    by the same launch (dge_conv_desc.rgb_*, where conv_rgb_supported() says so); with skip_y the activation itself is not stored
    and None is returned.
    `in_bwd`: dict(coef [B,cout,3], noise [B,H,W] or None, red=SlotStats(B, cout)) - the instance-norm + activation backward of the
    layer input dot_src in the epilogue (dge_conv_desc.in_bwd_coef, where conv_in_bwd_supported() says so): y is g_pre, red receives
    (sum g_pre, sum g_pre*noise).
    `pool_out`: the launch stores the 2x2 average pool of its result, [B,OH/2,OW/2,cout] (dge_conv_desc.pool_out, where
    conv_pool_supported() says so); with pool_mask the signs of the full-resolution values come back too: returns (y, mask)."""
    B, H, W, Cin = x.shape
    if in_s2d:            # x is the fine grid [B,2H,2W,C]; logical input is [B,H,W,4C]
        H, W, Cin = H // 2, W // 2, Cin * 4
    if in_up2:            # x is the coarse grid; the conv runs on its nearest x2 upsample
        H, W = 2 * H, 2 * W
    if in_t2d:            # x is fir_t2d output [B,H+1,W+1,4C]; the conv runs on the H x W grid (phase-form adjoint of the up layer)
        H, W = H - 1, W - 1
    dt = dtype_of(x)
    OH, OW = (2 * H, 2 * W) if up else (H, W)
    mask = None
    if pool_out:
        if out is not None or up:
            raise DgeError("conv2d: pool_out allocates its own result and excludes up")
        out = torch.empty((B, OH // 2, OW // 2, cout), dtype=x.dtype, device=x.device)
        if pool_mask:
            mask = torch.empty((B, (OH // 2) * (OW // 2), cout // 8), dtype=torch.int32, device=x.device)
    if out is None:
        out = torch.empty((B, OH, OW, cout), dtype=x.dtype, device=x.device)
    d = ConvDesc()
    d.pool_out, d.pool_mask = 1 if pool_out else 0, _p(mask)
    if relu_mask is not None:
        if dot_src is not None or prep is not None:
            raise DgeError("conv2d: relu_mask excludes dot_src / prep")
        dot_src = relu_mask
    d.mask_relu = 0 if relu_mask is None else 1
    d.in_t2d = 1 if in_t2d else 0
    d.x, d.w_packed, d.y, d.addend, d.dot_src = _p(x), _p(w_packed), _p(out), _p(addend), _p(dot_src)
    d.in_scale, d.in_shift, d.out_scale = _f32(in_scale), _f32(in_shift), _f32(out_scale)
    # statistics atomics of large grids are spread over several copies (same-address contention), then combined
    partial, nslot = None, 1
    lazy = isinstance(stats, SlotStats)
    if stats is not None:
        nblk = ((H + 15) // 16) * ((W + 15) // 16) * B
        nslot = max(1, min(64, nblk // 16))
        if lazy:
            partial = stats.alloc(nslot)
        elif nslot > 1:
            partial = zeros((nslot,) + tuple(stats.shape), x.device)
    d.bias, d.noise, d.noise_w, d.stats = _f32(bias), _f32(noise), _f32(noise_w), _f32(partial if partial is not None else stats)
    d.stats_slots = nslot
    d.B, d.H, d.W, d.Cin, d.Cout = B, H, W, Cin, cout
    d.ksize, d.up, d.in_s2d, d.in_up2 = ksize, 1 if up else 0, 1 if in_s2d else 0, 1 if in_up2 else 0
    d.in_relu = 1 if in_relu else 0
    d.w_layout = 1 if getattr(w_packed, "_dge_frag", False) else 0
    if d.w_layout and _PF["on"]:
        # L2 warm-up hint for the low-resolution kernel: the launch warms the weights of the low-resolution launch that followed it
        # the LAST time it ran (a step repeats its launch sequence; packed copies keep their addresses) - dge_conv_desc.prefetch_w
        # The chain is keyed by (stream, address): launches of different streams (the three loss windows) do not follow each other.
        # An entry holds a WEAK reference to the packed tensor it points at: a re-allocated copy (load_state_dict, a new model) makes
        # the entry stale and it is dropped instead of warming freed memory; the table is bounded.
        sk = (_stream().value or 0)
        key = w_packed.data_ptr()
        prev = _PF["prev"].get(sk) if isinstance(_PF["prev"], dict) else None
        if not isinstance(_PF["prev"], dict):
            _PF["prev"] = {}
        if prev is not None and prev != key:
            if len(_PF["next"]) > 512:
                _PF["next"].clear()
            _PF["next"][(sk, prev)] = (key, w_packed.shape[1], w_packed.shape[2], _weakref.ref(w_packed))
        _PF["prev"][sk] = key
        nxt = _PF["next"].get((sk, key))
        if nxt is not None:
            t = nxt[3]()
            if t is None or t.data_ptr() != nxt[0]:
                del _PF["next"][(sk, key)]
            else:
                d.prefetch_w, d.prefetch_ntot, d.prefetch_cin = C.c_void_p(nxt[0]), int(nxt[1]), int(nxt[2])
    if prep is not None:
        if stats is None or dot_src is None:
            raise DgeError("conv2d: prep needs stats and dot_src")
        pn = prep.get("noise")
        d.prep, d.prep_gain = 1, float(prep["gain"])
        d.prep_noise, d.prep_ns = _f32(pn), _f32(prep.get("ns") if pn is not None else None)
        d.prep_noise_batch = 1 if pn is None else pn.shape[0]
        d.prep_stats = _f32(prep["stats"].alloc(nslot))
    if in_bwd is not None:
        if dot_src is None or stats is not None or prep is not None:
            raise DgeError("conv2d: in_bwd needs dot_src and excludes stats / prep")
        nblk = ((H + 15) // 16) * ((W + 15) // 16) * B
        d.stats_slots = max(1, min(64, nblk // 16))
        pn = in_bwd.get("noise")
        d.in_bwd_coef = _f32(in_bwd["coef"])
        d.prep_noise, d.prep_noise_batch = _f32(pn), 1 if pn is None else pn.shape[0]
        if in_bwd.get("fr") is not None:          # FromRGB reduction flavour: nothing is stored, fr = SlotStats-like holder of [slots,B,cout,4]
            d.fr_img4, d.in_bwd_extra, d.in_bwd_extra_scale = _f32(in_bwd["img4"]), _p(in_bwd.get("extra")), float(in_bwd.get("extra_scale", 1.0))
            in_bwd["fr"].buf = zeros((d.stats_slots, B, cout, 4), x.device)
            d.fr_out = _f32(in_bwd["fr"].buf)
        elif in_bwd.get("red") is not None:
            d.prep_stats = _f32(in_bwd["red"].alloc(d.stats_slots))
        else:                                     # block-input form: no activation / reductions, pooled skip gradient added
            d.prep_noise = None
            d.in_bwd_extra, d.in_bwd_extra_scale = _p(in_bwd.get("extra")), float(in_bwd.get("extra_scale", 1.0))
    if rgb is not None:
        d.rgb_w, d.rgb_style, d.rgb_bias, d.rgb_out = _f32(rgb["w"]), _f32(rgb["style"]), _f32(rgb["bias"]), _f32(rgb["out"])
        d.rgb_wscale, d.rgb_skip_y = float(rgb["wscale"]), 1 if rgb.get("skip_y") else 0
    d.noise_batch = 1 if noise is None else noise.shape[0]
    d.noise_w_per_channel = 0 if (noise_w is None or noise_w.numel() == 1) else 1
    d.act, d.bias_scale, d.gain, d.add_scale, d.dtype = act, bias_scale, gain, add_scale, dt
    if w_packed.dtype != x.dtype:
        raise DgeError("conv2d: packed weight dtype differs from activation dtype")
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().dge_conv2d(C.byref(d), _stream()), "dge_conv2d")
        e1.record()
        # algorithmic work: a folded up layer / its adjoint count as the 3x3 transposed conv they replace
        # (9*Cin*Cout MACs per INPUT pixel), everything else k*k*Cin*Cout per output pixel
        if up:
            macs = 9.0 * Cin * cout * H * W
        elif in_s2d or in_t2d:
            macs = 9.0 * (Cin // 4) * cout * H * W
        else:
            macs = float(ksize * ksize) * Cin * cout * H * W
        # algorithmic bytes: every operand tensor crosses HBM once (input, output, optional addend / dot_src, packed weights)
        skip_y = rgb is not None and rgb.get("skip_y")
        abytes = sum(t.numel() * t.element_size() for t in (x, None if skip_y else out, addend, dot_src, w_packed,
                                                            rgb["out"] if rgb is not None else None) if t is not None)
        PROFILE.append((e0, e1, 2.0 * macs * B, (B, H, W, Cin, cout, ksize, up, in_s2d), abytes))
    else:
        check(lib().dge_conv2d(C.byref(d), _stream()), "dge_conv2d")
    if partial is not None and not lazy:
        check(lib().dge_sum_slots(_p(partial), _p(stats), nslot, stats.numel(), 1, _stream()), "dge_sum_slots")
    if rgb is not None and rgb.get("skip_y"):
        return None
    if pool_out and pool_mask:
        return out, mask
    return out


def conv_in_bwd_supported(B, H, W, cin, cout, dtype):
    """True when a 3x3 data-gradient launch cin -> cout of this shape may carry the instance-norm backward epilogue (conv2d(in_bwd=...))"""
    return bool(lib().dge_conv_in_bwd_supported(int(B), int(H), int(W), int(cin), int(cout), 3, int(dtype)))


def conv_in_bwd_x_supported(B, H, W, cin, cout, dtype):
    return bool(lib().dge_conv_in_bwd_x_supported(int(B), int(H), int(W), int(cin), int(cout), 3, int(dtype)))


def conv_in_bwd_fromrgb_supported(B, H, W, cin, cout, dtype):
    return bool(lib().dge_conv_in_bwd_fromrgb_supported(int(B), int(H), int(W), int(cin), int(cout), 3, int(dtype)))


def up_pp_supported(B, H, W, cin, cout, dtype):
    """True when an up layer of this shape runs on the ping-pong kernel of csrc/up_pp.hip (Cin >= 128; H, W = the INPUT grid)"""
    return bool(lib().dge_up_pp_supported(int(B), int(H), int(W), int(cin), int(cout), int(dtype)))


def pack_up_pp(w_units, cout, cin, in_scale=None, out_scale=None, gain=1.0, out=None):
    """Weight image of up_pp from pack_upconv_weight's bf16 [9, Cout, Cin] units: one shared copy, or - with in_scale [B, Cin] /
    out_scale [B, Cout] - B copies with style, demodulation and gain folded in (stylegan2_generator.py:858-875).  [nb, 9*Cin*Cout] bf16."""
    if w_units.dtype != torch.bfloat16:
        raise DgeError("pack_up_pp: expects the bf16 units of pack_upconv_weight")
    nb = 1
    for t in (in_scale, out_scale):
        if t is not None:
            nb = t.shape[0]
    if out is None:
        out = torch.empty((nb, 9 * cin * cout), dtype=torch.bfloat16, device=w_units.device)
    check(lib().dge_pack_up_pp(_p(w_units), _p(out), int(cout), int(cin), _f32(in_scale), _f32(out_scale), float(gain), nb, _stream()),
          "dge_pack_up_pp")
    return out


def up_pp(x, w_img, cout, bias=None, bias_scale=1.0, noise=None, noise_w=None, act=ACT_NONE, gain=1.0):
    """x [B,H,W,Cin] bf16 -> y [B,2H,2W,cout]: conv_transpose2d(stride 2) + 4x4 FIR + noise / bias / activation (dge_up_pp) with the
    per-sample weight image of pack_up_pp (style, demodulation and gain folded in)."""
    B, H, W, Cin = x.shape
    if x.dtype != torch.bfloat16 or not x.is_cuda:
        raise DgeError("up_pp: bf16 CUDA activations only (the HIP path has no CPU fallback)")
    y = torch.empty((B, 2 * H, 2 * W, cout), dtype=x.dtype, device=x.device)
    nbs = 0 if (noise is None or noise.shape[0] == 1) else 4 * H * W
    if noise_w is not None and noise_w.numel() != 1:
        raise DgeError("up_pp: one noise strength per layer (stylegan2_generator.py:911-916)")
    wbs = 0 if w_img.shape[0] == 1 else w_img.stride(0)

    def launch():
        check(lib().dge_up_pp(_p(x), _p(w_img), int(wbs), _p(y), _f32(noise), nbs, _f32(noise_w), _f32(bias), float(bias_scale),
                              float(gain), int(act), B, H, W, Cin, int(cout), _stream()), "dge_up_pp")
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(); e1.record()
        abytes = sum(t.numel() * t.element_size() for t in (x, y)) + 18 * Cin * cout * B
        PROFILE.append((e0, e1, 2.0 * 9.0 * Cin * cout * H * W * B, (B, H, W, Cin, cout, 3, "upfir", False), abytes))
    else:
        launch()
    if KERNEL_LOG is not None:
        KERNEL_LOG.append((last_kernel(), _stream()))
    return y


def conv_pp_supported(B, H, W, cin, cout, dtype):
    """True when a 3x3 stride-1 launch of this shape runs on the ping-pong implicit GEMM (csrc/conv_pp.hip: C >= 128 at 64^2 .. 256^2)"""
    return bool(lib().dge_conv_pp_supported(int(B), int(H), int(W), int(cin), int(cout), int(dtype)))


def pack_conv_pp(w, wscale=1.0, in_scale=None, out_scale=None, gain=1.0, dgrad=False, out=None):
    """LDS image of w [Cout,Cin,3,3] f32 for conv_pp: one shared copy, or - with in_scale [B,K] / out_scale [B,N] - B per-sample copies with
    style, demodulation and gain folded in (the reference's fused modulation, stylegan2_generator.py:858-875).  dgrad: the copy the
    data gradient reads ((n, k) = (in, out) channel, taps flipped).  Returns a [nb, 9*N*K] bf16 tensor."""
    cout, cin = w.shape[0], w.shape[1]
    N, K = (cin, cout) if dgrad else (cout, cin)
    nb = 1
    for t in (in_scale, out_scale):
        if t is not None:
            nb = t.shape[0]
    if out is None:
        out = torch.empty((nb, 9 * N * K), dtype=torch.bfloat16, device=w.device)
    check(lib().dge_pack_conv_pp(_f32(w.detach().contiguous()), _p(out), N, K, float(wscale), _f32(in_scale), _f32(out_scale), float(gain),
                                 nb, 1 if dgrad else 0, _stream()), "dge_pack_conv_pp")
    return out


def pack_conv_pp_rows(w_rows, N, in_scale=None, in_period=None, out_scale=None, gain=1.0, out=None, t2d=False):
    """The conv_pp weight image from an f32 pack_conv_weight copy w_rows [9, Npad, K] (rows 0 .. N-1 are used): the data-gradient
    layouts of the up layers.  in_scale [B, in_period] repeats along K (space-to-depth: the four phases of a channel)."""
    if w_rows.dtype != torch.float32 or w_rows.ndim != 3 or w_rows.shape[0] != 9:
        raise DgeError("pack_conv_pp_rows: expects an f32 [9, Npad, K] packed weight")
    K = w_rows.shape[2]
    nb = 1
    for t in (in_scale, out_scale):
        if t is not None:
            nb = t.shape[0]
    if out is None:
        out = torch.empty((nb, (4 if t2d else 9) * N * K), dtype=torch.bfloat16, device=w_rows.device)
    per = K if in_period is None else int(in_period)
    check(lib().dge_pack_conv_pp_rows(_f32(w_rows), int(w_rows.shape[1]), _p(out), int(N), int(K), _f32(in_scale), per, _f32(out_scale), float(gain),
                                      nb, 1 if t2d else 0, _stream()), "dge_pack_conv_pp_rows")
    return out


def conv_pp(x, w_pp, cout, out_scale=None, bias=None, bias_scale=1.0, noise=None, noise_w=None, act=ACT_NONE, gain=1.0, out=None,
            dgrad=False, in_s2d=False, dot_src=None, addend=None, add_scale=1.0, stats=None, prep=None, relu_mask=None, in_t2d=False):
    """x [B,H,W,Cin] bf16 -> y [B,H,W,cout]: 3x3 stride-1 conv with the weights of pack_conv_pp (w_pp.shape[0] = 1: shared, = B: per sample).
    dgrad: the data-gradient form (dge_conv_pp_desc.dgrad) with conv2d()'s menu: out_scale applied after the statistics
    stats (SlotStats) += (sum a*dot_src, sum a), addend, prep = dict(gain, noise, ns, stats=SlotStats), relu_mask; in_s2d: x is the fine
    grid [B,2H,2W,Cin/4]."""
    B, H, W, Cin = x.shape
    if in_s2d:
        H, W, Cin = H // 2, W // 2, Cin * 4
    if in_t2d:            # x is fir_t2d's [B,H+1,W+1,4C]: the conv runs on the H x W grid with the 4 taps of the phase form
        H, W = H - 1, W - 1
    if out is None:
        out = torch.empty((B, H, W, cout), dtype=x.dtype, device=x.device)
    if w_pp.shape[0] not in (1, B) or w_pp.shape[1] != (4 if in_t2d else 9) * Cin * cout or w_pp.dtype != torch.bfloat16:
        raise DgeError("conv_pp: weight image does not match the launch")
    d = ConvPPDesc()
    d.x, d.w_pp, d.y = _p(x), _p(w_pp), _p(out)
    d.w_bstride = 0 if w_pp.shape[0] == 1 else w_pp.shape[1]
    d.out_scale, d.bias, d.noise, d.noise_w = _f32(out_scale), _f32(bias), _f32(noise), _f32(noise_w)
    d.B, d.H, d.W, d.Cin, d.Cout = B, H, W, Cin, cout
    d.noise_batch = 1 if noise is None else noise.shape[0]
    d.noise_w_per_channel = 0 if (noise_w is None or noise_w.numel() == 1) else 1
    d.act, d.bias_scale, d.gain = act, bias_scale, gain
    if dgrad:
        if relu_mask is not None:
            if dot_src is not None or prep is not None:
                raise DgeError("conv_pp: relu_mask excludes dot_src / prep")
            dot_src = relu_mask
        d.dgrad, d.in_s2d, d.in_t2d, d.mask_relu = 1, 1 if in_s2d else 0, 1 if in_t2d else 0, 0 if relu_mask is None else 1
        d.dot_src, d.addend, d.add_scale = _p(dot_src), _p(addend), float(add_scale)
        nslot = max(1, min(64, (((H + 15) // 16) * ((W + 31) // 32) * B) // 16))
        if stats is not None:
            d.stats = _f32(stats.alloc(nslot) if isinstance(stats, SlotStats) else stats)
            if not isinstance(stats, SlotStats):
                nslot = 1
        d.stats_slots = nslot
        if prep is not None:
            if dot_src is None:
                raise DgeError("conv_pp: prep needs dot_src")
            pn = prep.get("noise")
            d.prep, d.prep_gain = 1, float(prep["gain"])
            d.prep_noise, d.prep_ns = _f32(pn), _f32(prep.get("ns") if pn is not None else None)
            d.prep_noise_batch = 1 if pn is None else pn.shape[0]
            d.prep_stats = _f32(prep["stats"].alloc(nslot))
    elif in_s2d or in_t2d or dot_src is not None or addend is not None or stats is not None or prep is not None or relu_mask is not None:
        raise DgeError("conv_pp: in_s2d / dot_src / addend / stats / prep / relu_mask need dgrad=True")
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().dge_conv_pp(C.byref(d), _stream()), "dge_conv_pp")
        e1.record()
        abytes = sum(t.numel() * t.element_size() for t in (x, out, w_pp, dot_src, addend) if t is not None)
        macs = 9.0 * (Cin // 4 if (in_s2d or in_t2d) else Cin) * cout * H * W         # (an up layer's adjoint counts as the 3x3 transposed conv it replaces)
        PROFILE.append((e0, e1, 2.0 * macs * B, (B, H, W, Cin, cout, 3, False, bool(in_s2d or in_t2d)), abytes))
    else:
        check(lib().dge_conv_pp(C.byref(d), _stream()), "dge_conv_pp")
    if KERNEL_LOG is not None:
        KERNEL_LOG.append((last_kernel(), _stream()))
    return out


def torgb(x, wrgb, style, bias, prev, wscale):
    B, H, W, Cin = x.shape
    img = torch.empty((B, 3, H, W), dtype=torch.float32, device=x.device)
    check(lib().dge_torgb(_p(x), _f32(wrgb), _f32(style), _f32(bias), _f32(prev), _p(img), B, H, W, Cin, float(wscale),
                          dtype_of(x), _stream()), "dge_torgb")
    return img


def nchw_to_nhwc(src, B, dtype):
    """src: [sB, C, H, W] f32 with sB in {1, B} -> [B, H, W, C] of dtype."""
    sB, Cc, H, W = src.shape
    dst = torch.empty((B, H, W, Cc), dtype=tdtype(dtype), device=src.device)
    check(lib().dge_nchw_to_nhwc(_f32(src.contiguous()), _p(dst), B, Cc, H * W, sB, dtype, _stream()), "dge_nchw_to_nhwc")
    return dst


def nhwc_to_nchw(src):
    B, H, W, Cc = src.shape
    dst = torch.empty((B, Cc, H, W), dtype=torch.float32, device=src.device)
    check(lib().dge_nhwc_to_nchw(_p(src), _p(dst), B, Cc, H * W, dtype_of(src), _stream()), "dge_nhwc_to_nchw")
    return dst


# ------------------------------------------------------------------ encoder streaming ops
def fromrgb(img, w, bias, dtype, stats=None, img4=False):
    """img4: also return the image in pixel-major form [B,H,W,4] f32 = (r, g, b, 1) (the backward's fused FromRGB reduction reads it)"""
    B, _, H, W = img.shape
    Cc = w.shape[0]
    y = torch.empty((B, H, W, Cc), dtype=tdtype(dtype), device=img.device)
    i4 = torch.empty((B, H, W, 4), dtype=torch.float32, device=img.device) if img4 else None
    check(lib().dge_fromrgb2(_f32(img.contiguous()), _f32(w.reshape(Cc, 3)), _f32(bias), _p(y), _f32(stats), _f32(i4), B, H * W, Cc,
                             dtype, _stream()), "dge_fromrgb")
    return (y, i4) if img4 else y


def stats_finalize(stats, npix, eps=1e-8, musig_out=None):
    """stats: [B,C,2] tensor or SlotStats; musig_out: optional preallocated [B, 2C] destination of (mean, std)"""
    nslot = 1
    if isinstance(stats, SlotStats):
        B, Cc, nslot, stats = stats.B, stats.C, stats.nslot, stats.buf
    else:
        B, Cc, _ = stats.shape
    musig = torch.empty((B, 2 * Cc), dtype=torch.float32, device=stats.device) if musig_out is None else musig_out
    sc = torch.empty((B, Cc), dtype=torch.float32, device=stats.device)
    sh = torch.empty((B, Cc), dtype=torch.float32, device=stats.device)
    check(lib().dge_stats_finalize_slots(_f32(stats), nslot, _p(musig), _p(sc), _p(sh), B, Cc, int(npix), float(eps), _stream()),
          "dge_stats_finalize")
    return musig, sc, sh


def blend(x, z=None, sc=None, sh=None, pool=False, alpha=1.0, beta=0.0, stats=None, mask=False):
    """mask=True (pooling blend without sc / sh): also returns the sign mask of x for act_bwd_mask -> (y, mask)"""
    B, H, W, Cc = x.shape
    OH, OW = (H // 2, W // 2) if pool else (H, W)
    y = torch.empty((B, OH, OW, Cc), dtype=x.dtype, device=x.device)
    if mask:
        if not pool or sc is not None or sh is not None:
            raise DgeError("blend: the sign mask is offered for the plain pooling blend")
        ep = 8 if x.dtype == torch.bfloat16 else 4
        m = torch.empty((B, OH * OW, Cc // ep), dtype=torch.int32, device=x.device)
        check(lib().dge_blend_pool_mask(_p(x), _p(z), _p(y), _f32(stats), _p(m), B, OH, OW, Cc, float(alpha), float(beta), dtype_of(x),
                                        _stream()), "dge_blend_pool_mask")
        return y, m
    check(lib().dge_blend(_p(x), _p(z), _p(y), _f32(sc), _f32(sh), _f32(stats), B, OH, OW, Cc, 1 if pool else 0,
                          float(alpha), float(beta), dtype_of(x), _stream()), "dge_blend")
    return y


# ------------------------------------------------------------------ StyleGAN2 backward ops
def modconv_bwd_prep(gx, x, d, noise, gain, R=None):
    B, H, W, Cc = x.shape
    gy = torch.empty_like(x)
    check(lib().dge_modconv_bwd_prep(_p(gx), _p(x), _f32(d), _f32(noise), _p(gy), _f32(R), B, H * W, Cc,
                                     1 if noise is None else noise.shape[0], float(gain), dtype_of(x), _stream()),
          "dge_modconv_bwd_prep")
    return gy


def torgb_bwd_prep(gimg, x, wrgb, style, wscale, noise, ns, gain):
    """top of the synthesis backward: toRGB adjoint + tail backward of the layer in one pass -> (g_z, gs [B,C], P [B,C,2])"""
    B, H, W, Cc = x.shape
    gz = torch.empty_like(x)
    gs = zeros((B, Cc), x.device)
    P = zeros((B, Cc, 2), x.device)
    check(lib().dge_torgb_bwd_prep(_f32(gimg), _p(x), _f32(wrgb), _f32(style), _f32(noise), _f32(ns if noise is not None else None),
                                   1 if noise is None else noise.shape[0], _p(gz), _p(gs), _p(P), B, H * W, Cc, float(wscale),
                                   float(gain), dtype_of(x), _stream()), "dge_torgb_bwd_prep")
    return gz, gs, P


def demod_bwd_prep(P, d, bias, bscale=1.0):
    """t [B,C] from the fused tail-backward sums P (a [B,C,2] tensor or a SlotStats filled by conv2d(prep=...))"""
    B, Cc = d.shape
    nslot = 1
    if isinstance(P, SlotStats):
        nslot, P = P.nslot, P.buf
    t = torch.empty_like(d)
    check(lib().dge_demod_bwd_prep(_f32(P), nslot, _f32(d), _f32(bias), _p(t), B, Cc, float(bscale), _stream()), "dge_demod_bwd_prep")
    return t


def demod_bwd(R, d, bias, noise_strength, bscale=1.0):
    B, Cc = d.shape
    t = torch.empty_like(d)
    check(lib().dge_demod_bwd(_f32(R), _f32(d), _f32(bias), _f32(noise_strength), _p(t), B, Cc, float(bscale), _stream()),
          "dge_demod_bwd")
    return t


def linear_t(x, w, y, mul=None, scale=1.0, accumulate=False, incx=1, incy=1, ldx=None, ldy=None, B=None, O=None):
    """y[b, k*incy] (+)= scale * mul[b,k] * sum_o x[b, o*incx] * w[o, k].  x / y are given as tensors whose
    data_ptr is element (0,0); row strides default to tensor strides."""
    O_, K = w.shape
    B = x.shape[0] if B is None else B
    O = O_ if O is None else O
    ldx = x.stride(0) if ldx is None else ldx
    ldy = y.stride(0) if ldy is None else ldy
    check(lib().dge_linear_t(C.c_void_p(x.data_ptr()), ldx, incx, _f32(w), _f32(mul), C.c_void_p(y.data_ptr()), ldy, incy,
                             B, O, K, float(scale), 1 if accumulate else 0, _stream()), "dge_linear_t")
    return y


def fir_t2d(g, scale=None):
    """g [B,2H,2W,C] -> Z [B,H+1,W+1,4C] = scale * FIR^T(g) stored t-grid-to-depth (dge_fir_t2d): the input of conv2d(in_t2d=True)"""
    B, FH, FW, Cc = g.shape
    H, W = FH // 2, FW // 2
    z = torch.empty((B, H + 1, W + 1, 4 * Cc), dtype=g.dtype, device=g.device)
    check(lib().dge_fir_t2d(_p(g), _f32(scale), _p(z), B, H, W, Cc, dtype_of(g), _stream()), "dge_fir_t2d")
    return z


def s2_style_grads(blocks, g_wp, wscale):
    """blocks: list of dicts - conv block: P (SlotStats or [B,out_c,2]), st (SlotStats), d, s, bias, wsq, bscale, wstyle, row;
    toRGB block: gs, wstyle, row.  One launch for every style gradient of the synthesis backward (dge_s2_style_grads)."""
    from ._lib import S2GradEntry
    arr = (S2GradEntry * len(blocks))()
    B, nrows, K = g_wp.shape
    for e, blk in zip(arr, blocks):
        e.wstyle, e.row = _f32(blk["wstyle"]), int(blk["row"])
        if "gs" in blk:
            e.gs, e.in_c = _f32(blk["gs"]), blk["gs"].shape[1]
            continue
        P, st = blk["P"], blk["st"]
        e.nslot_p, e.nslot_s = 1, 1
        if isinstance(P, SlotStats):
            e.nslot_p, P = P.nslot, P.buf
        if isinstance(st, SlotStats):
            e.nslot_s, st = st.nslot, st.buf
        e.P, e.st, e.d, e.s, e.bias, e.wsq = _f32(P), _f32(st), _f32(blk["d"]), _f32(blk["s"]), _f32(blk["bias"]), _f32(blk["wsq"])
        e.out_c, e.in_c = blk["d"].shape[1], blk["s"].shape[1]
        e.bscale = float(blk["bscale"])
    check(lib().dge_s2_style_grads(arr, len(blocks), _p(g_wp), B, nrows, K, float(wscale), _stream()), "dge_s2_style_grads")
    return g_wp


def torgb_bwd(gimg, x, wrgb, style, wscale):
    B, H, W, Cc = x.shape
    gx = torch.empty_like(x)
    gs = zeros((B, Cc), x.device)
    check(lib().dge_torgb_bwd(_f32(gimg), _p(x), _f32(wrgb), _f32(style), _p(gx), _p(gs), B, H * W, Cc, float(wscale),
                              dtype_of(x), _stream()), "dge_torgb_bwd")
    return gx, gs


def up2_bwd(g):
    B, Cc, H, W = g.shape
    out = torch.empty((B, Cc, H // 2, W // 2), dtype=torch.float32, device=g.device)
    check(lib().dge_up2_bwd(_f32(g), _p(out), B * Cc, H // 2, W // 2, _stream()), "dge_up2_bwd")
    return out


# ------------------------------------------------------------------ encoder backward ops
def conv_wgrad(g, x, dw, in_scale=None, in_shift=None):
    """dw [Cout,Cin,k,k] f32 (pre-zeroed or accumulating) += wgrad(g, x*in_scale+in_shift)."""
    B, H, W, Cin = x.shape
    cout, _, k, _ = dw.shape
    check(lib().dge_conv_wgrad(_p(g), _p(x), _f32(in_scale), _f32(in_shift), _f32(dw), B, H, W, cout, Cin, k, dtype_of(x),
                               _stream()), "dge_conv_wgrad")
    return dw


def conv_wgrad_dots(g, x, dw, in_scale, in_shift, w, dots):
    """conv_wgrad that also fills `dots` (SlotStats(B, Cin)) with (sum g_x*x, sum g_x) of the layer's data gradient g_x, where the
    streaming weight-gradient kernel covers the shape: returns True then; False (nothing launched) otherwise."""
    B, H, W, Cin = x.shape
    cout = dw.shape[0]
    if dw.shape[2] != 3 or is_deterministic():
        return False
    nslot = max(1, min(16, (H * W) // 4096))
    buf = dots.alloc(nslot)
    r = lib().dge_conv_wgrad_dots(_p(g), _p(x), _f32(in_scale), _f32(in_shift), _f32(dw), _f32(w.detach()), _f32(buf), nslot, B, H, W, cout, Cin,
                                  dtype_of(x), _stream())
    if r == 1:
        dots.buf, dots.nslot = None, 1
        return False
    check(r, "dge_conv_wgrad_dots")
    return True


class DeferredSums:
    """Planar slot sums whose results nobody reads before the end of a backward: collected, then run as ONE launch
    (dge_sum_slots_planar_multi) by flush()."""

    def __init__(self):
        self.items = []

    def add(self, part, red):
        self.items.append((part, red))
        if len(self.items) == 32:
            self.flush()
        return red

    def flush(self):
        if not self.items:
            return
        from ._lib import SumPlanarEntry
        arr = (SumPlanarEntry * len(self.items))()
        for e, (part, red) in zip(arr, self.items):
            e.partial, e.out = _p(part), _p(red)
            e.nslot, e.C, e.NS = part.shape
        check(lib().dge_sum_slots_planar_multi(arr, len(self.items), _stream()), "dge_sum_slots_planar_multi")
        self.items = []


def _sum_planar(part, red, defer=None):
    """part [B, C, NS] per-sample sums -> red [NS, C] (each reduction a contiguous vector)."""
    if defer is not None:
        return defer.add(part, red)
    B, Cc, NS = part.shape
    check(lib().dge_sum_slots_planar(_p(part), _p(red), B, Cc, NS, _stream()), "dge_sum_slots_planar")
    return red


def act_bwd(gup, a, noise=None, pool=False, scale=1.0, red=None, slope=0.2, planar=False, defer=None):
    """red [C, 2|3] (pre-zeroed; [2|3, C] when planar) receives the batch-summed reductions: the kernel writes per-sample
    partial sums (no cross-sample contention on the atomics) that are added over the batch here."""
    B, H, W, Cc = a.shape
    gpre = torch.empty_like(a)
    ncol = 2 if red is None else (red.shape[0] if planar else red.shape[-1])
    part = zeros((B, Cc, ncol), a.device) if red is not None else None
    check(lib().dge_act_bwd(_p(gup), _p(a), _f32(noise), _p(gpre), _f32(part), ncol, B, H, W, Cc,
                            1 if pool else 0, float(scale), float(slope), dtype_of(a), _stream()), "dge_act_bwd")
    if red is not None:
        _sum_planar(part, red, defer) if planar else _sum_over_batch(part, red)
    return gpre


def act_bwd_mask(gup, mask, noise, scale=1.0, red=None, slope=0.2, planar=False, defer=None):
    """act_bwd(pool=True) from the sign mask of blend(..., mask=True): gup [B,H/2,W/2,C] -> g_pre [B,H,W,C]"""
    B, UH, UW, Cc = gup.shape
    H, W = 2 * UH, 2 * UW
    gpre = torch.empty((B, H, W, Cc), dtype=gup.dtype, device=gup.device)
    ncol = 2 if red is None else (red.shape[0] if planar else red.shape[-1])
    part = zeros((B, Cc, ncol), gup.device) if red is not None else None
    check(lib().dge_act_bwd_mask(_p(gup), _p(mask), _f32(noise), _p(gpre), _f32(part), ncol, B, H, W, Cc, float(scale), float(slope),
                                 dtype_of(gup), _stream()), "dge_act_bwd_mask")
    if red is not None:
        _sum_planar(part, red, defer) if planar else _sum_over_batch(part, red)
    return gpre


def in_bwd_coef(dots, gms, musig, sc, sh, npix):
    B, Cc = sc.shape
    coef = torch.empty((B, Cc, 3), dtype=torch.float32, device=sc.device)
    nslot = 1
    if isinstance(dots, SlotStats):
        nslot, dots = dots.nslot, dots.buf
    check(lib().dge_in_bwd_coef_slots(_f32(dots), nslot, _f32(gms), _f32(musig), _f32(sc), _f32(sh), _p(coef), B, Cc, int(npix),
                                      _stream()), "dge_in_bwd_coef")
    return coef


def in_bwd(gy, x, coef, extra=None, extra_pool=False, extra_scale=1.0, noise=None, act=False, red=None, planar=False, defer=None):
    """red [C, 2] (or [2, C] when planar).  `coef`: [B,C,3] from in_bwd_coef, or the tuple (dots, gms, musig, sc, sh, npix) of its
    arguments - the launch then computes the coefficients itself (dge_in_bwd_fused, C <= 512)."""
    B, H, W, Cc = x.shape
    gout = torch.empty_like(x)
    part = zeros((B, Cc, 2), x.device) if red is not None else None
    if isinstance(coef, tuple) and Cc > 512:
        coef = in_bwd_coef(*coef)
    if isinstance(coef, tuple):
        dots, gms, musig, sc, sh, npix = coef
        nslot = 1
        if isinstance(dots, SlotStats):
            nslot, dots = dots.nslot, dots.buf
        check(lib().dge_in_bwd_fused(_p(gy), _p(x), _f32(dots), nslot, _f32(gms), _f32(musig), _f32(sc), _f32(sh), int(npix), _p(extra),
                                     _f32(noise), _p(gout), _f32(part), B, H, W, Cc, 1 if extra_pool else 0, float(extra_scale),
                                     1 if act else 0, dtype_of(x), _stream()), "dge_in_bwd_fused")
    else:
        check(lib().dge_in_bwd(_p(gy), _p(x), _f32(coef), _p(extra), _f32(noise), _p(gout), _f32(part), B, H, W, Cc,
                               1 if extra_pool else 0, float(extra_scale), 1 if act else 0, dtype_of(x), _stream()), "dge_in_bwd")
    if red is not None:
        _sum_planar(part, red, defer) if planar else _sum_over_batch(part, red)
    return gout


def in_bwd_fromrgb(gy, x0, coef, img, extra=None, extra_pool=False, extra_scale=1.0, defer=None):
    """Last step of the encoder backward: in_bwd (coefficients computed in the launch, `coef` = (dots, gms, musig, sc, sh, npix))
    on the FromRGB output x0 with the FromRGB parameter gradients reduced from the result in registers -> [4, C] (planar: weight
    gradient rows 0..2, bias gradient row 3).  The gradient w.r.t. x0 is never stored."""
    B, H, W, Cc = x0.shape
    dots, gms, musig, sc, sh, npix = coef
    nslot = 1
    if isinstance(dots, SlotStats):
        nslot, dots = dots.nslot, dots.buf
    part = zeros((B, Cc, 4), x0.device)
    check(lib().dge_in_bwd_fromrgb(_p(gy), _p(x0), _f32(dots), nslot, _f32(gms), _f32(musig), _f32(sc), _f32(sh), int(npix), _p(extra),
                                   _f32(img.contiguous()), _p(part), B, H, W, Cc, 1 if extra_pool else 0, float(extra_scale),
                                   dtype_of(x0), _stream()), "dge_in_bwd_fromrgb")
    return _sum_planar(part, torch.empty((4, Cc), dtype=torch.float32, device=x0.device), defer)


def conv_pool_supported(B, H, W, cin, cout, ksize, dtype):
    return bool(lib().dge_conv_pool_supported(B, H, W, cin, cout, ksize, dtype))


def conv_rgb_supported(B, H, W, cin, cout, ksize, dtype):
    return bool(lib().dge_conv_rgb_supported(B, H, W, cin, cout, ksize, dtype))


def rgb_upsample_add(img, prev):
    """img [B,3,H,W] += up2(prev [B,3,H/2,W/2]) in place (the skip connection of SynthesisModule.forward :517-522)"""
    B, Cc, H, W = img.shape
    check(lib().dge_rgb_upsample_add(_p(img), _f32(prev), B * Cc, H, W, _stream()), "dge_rgb_upsample_add")
    return img


def chan_sum(x, scale=1.0):
    B, H, W, Cc = x.shape
    part = zeros((B, Cc), x.device)
    check(lib().dge_chan_sum(_p(x), _p(part), B, H * W, Cc, float(scale), dtype_of(x), _stream()), "dge_chan_sum")
    return _sum_over_batch(part)


def fromrgb_bwd(gx, x0, img, planar=False, defer=None):
    """-> [C, 4] (weight gradient columns 0..2, bias gradient column 3); planar: [4, C]"""
    B, H, W, Cc = x0.shape
    part = zeros((B, Cc, 4), x0.device)
    check(lib().dge_fromrgb_bwd(_p(gx), _p(x0), _f32(img.contiguous()), _p(part), B, H * W, Cc, dtype_of(x0), _stream()),
          "dge_fromrgb_bwd")
    if planar:
        return _sum_planar(part, torch.empty((4, Cc), dtype=torch.float32, device=x0.device), defer)
    return _sum_over_batch(part)


def dense_wgrad(gy, x, gw, gb=None, accumulate=False):
    """gy: [B,O] view (row stride free), x: [B,I]; gw [O,I], gb [O]."""
    B, O = gy.shape
    I = x.shape[1]
    check(lib().dge_dense_wgrad(C.c_void_p(gy.data_ptr()), gy.stride(0), C.c_void_p(x.data_ptr()), x.stride(0), _f32(gw),
                                _f32(gb), B, O, I, 1 if accumulate else 0, _stream()), "dge_dense_wgrad")
    return gw, gb


def scale_(t, factor):
    """in-place t *= factor on the device (dge_axpy_scalar)."""
    check(lib().dge_axpy_scalar(_f32(t), None, _p(t), t.numel(), float(factor), 0, _stream()), "dge_axpy_scalar")
    return t


# ------------------------------------------------------------------ StyleGAN1 ops
def blur_noise_act(x, noise, noise_w, bias, blur=True, stats=None, act=True):
    B, H, W, Cc = x.shape
    y = torch.empty_like(x)
    check(lib().dge_blur_noise_act(_p(x), _f32(noise), _f32(noise_w), _f32(bias), _p(y), _f32(stats), B, H, W, Cc,
                                   1 if blur else 0, 1 if noise is None else noise.shape[0], 1 if act else 0, dtype_of(x),
                                   _stream()), "dge_blur_noise_act")
    return y


def affine_compose(sc, sh, style):
    B, Cc = sc.shape
    a = torch.empty_like(sc)
    b = torch.empty_like(sc)
    check(lib().dge_affine_compose(_f32(sc), _f32(sh), _f32(style), _p(a), _p(b), B, Cc, _stream()), "dge_affine_compose")
    return a, b


def lerp_layers(w, avg, coefs):
    """w [B,D]; avg [D] or [L,D]; coefs [L] -> [B,L,D]"""
    B, D = w.shape
    L = coefs.numel()
    out = torch.empty((B, L, D), dtype=torch.float32, device=w.device)
    stride = D if avg.numel() == L * D else 0
    check(lib().dge_lerp_layers(_f32(w), _f32(avg.contiguous()), stride, _f32(coefs.contiguous()), _p(out), B, L, D, _stream()),
          "dge_lerp_layers")
    return out


# ------------------------------------------------------------------ PGGAN ops
def pixelnorm_nhwc(x, eps=1e-8):
    B, H, W, Cc = x.shape
    y = torch.empty_like(x)
    check(lib().dge_pixelnorm_nhwc(_p(x), _p(y), B * H * W, Cc, float(eps), dtype_of(x), _stream()), "dge_pixelnorm_nhwc")
    return y


# ------------------------------------------------------------------ BigGAN ops
def cbn_affine(scale, offset, mean, var, eps):
    """scale/offset: [B,C] (or [1,C] batch-shared); mean/var: [C] -> a, b [B,C]"""
    B, Cc = scale.shape
    a = torch.empty((B, Cc), dtype=torch.float32, device=scale.device)
    b = torch.empty_like(a)
    # scale / offset may be column slices of one wide [B, sum C] matrix (all conditional-BN linears of a module in one launch)
    ld = scale.stride(0) if B > 1 else Cc
    if scale.dtype != torch.float32 or offset.dtype != torch.float32 or scale.stride(1) != 1 or offset.stride(1) != 1 or \
            (B > 1 and offset.stride(0) != ld):
        raise DgeError("cbn_affine: scale / offset must be float32 rows with unit inner stride and a common row stride")
    check(lib().dge_cbn_affine(C.c_void_p(scale.data_ptr()), C.c_void_p(offset.data_ptr()), ld, _f32(mean.contiguous()),
                               _f32(var.contiguous()), float(eps), _p(a), _p(b), B, Cc, _stream()), "dge_cbn_affine")
    return a, b


def slice_up(x, cout, up):
    B, H, W, Cin = x.shape
    f = 2 if up else 1
    y = torch.empty((B, H * f, W * f, cout), dtype=x.dtype, device=x.device)
    check(lib().dge_slice_up(_p(x), _p(y), B, H, W, Cin, cout, 1 if up else 0, dtype_of(x), _stream()), "dge_slice_up")
    return y


def attention(q, k, v):
    """q [B,N,D], k [B,M,D], v [B,M,DV] -> [B,N,DV] (softmax over the M keys, no scaling)"""
    B, N, D = q.shape
    M, DV = k.shape[1], v.shape[2]
    o = torch.empty((B, N, DV), dtype=q.dtype, device=q.device)
    check(lib().dge_attention(_p(q), _p(k), _p(v), _p(o), B, N, M, D, DV, dtype_of(q), _stream()), "dge_attention")
    return o


def maxpool2(x):
    B, H, W, Cc = x.shape
    y = torch.empty((B, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
    check(lib().dge_maxpool2(_p(x), _p(y), B, H, W, Cc, dtype_of(x), _stream()), "dge_maxpool2")
    return y


def rgb_tanh(x):
    B, H, W, Cc = x.shape
    img = torch.empty((B, 3, H, W), dtype=torch.float32, device=x.device)
    check(lib().dge_rgb_tanh(_p(x), _p(img), B, H * W, Cc, dtype_of(x), _stream()), "dge_rgb_tanh")
    return img


def sg1_in_bwd_coef(dots, sc, sh, style, npix):
    """-> (coef [B,C,3], gstyle [B,2C]) of the instance-norm + style_mod backward (dge_sg1_in_bwd_coef)."""
    B, Cc = sc.shape
    coef = torch.empty((B, Cc, 3), dtype=torch.float32, device=sc.device)
    gstyle = torch.empty((B, 2 * Cc), dtype=torch.float32, device=sc.device)
    check(lib().dge_sg1_in_bwd_coef(_f32(dots), _f32(sc), _f32(sh), _f32(style), _p(coef), _p(gstyle), B, Cc, int(npix), _stream()),
          "dge_sg1_in_bwd_coef")
    return coef, gstyle


def dot_stats(g, x):
    B, H, W, Cc = x.shape
    st = zeros((B, Cc, 2), x.device)
    check(lib().dge_dot_stats(_p(g), _p(x), _p(st), B, H * W, Cc, dtype_of(x), _stream()), "dge_dot_stats")
    return st


def nearest_up2_bwd(ghi, x=None):
    """Adjoint of the nearest x2 upsample; with x also the dot statistics (sum g*x, sum g) [B,C,2]."""
    B, H2, W2, Cc = ghi.shape
    glow = torch.empty((B, H2 // 2, W2 // 2, Cc), dtype=ghi.dtype, device=ghi.device)
    st = zeros((B, Cc, 2), ghi.device) if x is not None else None
    check(lib().dge_nearest_up2_bwd(_p(ghi), _p(x), _p(glow), _f32(st), B, H2 // 2, W2 // 2, Cc, dtype_of(ghi), _stream()),
          "dge_nearest_up2_bwd")
    return glow, st


def fromrgb_dgrad(gx, x0, w):
    """Gradient of FromRGB w.r.t. its input image -> [B,3,H,W] f32."""
    B, H, W, Cc = x0.shape
    gimg = torch.empty((B, 3, H, W), dtype=torch.float32, device=x0.device)
    check(lib().dge_fromrgb_dgrad(_p(gx), _p(x0), _f32(w.reshape(Cc, 3).contiguous()), _p(gimg), B, H * W, Cc, dtype_of(x0), _stream()),
          "dge_fromrgb_dgrad")
    return gimg


def nearest_up2(x, scale=1.0):
    B, H, W, Cc = x.shape
    y = torch.empty((B, 2 * H, 2 * W, Cc), dtype=x.dtype, device=x.device)
    check(lib().dge_nearest_up2(_p(x), _p(y), B, H, W, Cc, float(scale), dtype_of(x), _stream()), "dge_nearest_up2")
    return y


def linear_rows(x, ldx_b, row_xoff, w, bias, y, row_ybase, row_ybstride, B, wscale, bscale, add):
    R, K = w.shape
    check(lib().dge_linear_rows(_f32(x), int(ldx_b), _p(row_xoff), _f32(w), _f32(bias), _p(y), _p(row_ybase), _p(row_ybstride), B, R, K,
                                float(wscale), float(bscale), float(add), _stream()), "dge_linear_rows")
    return y


def demod_rows(s_all, wsq_cat, row_woff, row_sbase, row_cin, d_all, row_dbase, row_dbstride, B, eps):
    R = row_woff.numel()
    check(lib().dge_demod_rows(_f32(s_all), _f32(wsq_cat), _p(row_woff), _p(row_sbase), _p(row_cin), _p(d_all), _p(row_dbase),
                               _p(row_dbstride), B, R, float(eps), _stream()), "dge_demod_rows")
    return d_all


def pixelnorm_nhwc_bwd(gy, x, eps=1e-8):
    """x: [..., C] channel-last (NHWC activations, or [B,1,1,D] latent rows in f32)."""
    Cc = x.shape[-1]
    gx = torch.empty_like(x)
    check(lib().dge_pixelnorm_nhwc_bwd(_p(gy), _p(x), _p(gx), x.numel() // Cc, Cc, float(eps), dtype_of(x), _stream()), "dge_pixelnorm_nhwc_bwd")
    return gx


# ------------------------------------------------------------------ BigGAN-deep backward ops
def affine_relu_bwd(gu, x, a, b):
    """Backward of u = relu(a*x + b): -> (gx, stats [B,C,2] = (d/da, d/db))."""
    B, H, W, Cc = x.shape
    gx = torch.empty_like(x)
    st = zeros((B, Cc, 2), x.device)
    check(lib().dge_affine_relu_bwd(_p(gu), _p(x), _f32(a), _f32(b), _p(gx), _p(st), B, H * W, Cc, dtype_of(x), _stream()),
          "dge_affine_relu_bwd")
    return gx, st


def slice_up_bwd(gy, gx, up):
    """gx [B,H,W,Cin] += adjoint of slice_up applied to gy [B,H<<up,W<<up,Cout]."""
    B, H, W, Cin = gx.shape
    check(lib().dge_slice_up_bwd(_p(gy), _p(gx), B, H, W, Cin, gy.shape[3], 1 if up else 0, dtype_of(gx), _stream()), "dge_slice_up_bwd")
    return gx


def softmax_rows_(s):
    """in-place softmax over the last axis of a contiguous tensor"""
    M = s.shape[-1]
    check(lib().dge_softmax_rows(_p(s), s.numel() // M, M, dtype_of(s), _stream()), "dge_softmax_rows")
    return s


def softmax_rows_bwd_(p, gp):
    M = p.shape[-1]
    check(lib().dge_softmax_rows_bwd(_p(p), _p(gp), p.numel() // M, M, dtype_of(p), _stream()), "dge_softmax_rows_bwd")
    return gp


def rgb_tanh_bwd(gimg, img, Cc, dtype):
    B, _, H, W = img.shape
    gy = torch.empty((B, H, W, Cc), dtype=tdtype(dtype), device=img.device)
    check(lib().dge_rgb_tanh_bwd(_f32(gimg.contiguous()), _f32(img), _p(gy), B, H * W, Cc, dtype, _stream()), "dge_rgb_tanh_bwd")
    return gy


def maxpool2_bwd(gy, x, addend=None, relu=False):
    """gy [B,H/2,W/2,C], x [B,H,W,C] (the pooled tensor) -> gx [B,H,W,C] (+ addend); relu: times [x > 0] (the ReLU that made x)"""
    B, H, W, Cc = x.shape
    gx = torch.empty_like(x)
    fn = lib().dge_maxpool2_bwd_relu if relu else lib().dge_maxpool2_bwd
    check(fn(_p(gy), _p(x), _p(addend), _p(gx), B, H, W, Cc, dtype_of(x), _stream()), "dge_maxpool2_bwd")
    return gx


# ------------------------------------------------------------------ StyleGAN2 up layer at algorithmic cost
def upconv_supported(cin, cout, dtype):
    return bool(lib().dge_upconv_supported(int(cin), int(cout), int(dtype)))


def pack_upconv_weight(w, dtype=BF16, scale=1.0):
    """w: [Cout,Cin,3,3] f32 -> [9 (phase,tap) units, Cout, Cin] of `dtype` (dge_pack_upconv_weight)."""
    cout, cin = w.shape[0], w.shape[1]
    out = torch.empty((9, cout, cin), dtype=tdtype(dtype), device=w.device)
    check(lib().dge_pack_upconv_weight(_f32(w.detach().contiguous()), _p(out), cout, cin, float(scale), dtype, _stream()),
          "dge_pack_upconv_weight")
    return out


def upconv_fir(x, w_packed, cout, in_scale=None, out_scale=None, bias=None, bias_scale=1.0, noise=None, noise_w=None,
               act=ACT_NONE, gain=1.0):
    """x [B,H,W,Cin] -> y [B,2H,2W,cout]: conv_transpose2d(stride 2) + 4x4 FIR + noise / bias / activation in one kernel."""
    B, H, W, Cin = x.shape
    y = torch.empty((B, 2 * H, 2 * W, cout), dtype=x.dtype, device=x.device)
    nbs = 0 if (noise is None or noise.shape[0] == 1) else 4 * H * W
    nws = 0 if (noise_w is None or noise_w.numel() == 1) else 1
    def launch():
        check(lib().dge_upconv_fir(_p(x), _p(w_packed), _p(y), _f32(in_scale), _f32(out_scale), _f32(noise), nbs, _f32(noise_w), nws,
                                   _f32(bias), float(bias_scale), float(gain), int(act), B, H, W, Cin, cout, dtype_of(x), _stream()),
              "dge_upconv_fir")
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); launch(); e1.record()
        abytes = sum(t.numel() * t.element_size() for t in (x, y, w_packed))
        PROFILE.append((e0, e1, 2.0 * 9.0 * Cin * cout * H * W * B, (B, H, W, Cin, cout, 3, "upfir", False), abytes))
    else:
        launch()
    return y
