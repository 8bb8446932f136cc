"""StyleGAN2 (config F, 'skip' architecture) generator on the MI355X HIP kernels.

Drop-in surface of the reference's model/stylegan2_generator.py: same class names, constructor
arguments, forward signatures, returned dict keys and state_dict keys (165 keys at 1024), so
`generator.load_state_dict(torch.load(ckpt)['generator_smooth'])` works unchanged
(E_align_s2.py:50-55).  The math runs in libdge_hip.so:

  style / demodulation  -> dge_linear           (stylegan2_generator.py:825-829, :867-870)
  modulated 3x3 conv    -> dge_conv2d           (:855-922; up layers :879-896 folded, see csrc)
  toRGB + skip upsample -> dge_torgb            (:465-474, :515-522, :603-615)
  mapping / truncation  -> dge_pixelnorm, dge_linear, dge_truncation (:246-278, :311-333)

Activations are NHWC in `compute_dtype` ('bf16': bf16 storage + f32 MFMA accumulation; 'f32':
exact-f32 MFMA, the parity path).  Images and latents stay NCHW / row-major f32 like the
reference's tensors.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops

__all__ = ["StyleGAN2Generator"]

_RESOLUTIONS_ALLOWED = [8, 16, 32, 64, 128, 256, 512, 1024]
_INIT_RES = 4


def _dt(compute_dtype):
    if compute_dtype in ("bf16", ops.BF16):
        return ops.BF16
    if compute_dtype in ("f32", "fp32", ops.F32):
        return ops.F32
    raise ValueError(f"compute_dtype must be 'bf16' or 'f32', got {compute_dtype!r}")


import os as _os
_PP_GEN = not _os.environ.get("DGE_NO_PP_GEN")
_DENSE_CHAIN = _os.environ.get("DGE_DENSE_CHAIN") == "1"
_UP_PP = _os.environ.get("DGE_UP_PP", "1") != "0"      # round 6: the default for the Cin >= 128 up layers (dge_up_pp: up_s4 / up_pp kernels; DGE_UP_PP=0: upconv_fir)


class DenseBlock(nn.Module):
    """Equalised-lr dense layer (reference :925-996)."""

    def __init__(self, in_channels, out_channels, additional_bias=0.0, lr_mul=1.0, activation_type="lrelu"):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels) / lr_mul)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.wscale = lr_mul / math.sqrt(in_channels)
        self.bscale = lr_mul
        self.additional_bias = additional_bias
        if activation_type not in ("linear", "lrelu"):
            raise NotImplementedError(f"Not implemented activation function: `{activation_type}`!")
        self.act = ops.ACT_LRELU if activation_type == "lrelu" else ops.ACT_NONE
        self.gain = math.sqrt(2.0) if activation_type == "lrelu" else 1.0

    def forward(self, x, out=None):
        if x.ndim != 2:
            x = x.reshape(x.shape[0], -1)
        return ops.linear(x, self.weight, self.bias, self.wscale, self.bscale, self.additional_bias, self.act,
                          self.gain, out=out)


class MappingModule(nn.Module):
    """z -> w: pixel norm + 8 dense layers (reference :199-278)."""

    def __init__(self, input_space_dim=512, hidden_space_dim=512, final_space_dim=512, num_layers=8, lr_mul=0.01):
        super().__init__()
        self.input_space_dim = input_space_dim
        self.num_layers = num_layers
        for i in range(num_layers):
            cin = input_space_dim if i == 0 else hidden_space_dim
            cout = final_space_dim if i == num_layers - 1 else hidden_space_dim
            self.add_module(f"dense{i}", DenseBlock(cin, cout, lr_mul=lr_mul))

    def forward(self, z, label=None):
        if z.ndim != 2 or z.shape[1] != self.input_space_dim:
            raise ValueError(f"Input latent code should be with shape [batch_size, input_dim], where `input_dim` "
                             f"equals to {self.input_space_dim}!\nBut `{z.shape}` is received!")
        zn = ops.pixelnorm(z.float().contiguous())
        layers = [getattr(self, f"dense{i}") for i in range(self.num_layers)]
        if _DENSE_CHAIN and self.num_layers <= 8 and all(max(L.weight.shape) <= 1024 for L in layers):
            # the 8 dense layers in one launch, bit-identical to the per-layer calls.  OFF by default: one workgroup per sample
            # leaves a 256-CU part idle behind a latency-bound chain (round 4: +0.8 ms per mapping pass at batch 8 against
            # ~45 us for the eight chip-wide launches it replaces); DGE_DENSE_CHAIN=1 turns it on for launch-bound hosts at batch 1
            w = ops.dense_chain(zn, layers)
        else:
            w = zn
            for L in layers:
                w = L(w)
        return {"z": zn, "label": label, "w": w}


class TruncationModule(nn.Module):
    """Truncation trick towards w_avg (reference :281-333)."""

    def __init__(self, w_space_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        self.w_space_dim = w_space_dim
        self.register_buffer("w_avg", torch.zeros(w_space_dim))

    def forward(self, w, trunc_psi=None, trunc_layers=None):
        if w.ndim == 2:
            assert w.shape[1] == self.w_space_dim
        else:
            assert w.ndim == 3 and tuple(w.shape[1:]) == (self.num_layers, self.w_space_dim)
        psi = 1.0 if trunc_psi is None else trunc_psi
        layers = 0 if trunc_layers is None else trunc_layers
        return ops.truncation(w, self.w_avg, self.num_layers, psi, layers)


class UpsamplingLayer(nn.Module):
    """Holds the FIR buffer for state_dict compatibility (`filter.kernel`, `upsample.kernel`);
    the filtering itself is fused into dge_conv2d (up layers) and dge_torgb (skip branch)."""

    def __init__(self, gain=2.0):
        super().__init__()
        k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32)
        k = k / k.sum() * (gain ** 2)
        self.register_buffer("kernel", torch.from_numpy(k[None, None]))


class InputBlock(nn.Module):
    def __init__(self, init_resolution, channels):
        super().__init__()
        self.const = nn.Parameter(torch.randn(1, channels, init_resolution, init_resolution))

    def forward(self, w):
        return self.const.repeat(w.shape[0], 1, 1, 1)


class ModulateConvBlock(nn.Module):
    """Modulated convolution block (reference :742-922), shared-weight formulation."""

    def __init__(self, in_channels, out_channels, resolution, w_space_dim, kernel_size=3, scale_factor=1,
                 demodulate=True, add_noise=True, activation_type="lrelu", epsilon=1e-8):
        super().__init__()
        self.res, self.in_c, self.out_c, self.ksize, self.eps = resolution, in_channels, out_channels, kernel_size, epsilon
        self.up = scale_factor > 1
        if self.up:
            self.filter = UpsamplingLayer(gain=scale_factor)
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, kernel_size, kernel_size))
        self.wscale = 1.0 / math.sqrt(kernel_size * kernel_size * in_channels)
        self.style = DenseBlock(w_space_dim, in_channels, additional_bias=1.0, activation_type="linear")
        self.demodulate = demodulate
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.bscale = 1.0
        if activation_type not in ("linear", "lrelu"):
            raise NotImplementedError(f"Not implemented activation function: `{activation_type}`!")
        self.act = ops.ACT_LRELU if activation_type == "lrelu" else ops.ACT_NONE
        self.gain = math.sqrt(2.0) if activation_type == "lrelu" else 1.0
        self.add_noise = add_noise
        if add_noise:
            self.register_buffer("noise", torch.randn(1, 1, resolution, resolution))
            self.noise_strength = nn.Parameter(torch.zeros(()))
        self._cache = {}

    # -- derived weights, rebuilt only when the parameter changes ---------------------------
    def _prepared(self, dtype):
        key = (dtype, self.weight._version, self.weight.data_ptr(), getattr(self.weight, "_dge_gen", 0))
        c = self._cache.get("w")
        if c is None or c[0] != key:
            mode = ops.PACK_UPFOLD if self.up else ops.PACK_FWD
            hin = self.res // 2 if self.up else self.res         # the low-resolution layers get fragment-ordered weights (conv_small)
            packed = ops.pack_conv_weight(self.weight, ops.pack_mode_for(self.weight, mode, hin, hin, dtype), dtype, self.wscale) \
                if self.ksize == 3 else None
            wsq = ops.weight_sumsq(self.weight, self.wscale) if self.demodulate else None
            c = (key, packed, wsq)
            self._cache["w"] = c
        return c[1], c[2]

    def _prepared_up(self, dtype):
        """[9 units][Cout][Cin] weights of the phase-form up kernel (ops.upconv_fir), or None when the layer shape is not
        covered by it (then the folded 3x3-per-phase form of conv2d(up=True) runs)."""
        import os
        # (input resolutions below 16: a single 16x16 t-pixel tile per sample would be mostly padding -- the folded form is faster)
        if (not self.up or self.res < 32 or os.environ.get("DGE_UP_FOLDED") == "1"
                or not ops.upconv_supported(self.in_c, self.out_c, dtype)):
            return None
        key = (dtype, self.weight._version, self.weight.data_ptr(), getattr(self.weight, "_dge_gen", 0))
        c = self._cache.get("wu")
        if c is None or c[0] != key:
            c = (key, ops.pack_upconv_weight(self.weight, dtype, self.wscale))
            self._cache["wu"] = c
        return c[1]

    def conv(self, x, s, d, noise, dt, rgb=None):
        """The modulated conv proper (:898-921) on NHWC activations: shared-weight form with s / d as prologue / epilogue scales.
        `rgb`: fused toRGB of the result (ops.conv2d), stride-1 layers only."""
        nw = self.noise_strength.detach().reshape(1) if noise is not None else None
        wu = self._prepared_up(dt)
        if wu is not None:
            assert rgb is None
            B, H, W, _ = x.shape
            if (_UP_PP and s is not None and d is not None and (noise is None or nw.numel() == 1)
                    and ops.up_pp_supported(B, H, W, self.in_c, self.out_c, dt)):
                # MFMA-bound up layers (Cin >= 128): fused modulation (:858-875) folded into one weight image per sample, ping-pong
                # implicit GEMM with the FIR in registers (csrc/up_pp.hip)
                wimg = ops.pack_up_pp(wu, self.out_c, self.in_c, in_scale=s, out_scale=d, gain=self.gain)
                return ops.up_pp(x, wimg, self.out_c, bias=self.bias, bias_scale=self.bscale, noise=noise, noise_w=nw, act=self.act,
                                 gain=self.gain)
            return ops.upconv_fir(x, wu, self.out_c, in_scale=s, out_scale=d, bias=self.bias, bias_scale=self.bscale,
                                  noise=noise, noise_w=nw, act=self.act, gain=self.gain)
        B, H, W, _ = x.shape
        if (rgb is None and not self.up and d is not None and s is not None and self.ksize == 3 and _PP_GEN
                and ops.conv_pp_supported(B, H, W, self.in_c, self.out_c, dt)):
            # MFMA-bound layers (>= 128 channels at 64^2 .. 256^2): the reference's fused modulation (:858-875) - style, demodulation
            # and gain folded into one weight image per sample - feeding the ping-pong implicit GEMM (csrc/conv_pp.hip)
            wpp = ops.pack_conv_pp(self.weight, self.wscale, in_scale=s, out_scale=d, gain=self.gain)
            return ops.conv_pp(x, wpp, self.out_c, bias=self.bias, bias_scale=self.bscale, noise=noise, noise_w=nw, act=self.act,
                               gain=self.gain)
        packed, _ = self._prepared(dt)
        return ops.conv2d(x, packed, self.out_c, 3, up=self.up, in_scale=s, out_scale=d, bias=self.bias,
                          bias_scale=self.bscale, noise=noise, noise_w=nw, act=self.act, gain=self.gain, rgb=rgb)

    def styles(self, w):
        """style s[b,i] and demodulation d[b,o] for latent rows w [B, 512] (any row stride)."""
        s = self.style(w)
        d = None
        if self.demodulate:
            _, wsq = self._prepared(ops.BF16)  # wsq is dtype independent; key includes dtype only for packing
            d = ops.linear(s, wsq, None, 1.0, 1.0, self.eps, ops.LIN_RSQRT, 1.0, square_input=True)
        return s, d

    def forward(self, x, w, randomize_noise=False, prev_image=None):
        """x: NHWC activations.  3x3 blocks return (y NHWC, style); the 1x1 toRGB block returns
        (image NCHW f32 [+ upsampled prev_image], style)."""
        dt = ops.dtype_of(x)
        if self.ksize == 1:
            s = self.style(w)
            return ops.torgb(x, self.weight, s, self.bias, prev_image, self.wscale), s
        packed, wsq = self._prepared(dt)
        s = self.style(w)
        d = ops.linear(s, wsq, None, 1.0, 1.0, self.eps, ops.LIN_RSQRT, 1.0, square_input=True) if self.demodulate else None
        noise = None
        if self.add_noise:
            if randomize_noise:
                noise = ops.randn((x.shape[0], self.res, self.res), x.device)
            else:
                noise = self.noise.reshape(1, self.res, self.res)
        return self.conv(x, s, d, noise, dt), s


class SynthesisModule(nn.Module):
    """wp -> image (reference :336-539, architecture 'skip')."""

    def __init__(self, resolution=1024, init_resolution=4, w_space_dim=512, image_channels=3,
                 fmaps_base=32 << 10, fmaps_max=512, compute_dtype="bf16"):
        super().__init__()
        if image_channels != 3:
            raise ValueError("only image_channels=3 is supported")
        self.init_res, self.resolution, self.w_space_dim = init_resolution, resolution, w_space_dim
        self.init_res_log2, self.final_res_log2 = int(np.log2(init_resolution)), int(np.log2(resolution))
        self.fmaps_base, self.fmaps_max = fmaps_base, fmaps_max
        self.num_layers = (self.final_res_log2 - self.init_res_log2 + 1) * 2
        self.compute_dtype = compute_dtype
        for res_log2 in range(self.init_res_log2, self.final_res_log2 + 1):
            res = 2 ** res_log2
            blk = res_log2 - self.init_res_log2
            if res == init_resolution:
                self.add_module("early_layer", InputBlock(init_resolution, self.get_nf(res)))
            else:
                self.add_module(f"layer{2 * blk - 1}", ModulateConvBlock(self.get_nf(res // 2), self.get_nf(res), res,
                                                                        w_space_dim, scale_factor=2))
            self.add_module(f"layer{2 * blk}", ModulateConvBlock(self.get_nf(res), self.get_nf(res), res, w_space_dim))
            self.add_module(f"output{blk}", ModulateConvBlock(self.get_nf(res), image_channels, res, w_space_dim,
                                                             kernel_size=1, demodulate=False, add_noise=False,
                                                             activation_type="linear"))
        self.upsample = UpsamplingLayer(gain=2.0)

    def get_nf(self, res):
        return min(self.fmaps_base // res, self.fmaps_max)

    def forward(self, wp, randomize_noise=False):
        if wp.ndim != 3 or tuple(wp.shape[1:]) != (self.num_layers, self.w_space_dim):
            raise ValueError(f"Input tensor should be with shape [batch_size, num_layers, w_space_dim], where "
                             f"`num_layers` equals to {self.num_layers}, and `w_space_dim` equals to "
                             f"{self.w_space_dim}!\nBut `{wp.shape}` is received!")
        from .autograd_s2 import synthesis_forward
        return synthesis_forward(self, wp, randomize_noise)


def mixing_mask(num_layers, style_mixing_prob=0.9):
    """Host side of the train-mode style mixing (reference :183-191), same np.random draw order: returns the [L] float
    mask (CPU) that `StyleGAN2Generator.forward(..., mix_mask=...)` applies on the device."""
    m = torch.zeros(num_layers, dtype=torch.float32)
    if np.random.uniform() < style_mixing_prob:
        m[:np.random.randint(1, num_layers)] = 1.0
    return m


class StyleGAN2Generator(nn.Module):
    """Reference :35-196.  Extra keyword `compute_dtype` selects 'bf16' (default) or 'f32'."""

    def __init__(self, resolution, z_space_dim=512, w_space_dim=512, label_size=0, mapping_layers=8,
                 mapping_fmaps=512, mapping_lr_mul=0.01, repeat_w=True, image_channels=3, final_tanh=False,
                 const_input=True, architecture="skip", fused_modulate=True, demodulate=True, use_wscale=True,
                 fmaps_base=32 << 10, fmaps_max=512, compute_dtype="bf16"):
        super().__init__()
        if resolution not in _RESOLUTIONS_ALLOWED:
            raise ValueError(f"Invalid resolution: `{resolution}`!\nResolutions allowed: {_RESOLUTIONS_ALLOWED}.")
        if architecture != "skip" or label_size or not repeat_w or final_tanh or not const_input \
                or not demodulate or not use_wscale:
            raise ValueError("only the configuration the released checkpoints use is implemented: architecture="
                             "'skip', label_size=0, repeat_w, const_input, demodulate, use_wscale, no final tanh")
        _dt(compute_dtype)
        self.resolution, self.z_space_dim, self.w_space_dim = resolution, z_space_dim, w_space_dim
        self.num_layers = int(np.log2(resolution // _INIT_RES * 2)) * 2
        self.mapping = MappingModule(z_space_dim, mapping_fmaps, w_space_dim, mapping_layers, mapping_lr_mul)
        self.truncation = TruncationModule(w_space_dim, self.num_layers)
        self.synthesis = SynthesisModule(resolution, _INIT_RES, w_space_dim, image_channels, fmaps_base, fmaps_max,
                                         compute_dtype)

    def forward(self, z, label=None, w_moving_decay=0.995, style_mixing_prob=0.9, trunc_psi=None, trunc_layers=None,
                randomize_noise=False, **_unused_kwargs):
        """Extra keywords: `mix_mask` (device [L] mask, hipGraph form of the style mixing) and `new_z` (the second latent of the
        style mixing, reference :187 `torch.randn_like(z)`; default: drawn by the counter-based device generator, which under
        data parallelism yields the rank's rows of the global-batch draw)."""
        if z.requires_grad and torch.is_grad_enabled():
            # the reference lets autograd run through the mapping network (baseline_utils/image2stylegan_w2z_opW.py optimises z);
            # the HIP mapping has no backward: refuse loudly instead of returning a silently constant w
            raise ops.DgeError("StyleGAN2Generator.forward: d/dz through the mapping network is not implemented on the HIP path; "
                               "optimise w / wp (synthesis has a hand-written data gradient) or call under torch.no_grad()")
        with torch.no_grad():
            mapping_results = self.mapping(z, label)
            w = mapping_results["w"]
            # train-mode side effects the reference keeps active during E_align
            if self.training and w_moving_decay < 1:
                batch_w_avg = w.mean(dim=0)
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                    # the upstream code all_gather'ed w here (reference :178, commented out): global batch mean
                    from .e_align import _all_reduce
                    _all_reduce(batch_w_avg)
                    batch_w_avg = batch_w_avg / dist.get_world_size()
                self.truncation.w_avg.copy_(self.truncation.w_avg * w_moving_decay + batch_w_avg * (1 - w_moving_decay))
            if self.training and style_mixing_prob > 0:
                new_z = _unused_kwargs.get("new_z")
                if new_z is None:
                    new_z = ops.randn(tuple(z.shape), z.device)
                new_w = self.mapping(new_z, label)["w"]
                mix_mask = _unused_kwargs.get("mix_mask")
                if mix_mask is not None:
                    # hipGraph-friendly form of the same mixing: the host decision (uniform < prob, cutoff) arrives as a
                    # device mask [L] (1 for layers below the cutoff, all zero when this step does not mix), so the
                    # kernel sequence of a step is identical for every outcome (see EAlignStep.capture / mixing_mask)
                    wt, nwt = self.truncation(w), self.truncation(new_w)
                    w = wt + mix_mask.view(1, -1, 1) * (nwt - wt)
                elif np.random.uniform() < style_mixing_prob:
                    mixing_cutoff = np.random.randint(1, self.num_layers)
                    w = self.truncation(w)
                    new_w = self.truncation(new_w)
                    w[:, :mixing_cutoff] = new_w[:, :mixing_cutoff]
            wp = self.truncation(w, trunc_psi, trunc_layers)
        synthesis_results = self.synthesis(wp, randomize_noise)
        return {**mapping_results, **synthesis_results}
