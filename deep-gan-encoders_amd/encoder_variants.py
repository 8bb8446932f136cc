"""Encoder variants of the reference on the HIP kernels (forward):

* `E_Blur.BE`  (model/E/E_Blur.py:16-134, "case 2", used by embedding_img.py): E.BE plus a depthwise
  blur before conv_2 and, for block resolutions >= 128, a stride-2 conv_2 with `transform_kernel`
  (model/utils/lreq.py:145-147).  That 4x4 stride-2 kernel 0.25*sum-of-4-shifts is algebraically
  conv3x3 followed by avg_pool2d(2) (zero padding included), so it runs as conv3x3 + the pooling
  blend; noise / bias / leaky_relu are then applied at the HALF resolution, as in the reference.
* `E_PG.BE`    (model/E/E_PG.py:39-164, PGGAN encoder; forward + backward in autograd_encpg.py): IN -> conv -> noise -> bias -> lrelu -> IN ->
  conv -> noise -> bias -> (+ affine-IN(conv1x1(residual))) -> lrelu -> avgpool; FC head.
  The reference returns (tensor(0), tensor(0)); the evident intent is implemented:
  `(tensor(0), new_final(x.view(B, -1)))`, and `trunk()` exposes the activation before the head.

State_dict keys match the reference (110 keys for E_Blur 1024/16/9, 57 for E_PG 256/64/7).
"""
import numpy as np
import torch
from torch import nn

from . import lreq as ln
from . import ops
from .autograd_enc import _packed, draw_noises
from .encoder import FromRGB
from .stylegan1 import Blur
from .stylegan2_generator import _dt


# ----------------------------------------------------------------------------------- E_Blur
class BlurBEBlock(nn.Module):
    def __init__(self, inputs, outputs, latent_size, has_last_conv=True, fused_scale=True):
        super().__init__()
        self.has_last_conv, self.fused_scale, self.inputs, self.outputs = has_last_conv, fused_scale, inputs, outputs
        self.noise_weight_1 = nn.Parameter(torch.zeros(1, inputs, 1, 1))
        self.bias_1 = nn.Parameter(torch.zeros(1, inputs, 1, 1))
        self.inver_mod1 = ln.Linear(2 * inputs, latent_size, gain=1)
        self.conv_1 = ln.Conv2d(inputs, inputs, 3, 1, 1, bias=False)
        self.noise_weight_2 = nn.Parameter(torch.zeros(1, outputs, 1, 1))
        self.bias_2 = nn.Parameter(torch.zeros(1, outputs, 1, 1))
        self.inver_mod2 = ln.Linear(2 * inputs, latent_size, gain=1)
        self.blur = Blur(inputs)
        if has_last_conv:
            self.conv_2 = ln.Conv2d(inputs, outputs, 3, 1, 1, bias=False)     # stride 2 is realised as conv + pool (see module doc)
        if inputs != outputs:
            self.conv_3 = ln.Conv2d(inputs, outputs, 1, 1, 0)


class BlurBE(nn.Module):
    """E_Blur.BE"""

    def __init__(self, startf=16, maxf=512, layer_count=9, latent_size=512, channels=3, compute_dtype="bf16"):
        super().__init__()
        _dt(compute_dtype)
        self.maxf, self.startf, self.latent_size, self.layer_count, self.compute_dtype = maxf, startf, latent_size, layer_count, compute_dtype
        self.decode_block = nn.ModuleList()
        self.FromRGB = FromRGB(channels, startf)
        inputs, outputs, resolution = startf, startf * 2, 1024
        for i in range(layer_count):
            self.decode_block.append(BlurBEBlock(inputs, outputs, latent_size, i + 1 != layer_count, fused_scale=resolution >= 128))
            inputs, outputs = min(maxf, inputs * 2), min(maxf, outputs * 2)
            resolution /= 2

    def forward(self, img, block_num=9, noises=None):
        """img [B,3,R,R] -> (x [B,C,4,4], w [B,2*layer_count,512]); differentiable w.r.t. the parameters AND the image
        (autograd_encblur: embedding_img.py:86-127 back-propagates through both outputs and through the input)."""
        if block_num != 9:
            raise ValueError("progressive block_num != 9 is not used by the reference's scripts")
        from .autograd_encblur import BlurEncoderFunction
        return BlurEncoderFunction.apply(self, img, noises, *list(self.parameters()))


# ----------------------------------------------------------------------------------- E_PG
class _AffineIN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class PGBEBlock(nn.Module):
    def __init__(self, inputs, outputs, latent_size, has_second_conv=True):
        super().__init__()
        self.has_second_conv, self.inputs, self.outputs = has_second_conv, inputs, outputs
        self.noise_weight_1 = nn.Parameter(torch.zeros(1, inputs, 1, 1))
        self.bias_1 = nn.Parameter(torch.zeros(1, inputs, 1, 1))
        self.conv_1 = ln.Conv2d(inputs, inputs, 3, 1, 1, bias=False)
        self.noise_weight_2 = nn.Parameter(torch.zeros(1, outputs, 1, 1))
        self.bias_2 = nn.Parameter(torch.zeros(1, outputs, 1, 1))
        if has_second_conv:
            self.conv_2 = ln.Conv2d(inputs, outputs, 3, 1, 1, bias=False)
        if inputs != outputs:
            self.conv_3 = ln.Conv2d(inputs, outputs, 1, 1, 0)
            self.instance_norm_3 = _AffineIN(outputs)


class PGBE(nn.Module):
    """E_PG.BE"""

    def __init__(self, startf=16, maxf=512, layer_count=9, latent_size=512, channels=3, pggan=False, compute_dtype="bf16"):
        super().__init__()
        _dt(compute_dtype)
        self.maxf, self.startf, self.latent_size, self.layer_count, self.compute_dtype = maxf, startf, latent_size, layer_count, compute_dtype
        self.decode_block = nn.ModuleList()
        self.FromRGB = FromRGB(channels, startf)
        inputs, outputs = startf, startf * 2
        for i in range(layer_count):
            self.decode_block.append(PGBEBlock(inputs, outputs, latent_size, i + 1 != layer_count))
            inputs, outputs = min(maxf, inputs * 2), min(maxf, outputs * 2)
        self.pggan = pggan
        if pggan:
            self.new_final = ln.Linear(512 * 16, latent_size, gain=1)

    @torch.no_grad()
    def trunk(self, img, noises=None):
        """Activation [B,C,4,4] (NCHW f32) after the last block - what the reference computes and then discards."""
        from .autograd_encpg import pg_encoder_forward
        return pg_encoder_forward(self, img, noises, save=False)[0]

    def forward(self, img, block_num=9, noises=None):
        """-> (tensor(0), z): z = new_final(trunk) is differentiable w.r.t. every parameter (autograd_encpg)."""
        if block_num != 9:
            raise ValueError("progressive block_num != 9 is not used by the reference's scripts")
        if not self.pggan:
            self.trunk(img, noises)
            return torch.tensor(0), torch.tensor(0)
        from .autograd_encpg import PGEncoderFunction
        return torch.tensor(0), PGEncoderFunction.apply(self, img, noises, *list(self.parameters()))


# ----------------------------------------------------------------------------------- E_BIG
class _PlainConv(nn.Module):
    """torch.nn.Conv2d(channels, outputs, 1) parameter holder (E_BIG's FromRGB is not an lreq layer, E_BIG.py:84-92)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(cout, cin, 1, 1) * (1.0 / cin) ** 0.5)
        self.bias = nn.Parameter(torch.zeros(cout))


class _FromRGBPlain(nn.Module):
    def __init__(self, channels, outputs):
        super().__init__()
        self.from_rgb = _PlainConv(channels, outputs)


class BigBEBlock(nn.Module):
    def __init__(self, inputs, outputs, latent_size, has_second_conv=True):
        super().__init__()
        from .biggan_generator import BigGANBatchNorm
        self.has_second_conv, self.inputs, self.outputs = has_second_conv, inputs, outputs
        self.noise_weight_1 = nn.Parameter(torch.zeros(1, inputs, 1, 1))
        self.bias_1 = nn.Parameter(torch.zeros(1, inputs, 1, 1))
        self.batch_norm_1 = BigGANBatchNorm(inputs, condition_vector_dim=256, n_stats=51, eps=1e-12, conditional=True)
        self.conv_1 = ln.Conv2d(inputs, inputs, 3, 1, 1, bias=False)
        self.noise_weight_2 = nn.Parameter(torch.zeros(1, outputs, 1, 1))
        self.bias_2 = nn.Parameter(torch.zeros(1, outputs, 1, 1))
        self.batch_norm_2 = BigGANBatchNorm(inputs, condition_vector_dim=256, n_stats=51, eps=1e-12, conditional=True)
        if has_second_conv:
            self.conv_2 = ln.Conv2d(inputs, outputs, 3, 1, 1, bias=False)
        if inputs != outputs:
            self.batch_norm_3 = BigGANBatchNorm(inputs, condition_vector_dim=256, n_stats=51, eps=1e-12, conditional=True)
            self.conv_3 = ln.Conv2d(inputs, outputs, 1, 1, 0)


class BigBE(nn.Module):
    """E_BIG.BE (model/E/E_BIG.py:93-227): conditional-BN encoder for BigGAN; forward(x, cond_vector) -> (c_v [B,256], z [B,128])."""

    def __init__(self, startf=16, maxf=512, layer_count=9, latent_size=512, channels=3, pggan=False, biggan=False, compute_dtype="bf16"):
        super().__init__()
        _dt(compute_dtype)
        self.maxf, self.startf, self.latent_size, self.layer_count, self.compute_dtype = maxf, startf, latent_size, layer_count, compute_dtype
        self.decode_block = nn.ModuleList()
        self.FromRGB = _FromRGBPlain(channels, startf)
        inputs, outputs = startf, startf * 2
        for i in range(layer_count):
            self.decode_block.append(BigBEBlock(inputs, outputs, latent_size, i + 1 != layer_count))
            inputs, outputs = min(maxf, inputs * 2), min(maxf, outputs * 2)
        self.biggan = biggan
        if biggan:
            self.new_final_1 = ln.Linear(8192, 256, gain=1)
            self.new_final_2 = ln.Linear(256, 128, gain=1)

    @torch.no_grad()
    def trunk(self, img, cond_vector, noises=None, truncation=0.4):
        from .autograd_encbig import big_encoder_forward
        return big_encoder_forward(self, img, cond_vector, noises, save=False, truncation=truncation)[0]

    def forward(self, img, cond_vector, block_num=9, noises=None):
        """-> (c_v [B,256], z [B,128]), differentiable w.r.t. every parameter (autograd_encbig)."""
        if not self.biggan:
            raise RuntimeError("E_BIG.BE.forward needs biggan=True (the reference raises UnboundLocalError otherwise, E_BIG.py:223-227)")
        from .autograd_encbig import BigEncoderFunction
        return BigEncoderFunction.apply(self, img, cond_vector, noises, *list(self.parameters()))
