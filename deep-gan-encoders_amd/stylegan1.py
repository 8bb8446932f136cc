"""StyleGAN1 generator + mapping (reference model/stylegan1/net.py: DecodeBlock :110-169,
Generator :256-363, Mapping :441-466, Blur :48-58, style_mod :32-34) on the HIP kernels.
Same class names, constructor arguments, forward signatures and state_dict keys (91 keys for
Cat-256, 117 for FFHQ-1024, 16 for the mapping; `buffer1` is a plain attribute as in the
reference), so `Gs.load_state_dict(torch.load('Gs_dict.pth'))` works unchanged
(E_align_s2.py:29-35).

Per block: [upscale2d + conv3x3 | ConvTranspose2d(3,s2,p1)+transform_kernel] -> blur -> +noise
-> +bias -> lrelu -> instance norm -> style_mod -> conv3x3 -> +noise -> +bias -> lrelu ->
instance norm -> style_mod.  Instance norm + style_mod collapse into one per-(b,c) affine that
is fused into the prologue of the next convolution (dge_affine_compose + dge_conv2d), the
nearest upsample is fused into the conv's read, the fused-scale transposed conv runs as a
phase-folded implicit GEMM with a depth-to-space store.
"""
import numpy as np
import torch
from torch import nn

from . import lreq as ln
from . import ops
from .stylegan2_generator import _dt


class Blur(nn.Module):
    def __init__(self, channels):
        super().__init__()
        f = np.array([1, 2, 1], dtype=np.float32)
        f = f[:, None] * f[None, :]
        f /= f.sum()
        self.register_buffer("weight", torch.tensor(f).view(1, 1, 3, 3).repeat(channels, 1, 1, 1))


class _ConvT(nn.Module):
    """ln.ConvTranspose2d(inputs, outputs, 3, 2, 1, bias=False, transform_kernel=True): weight [in,out,3,3]"""

    def __init__(self, inputs, outputs):
        super().__init__()
        self.std = np.sqrt(2.0) / np.sqrt(9 * inputs)
        self.weight = nn.Parameter(torch.randn(inputs, outputs, 3, 3) * self.std)
        setattr(self.weight, "lr_equalization_coef", self.std)


class DecodeBlock(nn.Module):
    def __init__(self, inputs, outputs, latent_size, has_first_conv=True, fused_scale=True):
        super().__init__()
        self.has_first_conv, self.fused_scale, self.inputs, self.outputs = has_first_conv, fused_scale, inputs, outputs
        if has_first_conv:
            self.conv_1 = _ConvT(inputs, outputs) if fused_scale else ln.Conv2d(inputs, outputs, 3, 1, 1, bias=False)
        self.blur = Blur(outputs)
        self.noise_weight_1 = nn.Parameter(torch.zeros(1, outputs, 1, 1))
        self.bias_1 = nn.Parameter(torch.zeros(1, outputs, 1, 1))
        self.style_1 = ln.Linear(latent_size, 2 * outputs, gain=1)
        self.conv_2 = ln.Conv2d(outputs, outputs, 3, 1, 1, bias=False)
        self.noise_weight_2 = nn.Parameter(torch.zeros(1, outputs, 1, 1))
        self.bias_2 = nn.Parameter(torch.zeros(1, outputs, 1, 1))
        self.style_2 = ln.Linear(latent_size, 2 * outputs, gain=1)
        self._cache = {}

    def _packed(self, conv, dtype, mode, hw=None):
        """`hw`: resolution a stride-1 3x3 conv reads this copy at (the low-resolution blocks keep theirs in fragment order for
        csrc/conv_small.hip)"""
        w = conv.weight
        if hw is not None:
            mode = ops.pack_mode_for(w, mode, hw, hw, dtype)
        key = (id(conv), mode, dtype)
        ver = (w._version, w.data_ptr(), getattr(w, "_dge_gen", 0))
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, ops.pack_conv_weight(w, mode, dtype, 1.0))
            self._cache[key] = hit
        return hit[1]


class ToRGB(nn.Module):
    def __init__(self, inputs, channels):
        super().__init__()
        self.to_rgb = ln.Conv2d(inputs, channels, 1, 1, 0, gain=1)


class Generator(nn.Module):
    def __init__(self, startf=32, maxf=256, layer_count=3, latent_size=128, channels=3, compute_dtype="bf16"):
        super().__init__()
        if channels != 3:
            raise ValueError("channels=3 only")
        _dt(compute_dtype)
        self.maxf, self.startf, self.layer_count, self.channels, self.latent_size = maxf, startf, layer_count, channels, latent_size
        self.compute_dtype = compute_dtype
        mul = 2 ** (layer_count - 1)
        inputs = min(maxf, startf * mul)
        self.const = nn.Parameter(torch.ones(1, inputs, 4, 4))
        self.layer_to_resolution = [0] * layer_count
        resolution = 2
        to_rgb = nn.ModuleList()
        self.decode_block = nn.ModuleList()
        for i in range(layer_count):
            outputs = min(maxf, startf * mul)
            block = DecodeBlock(inputs, outputs, latent_size, i != 0, fused_scale=resolution * 2 >= 128)
            resolution *= 2
            self.layer_to_resolution[i] = resolution
            to_rgb.append(ToRGB(outputs, channels))
            self.decode_block.append(block)
            inputs = outputs
            mul //= 2
        self.to_rgb = to_rgb

    def decode(self, styles, lod, noise=0, noises=None):
        """styles [B, 2*layer_count, latent]; `noises`: optional list of N(0,1) tensors in the
        reference's draw order (2 per block; the very first has batch 1 because the block input
        is the batch-1 const, net.py:148).  Differentiable w.r.t. `styles` (autograd_sg1)."""
        from .autograd_sg1 import DecodeFunction
        return DecodeFunction.apply(self, styles, lod, noises)

    def forward(self, styles, lod, blend=1, remove_blob=False, noises=None):
        if remove_blob or blend != 1:
            raise NotImplementedError("only the decode() path (blend == 1, remove_blob=False) is on the E_align hot path")
        return self.decode(styles, lod, 1, noises=noises)


class MappingBlock(nn.Module):
    def __init__(self, inputs, output, lrmul=0.01):
        super().__init__()
        self.fc = ln.Linear(inputs, output, lrmul=lrmul)


class Mapping(nn.Module):
    def __init__(self, num_layers=18, mapping_layers=8, latent_size=512, dlatent_size=512, mapping_fmaps=512, trunc_tensor=None):
        super().__init__()
        inputs = latent_size
        self.mapping_layers, self.num_layers = mapping_layers, num_layers
        for i in range(mapping_layers):
            outputs = dlatent_size if i == mapping_layers - 1 else mapping_fmaps
            setattr(self, "block_%d" % (i + 1), MappingBlock(inputs, outputs, lrmul=0.01))
            inputs = outputs
        self.buffer1 = trunc_tensor          # plain attribute: not part of the state_dict (reference :452)

    def forward(self, z, coefs_m=0):
        with torch.no_grad():
            dev = self.block_1.fc.weight.device
            x = ops.pixelnorm(z.to(dev).float().contiguous())
            for i in range(self.mapping_layers):
                fc = getattr(self, "block_%d" % (i + 1)).fc
                x = ops.linear(x, fc.weight.detach(), fc.bias.detach(), act=ops.ACT_LRELU)
            if self.buffer1 is None:
                return x.view(x.shape[0], 1, -1).repeat(1, self.num_layers, 1)
            # device copies of the truncation centre / coefficients are cached: the reference passes CPU tensors every
            # step (E_align_s2.py:35-41,105), re-uploading them would put two synchronising copies into each iteration
            key = (id(self.buffer1), getattr(self.buffer1, "_version", 0), id(coefs_m), getattr(coefs_m, "_version", 0), str(dev))
            hit = self.__dict__.get("_trunc_cache")
            if hit is None or hit[0] != key:
                coefs = torch.as_tensor(coefs_m, dtype=torch.float32).reshape(-1)
                if coefs.numel() == 1:
                    coefs = coefs.repeat(self.num_layers)
                hit = (key, self.buffer1.to(dev).float().reshape(-1, x.shape[1]).contiguous(), coefs.to(dev).contiguous(), self.buffer1, coefs_m)
                self.__dict__["_trunc_cache"] = hit
            return ops.lerp_layers(x, hit[1], hit[2])
