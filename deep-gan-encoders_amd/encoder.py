"""Encoder "case 1" (reference model/E/E.py: BEBlock :16-85, BE :88-135) on the HIP kernels.

Same constructor / forward signature / state_dict keys as the reference (101 keys for
startf=16, layer_count=9; 77 for startf=64, layer_count=7), so released encoder checkpoints and
`E.load_state_dict(torch.load(...))` (E_align_s2.py:92-93) work unchanged.  Parameters carry the
`lr_equalization_coef` tags LREQAdam consumes.

Per block (E.py:50-85): (mean,std) -> w1 ; IN -> conv3x3 -> +noise -> +bias -> lrelu ; (mean,std) -> w2 ;
IN -> [conv3x3 -> noise -> bias -> lrelu -> avgpool2] ; residual [avgpool2] [-> 1x1 conv] ;
x = 0.111*x + 0.889*res.  Instance-norm application is fused into the conv prologue and the
statistics of every conv output come out of the conv epilogue, so each activation is written
once and read once per consumer.
"""
import torch
from torch import nn

from . import lreq as ln
from . import ops
from .stylegan2_generator import _dt


class FromRGB(nn.Module):
    def __init__(self, channels, outputs):
        super().__init__()
        self.from_rgb = ln.Conv2d(channels, outputs, 1, 1, 0)


class BEBlock(nn.Module):
    def __init__(self, inputs, outputs, latent_size, has_last_conv=True, fused_scale=False):
        super().__init__()
        if fused_scale:
            raise ValueError("fused_scale=True belongs to E_Blur (case 2), not E.BE")
        self.has_last_conv, self.inputs, self.outputs = has_last_conv, inputs, outputs
        self.noise_weight_1 = nn.Parameter(torch.zeros(1, inputs, 1, 1))
        self.bias_1 = nn.Parameter(torch.zeros(1, inputs, 1, 1))
        self.inver_mod1 = ln.Linear(2 * inputs, latent_size, gain=1)
        self.conv_1 = ln.Conv2d(inputs, inputs, 3, 1, 1, bias=False)
        self.noise_weight_2 = nn.Parameter(torch.zeros(1, outputs, 1, 1))
        self.bias_2 = nn.Parameter(torch.zeros(1, outputs, 1, 1))
        self.inver_mod2 = ln.Linear(2 * inputs, latent_size, gain=1)
        if has_last_conv:
            self.conv_2 = ln.Conv2d(inputs, outputs, 3, 1, 1, bias=False)
        if inputs != outputs:
            self.conv_3 = ln.Conv2d(inputs, outputs, 1, 1, 0)


class BE(nn.Module):
    def __init__(self, startf=16, maxf=512, layer_count=9, latent_size=512, channels=3, compute_dtype="bf16"):
        super().__init__()
        if latent_size != 512 or channels != 3:
            raise ValueError("latent_size=512 and channels=3 are hard-wired in the reference forward (E.py:130)")
        _dt(compute_dtype)
        self.maxf, self.startf, self.latent_size, self.layer_count = maxf, startf, latent_size, layer_count
        self.compute_dtype = compute_dtype
        self.decode_block = nn.ModuleList()       # registered before FromRGB, like the reference (E.py:95,101)
        self.FromRGB = FromRGB(channels, startf)
        inputs, outputs = startf, startf * 2
        for i in range(layer_count):
            self.decode_block.append(BEBlock(inputs, outputs, latent_size, i + 1 != layer_count, False))
            inputs, outputs = min(maxf, inputs * 2), min(maxf, outputs * 2)

    def forward(self, x, block_num=9, noises=None):
        """x: images [B,3,R,R] f32 (NCHW).  Returns (x [B,C,4,4] f32, w [B,2*layer_count,512]).
        `noises`: optional list of N(0,1) tensors [B,1,H,W] in the reference's draw order
        (2 per block, 1 for the last; E.py:60,73); default: drawn on the device."""
        if block_num != 9:
            raise ValueError("progressive block_num != 9 is not used by E_align (E.py:122,127)")
        from .autograd_enc import EncoderFunction
        params = [p for p in self.parameters()]
        return EncoderFunction.apply(self, x, noises, *params)
