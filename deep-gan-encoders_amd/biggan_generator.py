"""BigGAN-deep generator (reference model/biggan_generator.py: BigGANBatchNorm :100-150, GenBlock
:153-203, SelfAttn :58-97, Generator :206-256, BigGAN :259-304; config model/utils/biggan_config.py)
on the HIP kernels.  Same class names, forward signatures and state_dict keys (602 keys for
deep-256: `embeddings.weight`, `generator.gen_z.{bias,weight_orig,weight_u,weight_v}`,
`generator.layers.{n}.bn_{k}.{running_means,running_vars,scale.weight_*,offset.weight_*}`, ...).

Conditional batch norm + ReLU are fused into the prologue of the following convolution, the
nearest upsample into its read, bias and the residual add into its epilogue; self-attention is
a dedicated kernel.  Spectral normalisation keeps torch.nn.utils.spectral_norm's semantics
(`weight_orig / sigma`, one power iteration per forward in train mode - SURVEY Q2) as a few
small matvecs on the parameters; the effective weight is what gets packed for the conv kernel.
Only 3 of the 128 output channels of `conv_to_rgb` are ever used (:251-253): 16 are
computed.
"""
import copy
import json
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .stylegan2_generator import _dt


class BigGANConfig(object):
    """model/utils/biggan_config.py:11-71"""

    def __init__(self, output_dim=128, z_dim=128, class_embed_dim=128, channel_width=128, num_classes=1000,
                 layers=((False, 16, 16), (True, 16, 16), (False, 16, 16), (True, 16, 8), (False, 8, 8), (True, 8, 4),
                         (False, 4, 4), (True, 4, 2), (False, 2, 2), (True, 2, 1)),
                 attention_layer_position=8, eps=1e-4, n_stats=51):
        self.output_dim, self.z_dim, self.class_embed_dim, self.channel_width = output_dim, z_dim, class_embed_dim, channel_width
        self.num_classes, self.layers, self.attention_layer_position, self.eps, self.n_stats = \
            num_classes, [tuple(l) for l in layers], attention_layer_position, eps, n_stats

    @classmethod
    def from_dict(cls, json_object):
        config = BigGANConfig()
        for key, value in json_object.items():
            config.__dict__[key] = value
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding="utf-8") as reader:
            return cls.from_dict(json.loads(reader.read()))

    def to_dict(self):
        return copy.deepcopy(self.__dict__)


class _SN(nn.Module):
    """Parameter holder with the state_dict layout of torch.nn.utils.spectral_norm(module)."""

    def __init__(self, weight_shape, bias, eps):
        super().__init__()
        if bias:
            self.bias = nn.Parameter(torch.zeros(weight_shape[0]))
        else:
            self.bias = None
        self.weight_orig = nn.Parameter(torch.randn(*weight_shape) * 0.05)
        out = weight_shape[0]
        self.register_buffer("weight_u", F.normalize(torch.randn(out), dim=0, eps=eps))
        self.register_buffer("weight_v", F.normalize(torch.randn(int(torch.tensor(weight_shape[1:]).prod())), dim=0, eps=eps))
        self.eps = eps

    @torch.no_grad()
    def effective_weight(self, training):
        w = self.weight_orig.detach()
        wm = w.reshape(w.shape[0], -1)
        u, v = self.weight_u, self.weight_v
        if training:                               # one power iteration, buffers updated in place
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=self.eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=self.eps))
        sigma = torch.dot(u, torch.mv(wm, v))
        return w / sigma


class BigGANBatchNorm(nn.Module):
    def __init__(self, num_features, condition_vector_dim=None, n_stats=51, eps=1e-4, conditional=True):
        super().__init__()
        self.num_features, self.eps, self.conditional = num_features, eps, conditional
        self.register_buffer("running_means", torch.zeros(n_stats, num_features))
        self.register_buffer("running_vars", torch.ones(n_stats, num_features))
        self.step_size = 1.0 / (n_stats - 1)
        if conditional:
            self.scale = _SN((num_features, condition_vector_dim), False, eps)
            self.offset = _SN((num_features, condition_vector_dim), False, eps)
        else:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))

    def affine(self, truncation, cond, training):
        coef, start_idx = math.modf(truncation / self.step_size)
        start_idx = int(start_idx)
        if coef != 0.0:
            mean = self.running_means[start_idx] * coef + self.running_means[start_idx + 1] * (1 - coef)
            var = self.running_vars[start_idx] * coef + self.running_vars[start_idx + 1] * (1 - coef)
        else:
            mean, var = self.running_means[start_idx], self.running_vars[start_idx]
        if self.conditional:
            sc = ops.linear(cond, self.scale.effective_weight(training).contiguous())
            of = ops.linear(cond, self.offset.effective_weight(training).contiguous())
        else:
            sc = (self.weight.detach() - 1.0).reshape(1, -1).contiguous()
            of = self.bias.detach().reshape(1, -1).contiguous()
        return ops.cbn_affine(sc, of, mean, var, self.eps)


class GenBlock(nn.Module):
    def __init__(self, in_size, out_size, condition_vector_dim, reduction_factor=4, up_sample=False, n_stats=51, eps=1e-12):
        super().__init__()
        self.up_sample, self.drop_channels = up_sample, in_size != out_size
        self.in_size, self.out_size = in_size, out_size
        mid = in_size // reduction_factor
        self.mid = mid
        self.bn_0 = BigGANBatchNorm(in_size, condition_vector_dim, n_stats, eps, True)
        self.conv_0 = _SN((mid, in_size, 1, 1), True, eps)
        self.bn_1 = BigGANBatchNorm(mid, condition_vector_dim, n_stats, eps, True)
        self.conv_1 = _SN((mid, mid, 3, 3), True, eps)
        self.bn_2 = BigGANBatchNorm(mid, condition_vector_dim, n_stats, eps, True)
        self.conv_2 = _SN((mid, mid, 3, 3), True, eps)
        self.bn_3 = BigGANBatchNorm(mid, condition_vector_dim, n_stats, eps, True)
        self.conv_3 = _SN((out_size, mid, 1, 1), True, eps)

    def run(self, x, cond, truncation, dt, training):
        def conv(sn, bn, inp, k, cout, **kw):
            a, b = bn.affine(truncation, cond, training)
            wp = ops.pack_conv_weight(sn.effective_weight(training).contiguous(), ops.PACK_FWD, dt, 1.0)
            return ops.conv2d(inp, wp, cout, k, in_scale=a, in_shift=b, in_relu=True, bias=sn.bias.detach(), **kw)
        t = conv(self.conv_0, self.bn_0, x, 1, self.mid)
        t = conv(self.conv_1, self.bn_1, t, 3, self.mid, in_up2=self.up_sample)
        t = conv(self.conv_2, self.bn_2, t, 3, self.mid)
        skip = ops.slice_up(x, self.out_size, self.up_sample) if (self.drop_channels or self.up_sample) else x
        return conv(self.conv_3, self.bn_3, t, 1, self.out_size, addend=skip, add_scale=1.0)


class SelfAttn(nn.Module):
    def __init__(self, in_channels, eps=1e-12):
        super().__init__()
        self.in_channels = in_channels
        self.snconv1x1_theta = _SN((in_channels // 8, in_channels, 1, 1), False, eps)
        self.snconv1x1_phi = _SN((in_channels // 8, in_channels, 1, 1), False, eps)
        self.snconv1x1_g = _SN((in_channels // 2, in_channels, 1, 1), False, eps)
        self.snconv1x1_o_conv = _SN((in_channels, in_channels // 2, 1, 1), False, eps)
        self.gamma = nn.Parameter(torch.zeros(1))

    def run(self, x, dt, training):
        B, H, W, Cc = x.shape
        pk = lambda sn: ops.pack_conv_weight(sn.effective_weight(training).contiguous(), ops.PACK_FWD, dt, 1.0)
        theta = ops.conv2d(x, pk(self.snconv1x1_theta), Cc // 8, 1)
        phi = ops.maxpool2(ops.conv2d(x, pk(self.snconv1x1_phi), Cc // 8, 1))
        g = ops.maxpool2(ops.conv2d(x, pk(self.snconv1x1_g), Cc // 2, 1))
        o = ops.attention(theta.view(B, H * W, Cc // 8), phi.view(B, H * W // 4, Cc // 8), g.view(B, H * W // 4, Cc // 2))
        gam = self.gamma.detach().reshape(1, 1).expand(B, Cc).contiguous()
        return ops.conv2d(o.view(B, H, W, Cc // 2), pk(self.snconv1x1_o_conv), Cc, 1, out_scale=gam, addend=x, add_scale=1.0)


class Generator(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        ch = config.channel_width
        cdim = config.z_dim * 2
        self.gen_z = _SN((4 * 4 * 16 * ch, cdim), True, config.eps)
        layers = []
        for i, layer in enumerate(config.layers):
            if i == config.attention_layer_position:
                layers.append(SelfAttn(ch * layer[1], eps=config.eps))
            layers.append(GenBlock(ch * layer[1], ch * layer[2], cdim, up_sample=layer[0], n_stats=config.n_stats, eps=config.eps))
        self.layers = nn.ModuleList(layers)
        self.bn = BigGANBatchNorm(ch, n_stats=config.n_stats, eps=config.eps, conditional=False)
        self.conv_to_rgb = _SN((ch, ch, 3, 3), True, config.eps)

    def forward(self, cond_vector, truncation, compute_dtype="bf16"):
        dt = _dt(compute_dtype)
        training = self.training
        truncation = float(truncation)
        B = cond_vector.shape[0]
        ch = self.config.channel_width
        z = ops.linear(cond_vector, self.gen_z.effective_weight(training).contiguous(), self.gen_z.bias.detach())   # [B, 4*4*16ch] == NHWC
        x = ops.nchw_to_nhwc(z.view(B, 4 * 4 * 16 * ch, 1, 1), B, dt).view(B, 4, 4, 16 * ch)
        for layer in self.layers:
            x = layer.run(x, cond_vector, truncation, dt, training) if isinstance(layer, GenBlock) else layer.run(x, dt, training)
        a, b = self.bn.affine(truncation, None, training)
        a, b = a.expand(B, -1).contiguous(), b.expand(B, -1).contiguous()
        w = self.conv_to_rgb.effective_weight(training)[:16].contiguous()            # only channels 0..2 are used (:253)
        y = ops.conv2d(x, ops.pack_conv_weight(w, ops.PACK_FWD, dt, 1.0), 16, 3, in_scale=a, in_shift=b, in_relu=True,
                       bias=self.conv_to_rgb.bias.detach()[:16].contiguous())
        return ops.rgb_tanh(y)


class BigGAN(nn.Module):
    def __init__(self, config, compute_dtype="bf16"):
        super().__init__()
        _dt(compute_dtype)
        self.config, self.compute_dtype = config, compute_dtype
        self.embeddings = nn.Linear(config.num_classes, config.z_dim, bias=False)
        self.generator = Generator(config)

    def forward(self, z, class_label, truncation):
        truncation = float(truncation)
        assert 0 < truncation <= 1
        with torch.no_grad():
            embed = ops.linear(class_label.float().contiguous(), self.embeddings.weight.detach())
            cond_vector = torch.cat((z.float(), embed), dim=1).contiguous()
            img = self.generator(cond_vector, truncation, self.compute_dtype)
        return img, cond_vector
