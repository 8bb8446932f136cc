"""PGGAN generator (reference model/pggan/pggan_generator.py: PGGANGenerator :28-204,
ConvBlock :236-339) on the HIP kernels.  Same constructor arguments, forward signature, result
dict and state_dict keys (`lod`, `layer{i}.{weight,bias}`, `output{k}.{weight,bias}`; 43 keys at
256).  Per block: pixel norm -> [nearest x2] -> conv (weight*sqrt(2)/sqrt(fan_in)) + bias ->
lrelu(0.2); `layer0` is the 4x4 "dense" conv on the 1x1 latent.  The nearest upsample is fused
into the conv read, bias + activation into its epilogue.  Only lod == 0 (what the released
checkpoints and E_align use) is implemented; the stray `print(x.shape)` of the reference (:196) is
not reproduced.
"""
import numpy as np
import torch
from torch import nn

from . import ops
from .stylegan2_generator import _dt

_RESOLUTIONS_ALLOWED = [8, 16, 32, 64, 128, 256, 512, 1024]
_INIT_RES = 4


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, upsample=False, wscale_gain=np.sqrt(2.0),
                 activation_type="lrelu"):
        super().__init__()
        self.in_c, self.out_c, self.ksize, self.padding, self.upsample = in_channels, out_channels, kernel_size, padding, upsample
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, kernel_size, kernel_size))
        self.wscale = wscale_gain / np.sqrt(kernel_size * kernel_size * in_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        if activation_type not in ("linear", "lrelu"):
            raise NotImplementedError(f"Not implemented activation function: `{activation_type}`!")
        self.act = ops.ACT_LRELU if activation_type == "lrelu" else ops.ACT_NONE
        self._cache = {}

    def packed_dgrad(self, dtype):
        w = self.weight
        ver = (dtype, w._version, w.data_ptr(), getattr(w, "_dge_gen", 0))
        hit = self._cache.get("wd")
        if hit is None or hit[0] != ver:
            hit = (ver, ops.pack_conv_weight(w, ops.PACK_DGRAD, dtype, self.wscale))
            self._cache["wd"] = hit
        return hit[1]

    def packed(self, dtype):
        w = self.weight
        ver = (dtype, w._version, w.data_ptr(), getattr(w, "_dge_gen", 0))
        hit = self._cache.get("w")
        if hit is None or hit[0] != ver:
            hit = (ver, ops.pack_conv_weight(w, ops.PACK_FWD, dtype, self.wscale))
            self._cache["w"] = hit
        return hit[1]


class PGGANGenerator(nn.Module):
    def __init__(self, resolution, z_space_dim=512, image_channels=3, final_tanh=False, label_size=0, fused_scale=False,
                 use_wscale=True, fmaps_base=16 << 10, fmaps_max=512, compute_dtype="bf16"):
        super().__init__()
        if resolution not in _RESOLUTIONS_ALLOWED:
            raise ValueError(f"Invalid resolution: `{resolution}`!\nResolutions allowed: {_RESOLUTIONS_ALLOWED}.")
        if image_channels != 3 or final_tanh or label_size or fused_scale or not use_wscale:
            raise ValueError("only the released-checkpoint configuration is implemented (3 channels, no tanh, no labels, "
                             "fused_scale=False, use_wscale=True)")
        _dt(compute_dtype)
        self.compute_dtype = compute_dtype
        self.init_res, self.resolution, self.z_space_dim = _INIT_RES, resolution, z_space_dim
        self.init_res_log2, self.final_res_log2 = 2, int(np.log2(resolution))
        self.fmaps_base, self.fmaps_max = fmaps_base, fmaps_max
        self.num_layers = (self.final_res_log2 - self.init_res_log2 + 1) * 2
        self.register_buffer("lod", torch.zeros(()))
        for res_log2 in range(self.init_res_log2, self.final_res_log2 + 1):
            res = 2 ** res_log2
            k = res_log2 - self.init_res_log2
            if res == self.init_res:
                self.add_module(f"layer{2 * k}", ConvBlock(z_space_dim, self.get_nf(res), kernel_size=4, padding=3))
            else:
                self.add_module(f"layer{2 * k}", ConvBlock(self.get_nf(res // 2), self.get_nf(res), upsample=True))
            self.add_module(f"layer{2 * k + 1}", ConvBlock(self.get_nf(res), self.get_nf(res)))
            self.add_module(f"output{k}", ConvBlock(self.get_nf(res), image_channels, kernel_size=1, padding=0,
                                                    wscale_gain=1.0, activation_type="linear"))
        self._dense0 = None

    def get_nf(self, res):
        return min(self.fmaps_base // res, self.fmaps_max)

    def _dense0_weight(self):
        """layer0 is conv2d(4x4, pad 3) on a 1x1 input == a dense layer: out[(y,x),o] = sum_c z_c W[o,c,3-y,3-x]."""
        L = self.layer0
        ver = (L.weight._version, L.weight.data_ptr(), getattr(L.weight, "_dge_gen", 0))
        if self._dense0 is None or self._dense0[0] != ver:
            w = L.weight.detach().flip(2, 3).permute(2, 3, 0, 1).reshape(16 * L.out_c, L.in_c).contiguous()
            self._dense0 = (ver, w, L.bias.detach().repeat(16).contiguous())
        return self._dense0[1], self._dense0[2]

    def forward(self, z, label=None, lod=None, **_unused_kwargs):
        if z.ndim != 2 or z.shape[1] != self.z_space_dim:
            raise ValueError(f"Input latent code should be with shape [batch_size, latent_dim], where `latent_dim` equals "
                             f"to {self.z_space_dim}!\nBut `{z.shape}` is received!")
        lod = float(self.lod) if lod is None else lod
        if lod != 0:
            raise ValueError("only lod == 0 is implemented (the released generators are fully grown)")
        image, zn = _PGGANFunction.apply(self, z)
        return {"z": zn, "label": label, "image": image}

    # ------------------------------------------------------------------ pipelines
    def _run(self, z, save):
        dt = _dt(self.compute_dtype)
        B = z.shape[0]
        zf = z.float().contiguous()
        zn = ops.pixelnorm(zf)
        wd, bd = self._dense0_weight()
        h = ops.linear(zn, wd, bd, wscale=self.layer0.wscale, act=ops.ACT_LRELU)           # [B, 16*C0] == NHWC [B,4,4,C0]
        C0 = self.layer0.out_c
        x = ops.nchw_to_nhwc(h.view(B, 16 * C0, 1, 1), B, dt).view(B, 4, 4, C0)
        saved = dict(z=zf, h=h, convs=[]) if save else None
        nblk = self.final_res_log2 - self.init_res_log2 + 1
        for k in range(nblk):
            if k > 0:
                L = getattr(self, f"layer{2 * k}")
                y = ops.conv2d(ops.pixelnorm_nhwc(x), L.packed(dt), L.out_c, 3, bias=L.bias.detach(), act=L.act, in_up2=True)
                if save:
                    saved["convs"].append((L, x, y, True))
                x = y
            L = getattr(self, f"layer{2 * k + 1}")
            y = ops.conv2d(ops.pixelnorm_nhwc(x), L.packed(dt), L.out_c, 3, bias=L.bias.detach(), act=L.act)
            if save:
                saved["convs"].append((L, x, y, False))
            x = y
        O_ = getattr(self, f"output{nblk - 1}")
        ones = torch.ones((B, x.shape[3]), dtype=torch.float32, device=x.device)
        xn = ops.pixelnorm_nhwc(x)
        image = ops.torgb(xn, O_.weight.detach().reshape(3, -1), ones, O_.bias.detach(), None, O_.wscale)
        if save:
            saved.update(x_last=x, xn_last=xn, out=O_)
        return image, zn, saved

    def _backward(self, saved, g_image):
        """d(image)/d(z) contracted with g_image (hand-written data gradient; the generator's parameters are frozen in E_align)."""
        dt = ops.dtype_of(saved["x_last"])
        B = g_image.shape[0]
        O_ = saved["out"]
        ones = torch.ones((B, saved["x_last"].shape[3]), dtype=torch.float32, device=g_image.device)
        g_xn, _ = ops.torgb_bwd(g_image.float().contiguous(), saved["xn_last"], O_.weight.detach().reshape(3, -1), ones, O_.wscale)
        g = ops.pixelnorm_nhwc_bwd(g_xn, saved["x_last"])
        for L, x_in, y, up in reversed(saved["convs"]):
            g_pre = ops.act_bwd(g, y, None, pool=False, scale=1.0) if L.act == ops.ACT_LRELU else g
            g_xn = ops.conv2d(g_pre, L.packed_dgrad(dt), L.in_c, 3)
            if up:
                g_xn, _ = ops.nearest_up2_bwd(g_xn)
            g = ops.pixelnorm_nhwc_bwd(g_xn, x_in)
        # layer0: h = lrelu(wscale * zn @ wd^T + bd) viewed as NHWC [B,4,4,C0]
        C0 = self.layer0.out_c
        g_h = ops.nhwc_to_nchw(g.view(B, 1, 1, 16 * C0)).view(B, 4, 4, C0)          # f32, same (y, x, c) order as h
        g_hpre = ops.act_bwd(g_h, saved["h"].view(B, 4, 4, C0), None, pool=False, scale=1.0).view(B, 16 * C0)
        wd, _ = self._dense0_weight()
        g_zn = torch.empty((B, self.z_space_dim), dtype=torch.float32, device=g_image.device)
        ops.linear_t(g_hpre, wd, g_zn, scale=self.layer0.wscale)
        return ops.pixelnorm_nhwc_bwd(g_zn.view(B, 1, 1, -1), saved["z"].view(B, 1, 1, -1)).view(B, -1)


class _PGGANFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, G, z):
        need = ctx.needs_input_grad[1]
        image, zn, saved = G._run(z.detach(), save=need)
        ctx.G, ctx.saved_acts = G, saved
        ctx.mark_non_differentiable(zn)
        return image, zn

    @staticmethod
    def backward(ctx, g_image, _g_zn):
        if ctx.saved_acts is None:
            raise RuntimeError("PGGAN forward ran without saved activations")
        return None, ctx.G._backward(ctx.saved_acts, g_image)
