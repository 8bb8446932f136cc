"""LPIPS(net='vgg') on the HIP kernels - the perceptual term of space_loss
(reference training_utils.py:93, E_align_s2.py:98: `lpips.LPIPS(net='vgg')`, third-party).

Parameter names follow the `lpips` package (`net.slice{k}.{idx}.weight/bias`,
`lin{k}.model.1.weight`, `scaling_layer.shift/scale`) so its checkpoints load with
load_state_dict; without them (this image has neither the package nor the weights) the module
is initialised with seeded stand-in weights and says so (`self.pretrained = False`).
Forward runs both images through VGG16 as one batch; the gradient is produced for the second
image only (the reconstruction), which is all E_align needs.
"""
import ctypes as C

import torch
from torch import nn

from . import ops
from ._lib import lib, check
from .ops import _f32, _p, _stream
from .stylegan2_generator import _dt

_VGG = [(3, 64), (64, 64), "M", (64, 128), (128, 128), "M", (128, 256), (256, 256), (256, 256), "M",
        (256, 512), (512, 512), (512, 512), "M", (512, 512), (512, 512), (512, 512)]
_CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
_SLICE = [1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5]
_TAP_AFTER = [1, 3, 6, 9, 12]
_CPAD = 16      # first conv: 3 input channels padded to one 16-channel K chunk


class _Conv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(cout, cin, 3, 3) * (2.0 / (cin * 9)) ** 0.5, requires_grad=False)
        self.bias = nn.Parameter(torch.randn(cout) * 0.05, requires_grad=False)


class _Lin(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.model = nn.Module()
        conv = nn.Module()
        conv.weight = nn.Parameter(torch.randn(1, c, 1, 1).abs() * 0.1, requires_grad=False)
        self.model.add_module("1", conv)


import os as _os
_PP_LPIPS = not _os.environ.get("DGE_NO_PP_LPIPS")


class LPIPS(nn.Module):
    def __init__(self, net="vgg", compute_dtype="bf16"):
        super().__init__()
        if net != "vgg":
            raise ValueError("only net='vgg' is used by the reference (E_align_s2.py:98)")
        self.compute_dtype = compute_dtype
        _dt(compute_dtype)
        self.pretrained = False
        self.scaling_layer = nn.Module()
        self.scaling_layer.register_buffer("shift", torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1))
        self.scaling_layer.register_buffer("scale", torch.tensor([.458, .448, .450]).view(1, 3, 1, 1))
        self.net = nn.Module()
        for k in range(1, 6):
            self.net.add_module(f"slice{k}", nn.Module())
        self.convs = []
        ci = 0
        for item in _VGG:
            if item == "M":
                continue
            conv = _Conv(*item)
            getattr(self.net, f"slice{_SLICE[ci]}").add_module(str(_CONV_IDX[ci]), conv)
            self.convs.append(conv)
            ci += 1
        for k, c in enumerate([64, 128, 256, 512, 512]):
            self.add_module(f"lin{k}", _Lin(c))
        self._cache = {}

    # ---------------------------------------------------------------- weights
    def load_pretrained(self, vgg_weights, lin_weights=None):
        """Real LPIPS-VGG weights from the two files a user of the reference already has (training_utils.py:93,
        requirements.txt:12 `lpips`): `vgg_weights` = torchvision's vgg16 checkpoint (`features.{i}.weight/bias`, what
        lpips.pretrained_networks.vgg16 wraps) or a full `lpips.LPIPS(net='vgg').state_dict()` (`net.slice{k}.{i}.*`);
        `lin_weights` = the package's `weights/v0.1/vgg.pth` (`lin{k}.model.1.weight`; the `lins.{k}.*` duplicates that
        newer versions write are ignored).  Paths or already-loaded dicts.  Sets `self.pretrained = True` only when every
        convolution and every linear head was found; raises KeyError otherwise."""
        def _load(src):
            return torch.load(src, map_location="cpu") if isinstance(src, (str, bytes)) or hasattr(src, "read") else dict(src)
        sd = self.state_dict()
        new = {}
        vgg = _load(vgg_weights)
        for ci, idx in enumerate(_CONV_IDX):
            for leaf in ("weight", "bias"):
                mine = f"net.slice{_SLICE[ci]}.{idx}.{leaf}"
                src = mine if mine in vgg else f"features.{idx}.{leaf}"
                if src not in vgg:
                    raise KeyError(f"LPIPS.load_pretrained: neither {mine} nor features.{idx}.{leaf} in the VGG16 weights")
                new[mine] = vgg[src]
        lins = _load(lin_weights) if lin_weights is not None else vgg
        for k in range(5):
            mine = f"lin{k}.model.1.weight"
            src = mine if mine in lins else f"lins.{k}.model.1.weight"
            if src not in lins:
                raise KeyError(f"LPIPS.load_pretrained: {mine} missing from the linear-head weights")
            new[mine] = lins[src]
        for k in ("scaling_layer.shift", "scaling_layer.scale"):
            new[k] = vgg.get(k, lins.get(k, sd[k]))
        for k, v in new.items():
            if tuple(v.shape) != tuple(sd[k].shape):
                raise ValueError(f"LPIPS.load_pretrained: {k} has shape {tuple(v.shape)}, expected {tuple(sd[k].shape)}")
        self.load_state_dict(new)
        self._cache.clear()
        self.pretrained = True
        return self

    def _packed(self, ci, dt, mode, hw=None):
        conv = self.convs[ci]
        if hw is not None and ci > 0:       # conv5_x on the cropped images run on the low-resolution kernel: fragment-ordered copy
            mode = ops.pack_mode_for(conv.weight, mode, hw[0], hw[1], dt)
        key = (ci, dt, mode)
        ver = (conv.weight._version, conv.weight.data_ptr(), getattr(conv.weight, "_dge_gen", 0))
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            w = conv.weight.detach()
            if ci == 0:                 # pad Cin 3 -> 16 with zeros
                wp = torch.zeros((w.shape[0], _CPAD, 3, 3), dtype=torch.float32, device=w.device)
                wp[:, :3] = w
                w = wp
            hit = (ver, ops.pack_conv_weight(w, mode, dt, 1.0))
            self._cache[key] = hit
        return hit[1]

    def _packed_pp(self, ci):
        """shared weight image of conv ci for ops.conv_pp (cached on the weight's version)"""
        w = self.convs[ci].weight
        ver = (w._version, w.data_ptr())
        hit = self._cache.get(("pp", ci))
        if hit is None or hit[0] != ver:
            hit = (ver, ops.pack_conv_pp(w.detach().float(), 1.0))
            self._cache[("pp", ci)] = hit
        return hit[1]

    def _scaling_host(self):
        """Host copies of the ScalingLayer constants (kernel arguments); read back from the device only when the
        buffers change -- a .tolist() per call would put a device synchronisation into every loss evaluation."""
        sl = self.scaling_layer
        key = (sl.shift._version, sl.shift.data_ptr(), sl.scale._version, sl.scale.data_ptr())
        hit = self._cache.get("scaling")
        if hit is None or hit[0] != key:
            hit = (key, (C.c_float * 3)(*sl.shift.flatten().tolist()), (C.c_float * 3)(*sl.scale.flatten().tolist()))
            self._cache["scaling"] = hit
        return hit[1], hit[2]

    # ---------------------------------------------------------------- forward (+ gradient w.r.t. b)
    def value_and_grad(self, a, b, need_grad=True):
        """a, b: [B,3,h,w] f32 in [-1,1].  Returns (mean over the batch of LPIPS(a,b) as a [1]
        device tensor, d mean / d b  [B,3,h,w] f32 or None)."""
        dt = _dt(self.compute_dtype)
        if a.shape[1] == 1:
            # single-channel maps (the Grad-CAM masks of E_mis_align_cropping_s1.py:182): lpips' ScalingLayer broadcasts
            # them against its [1,3,1,1] constants, i.e. the map is fed as three identical channels
            if need_grad:
                raise ValueError("LPIPS gradient of a single-channel input is not provided (the reference never uses it)")
            a = a.expand(-1, 3, -1, -1).contiguous()
            b = b.expand(-1, 3, -1, -1).contiguous()
        B, _, h, w = a.shape
        dev = a.device
        L = lib()
        shift, scale = self._scaling_host()
        x = torch.empty((2 * B, h, w, _CPAD), dtype=ops.tdtype(dt), device=dev)
        check(L.dge_lpips_prep(_f32(a.contiguous()), _p(x[:B]), B, h * w, _CPAD, shift, scale, dt, _stream()), "dge_lpips_prep")
        check(L.dge_lpips_prep(_f32(b.contiguous()), _p(x[B:]), B, h * w, _CPAD, shift, scale, dt, _stream()), "dge_lpips_prep")
        acts, pools = [], {}        # acts[ci] = output of conv ci (post relu); pools[ci] = pooled input of conv ci
        ci = 0
        cur = x
        for item in _VGG:
            if item == "M":
                Bc, Hc, Wc, Cc = cur.shape
                y = torch.empty((Bc, Hc // 2, Wc // 2, Cc), dtype=cur.dtype, device=dev)
                check(L.dge_maxpool2(_p(cur), _p(y), Bc, Hc, Wc, Cc, dt, _stream()), "dge_maxpool2")
                pools[ci] = y
                cur = y
                continue
            conv = self.convs[ci]
            Bc, Hc, Wc, Cc = cur.shape
            if Cc >= 64 and _PP_LPIPS and ops.conv_pp_supported(Bc, Hc, Wc, Cc, item[1], dt):
                # the MFMA-bound VGG layers (>= 128 output channels on a grid that fills the chip): ping-pong implicit GEMM
                cur = ops.conv_pp(cur, self._packed_pp(ci), item[1], bias=conv.bias.detach(), act=ops.ACT_RELU)
            else:
                cur = ops.conv2d(cur, self._packed(ci, dt, ops.PACK_FWD, cur.shape[1:3]), item[1], 3, bias=conv.bias.detach(), act=ops.ACT_RELU)
            acts.append(cur)
            ci += 1
        val = torch.zeros(B, dtype=torch.float32, device=dev)
        heads = []
        for k, cidx in enumerate(_TAP_AFTER):
            f = acts[cidx]
            _, fh, fw, fc = f.shape
            g1 = torch.empty((B, fh, fw, fc), dtype=f.dtype, device=dev) if need_grad else None
            lin = getattr(self, f"lin{k}").model._modules["1"].weight.detach().reshape(-1).contiguous()
            check(L.dge_lpips_head(_p(f), _f32(lin), _p(val), _p(g1), B, fh * fw, fc, 1.0 / B, dt, _stream()), "dge_lpips_head")
            heads.append(g1)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        check(L.dge_mean(_p(val), _p(out), B, _stream()), "dge_mean")
        if not need_grad:
            return out, None
        # ---- backward through VGG16 for the b half of the batch.  g_pre = gradient w.r.t. the PRE-activation of conv ci; the ReLU
        #      backward of conv ci-1 rides in the launch that produces its input gradient (data-gradient epilogue / pool adjoint),
        #      except where that launch runs on the streaming kernel (64 -> 64 at 256^2), which has no such stage.
        g_pre = ops.act_bwd(heads[_TAP_AFTER.index(12)], acts[12][B:], slope=0.0)
        for ci in range(12, -1, -1):
            cin = _CPAD if ci == 0 else _VGG_CIN[ci]
            wd = self._packed(ci, dt, ops.PACK_DGRAD, g_pre.shape[1:3])
            if ci == 0:
                g = ops.conv2d(g_pre, wd, cin, 3)
                break
            below = acts[ci - 1][B:]
            head = heads[_TAP_AFTER.index(ci - 1)] if (ci - 1) in _TAP_AFTER else None
            if ci in pools:          # conv ci read a pooled tensor: route through the max pool, add the tap gradient, ReLU backward
                g_in = ops.conv2d(g_pre, wd, cin, 3)
                g_pre = ops.maxpool2_bwd(g_in, below, head, relu=True)
            elif head is None and cin <= 64 and _VGG_COUT[ci] <= 64 and g_pre.shape[1] >= 128:
                g_pre = ops.act_bwd(ops.conv2d(g_pre, wd, cin, 3), below, slope=0.0)
            else:
                g_pre = ops.conv2d(g_pre, wd, cin, 3, addend=head, relu_mask=below)
        gb = torch.empty((B, 3, h, w), dtype=torch.float32, device=dev)
        check(L.dge_lpips_prep_bwd(_p(g), _p(gb), B, h * w, _CPAD, scale, 1.0, 0, dt, _stream()), "dge_lpips_prep_bwd")
        return out, gb

    def forward(self, a, b):
        """lpips.LPIPS.forward surface: per-sample distances [B,1,1,1] (no gradient)."""
        vals = []
        for i in range(a.shape[0]):
            v, _ = self.value_and_grad(a[i:i + 1], b[i:i + 1], need_grad=False)
            vals.append(v)
        return torch.stack(vals).view(-1, 1, 1, 1)


_VGG_CIN = [item[0] for item in _VGG if item != "M"]
_VGG_COUT = [item[1] for item in _VGG if item != "M"]
