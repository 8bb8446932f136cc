"""Grad-CAM / Grad-CAM++ attention maps, guided back-propagation and mask2cam on the HIP kernels
(reference metric/grad_cam.py, wired by E_mis_align_cropping_s1.py:99-106,159-170).

Surface follows the reference: `GradCAM(net, layer_name)`, `GradCamPlusPlus(net, layer_name)`,
`GuidedBackPropagation(net)` are callables `(inputs, index) -> tensor`, `mask2cam(mask, imgs) -> (heatmap, cam)`.
`net` is `VGG16` below - torchvision.models.vgg16's layout and state_dict keys (`features.{0..28}.weight/bias`,
`classifier.{0,3,6}.weight/bias`), so the torchvision checkpoint the script downloads loads with load_state_dict;
without it (this image has neither torchvision nor the weights) the module carries seeded stand-in weights and says
so (`pretrained = False`): parity with the real classifier is structural.

What the reference's hook mechanics amount to, and how they are kept (probed by running the reference's classes,
tools/gen_golden_gradcam.py):
  * the forward hook on the last Conv2d stores a tensor that the following in-place ReLU overwrites: `feature` is the
    POST-ReLU activation;
  * `GuidedBackPropagation(net)` registers `clamp(grad_in, min=0)` on every nn.ReLU of the SAME network the Grad-CAM
    objects use, so from then on every backward through it - Grad-CAM's too - is guided: constructing
    `GuidedBackPropagation(net)` sets `net.guided = True` here;
  * the returned masks are CPU float64 tensors in the reference (numpy + cv2 per image); here they are float32 device
    tensors and nothing leaves the GPU (the class index is selected on the device as well).
All device math is in libdge_hip.so; there is no CPU path.
"""
import ctypes as C

import numpy as np
import torch
from torch import nn

from . import ops
from ._lib import lib, check, DgeError
from .ops import _f32, _p, _stream
from .stylegan2_generator import _dt

VGG16_WIDTHS = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
_CPAD = 16      # first conv: 3 input channels padded to one 16-channel K chunk


class _Param(nn.Module):
    def __init__(self, wshape, fan_in, gen):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(wshape, generator=gen) * (2.0 / fan_in) ** 0.5, requires_grad=False)
        self.bias = nn.Parameter(torch.randn(wshape[0], generator=gen) * 0.05, requires_grad=False)


class VGG16(nn.Module):
    """torchvision.models.vgg16 as an inference + data-gradient pipeline: features (13 conv3x3+ReLU, 5 MaxPool2d(2,2)),
    AdaptiveAvgPool2d(7), classifier (Linear-ReLU-Dropout-Linear-ReLU-Dropout-Linear; eval mode: dropout is the
    identity, grad_cam.py:22).  `widths` / `fc` / `num_classes` default to VGG16; narrower ones serve the tests."""

    def __init__(self, widths=VGG16_WIDTHS, fc=4096, num_classes=1000, compute_dtype="bf16", seed=0):
        super().__init__()
        _dt(compute_dtype)
        self.compute_dtype = compute_dtype
        self.pretrained = False
        self.guided = False
        gen = torch.Generator().manual_seed(4321 + seed)
        self.features = nn.Module()
        self.classifier = nn.Module()
        self.plan = []          # ("conv", features index, cin, cout) | ("pool",)
        idx, cin = 0, 3
        for w in widths:
            if w == "M":
                self.plan.append(("pool",))
                idx += 1
                continue
            self.features.add_module(str(idx), _Param((w, cin, 3, 3), cin * 9, gen))
            self.plan.append(("conv", idx, cin, w))
            idx += 2
            cin = w
        self.final_channels = cin
        for k, (o, i) in zip((0, 3, 6), ((fc, cin * 49), (fc, fc), (num_classes, fc))):
            self.classifier.add_module(str(k), _Param((o, i), i, gen))
        self._cache = {}

    def load_pretrained(self, weights):
        """torchvision's vgg16 checkpoint (`features.{i}.weight/bias`, `classifier.{k}.weight/bias`: the names this module
        uses) from a path or a dict; marks the module `pretrained` (without it the class scores are those of seeded stand-in
        weights, and a Grad-CAM computed on them is structural only)."""
        sd = torch.load(weights, map_location="cpu") if isinstance(weights, (str, bytes)) or hasattr(weights, "read") else dict(weights)
        self.load_state_dict(sd)
        self._cache.clear()
        self.pretrained = True
        return self

    @property
    def final_layer(self):
        """Name of the last Conv2d, what E_mis_align_cropping_s1.py:101-104 searches for."""
        return "features.%d" % [p[1] for p in self.plan if p[0] == "conv"][-1]

    def _packed(self, idx, dt, mode):
        conv = getattr(self.features, str(idx))
        key = (idx, dt, mode)
        ver = (conv.weight._version, conv.weight.data_ptr())
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            w = conv.weight.detach()
            if w.shape[1] == 3:                   # pad Cin 3 -> 16 with zeros
                wp = torch.zeros((w.shape[0], _CPAD, 3, 3), dtype=torch.float32, device=w.device)
                wp[:, :3] = w
                w = wp
            hit = (ver, ops.pack_conv_weight(w, mode, dt, 1.0))
            self._cache[key] = hit
        return hit[1]

    # ------------------------------------------------------------------ forward
    def run(self, inputs):
        """inputs [N,3,H,W] f32 -> (logits [N,K] f32, saved state for the backward)."""
        if inputs.dim() != 4 or inputs.shape[1] != 3:
            raise ValueError(f"inputs must be [N,3,H,W], got {tuple(inputs.shape)}")
        npool = sum(1 for p in self.plan if p[0] == "pool")
        N, _, H, W = inputs.shape
        if H % (1 << npool) or W % (1 << npool):
            raise ValueError(f"H and W must be multiples of {1 << npool}, got {H}x{W}")
        dt = _dt(self.compute_dtype)
        L = lib()
        dev = inputs.device
        one, zero = (C.c_float * 3)(1, 1, 1), (C.c_float * 3)(0, 0, 0)
        x = torch.empty((N, H, W, _CPAD), dtype=ops.tdtype(dt), device=dev)
        check(L.dge_lpips_prep(_f32(inputs.contiguous()), _p(x), N, H * W, _CPAD, zero, one, dt, _stream()), "dge_lpips_prep")
        acts = []            # post-ReLU output of every conv, in plan order
        cur = x
        for p in self.plan:
            if p[0] == "pool":
                cur = ops.maxpool2(cur)
                continue
            conv = getattr(self.features, str(p[1]))
            cur = ops.conv2d(cur, self._packed(p[1], dt, ops.PACK_FWD), p[3], 3, bias=conv.bias.detach(), act=ops.ACT_RELU)
            acts.append(cur)
        Bp, Hp, Wp, Cp = cur.shape
        flat = torch.empty((N, Cp * 49), dtype=torch.float32, device=dev)
        check(L.dge_adaptive_pool7(_p(cur), _p(flat), N, Hp, Wp, Cp, dt, _stream()), "dge_adaptive_pool7")
        c0, c3, c6 = (getattr(self.classifier, k) for k in ("0", "3", "6"))
        h1 = ops.linear(flat, c0.weight.detach(), c0.bias.detach(), act=ops.ACT_RELU)
        h2 = ops.linear(h1, c3.weight.detach(), c3.bias.detach(), act=ops.ACT_RELU)
        logits = ops.linear(h2, c6.weight.detach(), c6.bias.detach())
        return logits, dict(acts=acts, pooled_shape=(Hp, Wp, Cp), h1=h1, h2=h2, N=N, H=H, W=W, dt=dt)

    def forward(self, inputs):
        return self.run(inputs)[0]

    # ------------------------------------------------------------------ backward of target = mean_n logits[n, index_max]
    def select_target(self, logits, index=None):
        """grad_cam.py:166-170 on the device.  Returns (index tensor int32 [1+N]: [index_max, per-row indices], glogits)."""
        N, K = logits.shape
        idx_in = None
        if index is not None:
            idx_in = torch.as_tensor(index, dtype=torch.int32).reshape(-1).to(logits.device)
            if idx_in.numel() != N:
                raise ValueError("index must hold one class id per input")
        out = torch.empty(1 + N, dtype=torch.int32, device=logits.device)
        g = torch.empty_like(logits)
        check(lib().dge_class_target(_f32(logits), _p(idx_in), _p(out), _f32(g), N, K, _stream()), "dge_class_target")
        return out, g

    def _relu_bwd(self, g, a):
        out = torch.empty_like(a)
        check(lib().dge_guided_relu_bwd(_p(g), _p(a), _p(out), a.numel(), 1 if self.guided else 0, ops.dtype_of(a), _stream()),
              "dge_guided_relu_bwd")
        return out

    def _pool_relu_bwd(self, gy, a):
        """MaxPool2d(2,2) backward + the (guided) backward of the ReLU that produced `a`, one pass."""
        B, H, W, Cc = a.shape
        out = torch.empty_like(a)
        check(lib().dge_maxpool2_relu_bwd(_p(gy), _p(a), _p(out), B, H, W, Cc, 1 if self.guided else 0, ops.dtype_of(a), _stream()),
              "dge_maxpool2_relu_bwd")
        return out

    def backward_to_last_conv(self, st, index_dev):
        """Gradient of the target w.r.t. the OUTPUT of the last conv (pre-ReLU), NHWC [N,h,w,C] - what the backward hook
        on `features.28` receives (grad_cam.py:30-40)."""
        L = lib()
        N, dt = st["N"], st["dt"]
        c0, c3, c6 = (getattr(self.classifier, k) for k in ("0", "3", "6"))
        dev = st["h2"].device
        g2 = torch.empty_like(st["h2"])
        check(L.dge_gather_row(_f32(c6.weight.detach()), _p(index_dev), _f32(g2), N, g2.shape[1], 1.0 / N, _stream()), "dge_gather_row")
        g2 = self._relu_bwd(g2, st["h2"])
        g1 = ops.linear_t(g2, c3.weight.detach(), torch.empty_like(st["h1"]))
        g1 = self._relu_bwd(g1, st["h1"])
        Hp, Wp, Cp = st["pooled_shape"]
        gflat = ops.linear_t(g1, c0.weight.detach(), torch.empty((N, Cp * 49), dtype=torch.float32, device=dev))
        gpool = torch.empty((N, Hp, Wp, Cp), dtype=ops.tdtype(dt), device=dev)
        check(L.dge_adaptive_pool7_bwd(_f32(gflat), _p(gpool), N, Hp, Wp, Cp, dt, _stream()), "dge_adaptive_pool7_bwd")
        last = st["acts"][-1]
        return self._pool_relu_bwd(gpool, last) if self.plan[-1][0] == "pool" else self._relu_bwd(gpool, last)

    def backward_to_input(self, st, gpre_last):
        """Continues the (guided) backward from the last conv's output down to the image: [N,3,H,W] f32."""
        dt, N, H, W = st["dt"], st["N"], st["H"], st["W"]
        acts = st["acts"]
        g = gpre_last
        ci = len(acts) - 1
        k = len(self.plan) - 1
        while self.plan[k][0] != "conv":
            k -= 1
        while True:
            p = self.plan[k]                                  # conv ci
            cin = _CPAD if p[2] == 3 else p[2]
            g = ops.conv2d(g, self._packed(p[1], dt, ops.PACK_DGRAD), cin, 3)
            if ci == 0:
                break
            k -= 1
            pooled = self.plan[k][0] == "pool"               # conv ci read the pooled output of conv ci-1
            if pooled:
                k -= 1
            ci -= 1
            g = self._pool_relu_bwd(g, acts[ci]) if pooled else self._relu_bwd(g, acts[ci])
        gimg = torch.empty((N, 3, H, W), dtype=torch.float32, device=g.device)
        one = (C.c_float * 3)(1, 1, 1)
        check(lib().dge_lpips_prep_bwd(_p(g), _p(gimg), N, H * W, _CPAD, one, 1.0, 0, dt, _stream()), "dge_lpips_prep_bwd")
        return gimg


# ---------------------------------------------------------------------------------------------- the reference's callables
class GradCAM(object):
    """grad_cam.py:11-116 (`__call__`: one backward for the whole batch, :84-116)."""
    _MODE = 0

    def __init__(self, net, layer_name):
        if not isinstance(net, VGG16):
            raise DgeError("GradCAM runs on dge_amd.grad_cam.VGG16 (torchvision vgg16 layout); there is no generic-module path")
        if layer_name != net.final_layer:
            raise ValueError(f"only the last Conv2d ({net.final_layer}) is exposed, as the reference uses it (got {layer_name})")
        self.net = net
        self.layer_name = layer_name
        self.feature = None          # NHWC [N,h,w,C]: post-ReLU (in-place quirk)
        self.gradient = None         # NHWC [N,h,w,C]
        self.index = None            # int32 device tensor [1+N]: index_max, then the per-row class ids
        self.net.eval()

    def remove_handlers(self):
        pass

    def _mask(self, st):
        N, h, w, Cc = self.feature.shape
        dev = self.feature.device
        wgt = torch.empty((N, Cc), dtype=torch.float32, device=dev)
        cam = torch.empty((N, h * w), dtype=torch.float32, device=dev)
        mm = torch.empty((N, 2), dtype=torch.float32, device=dev)
        check(lib().dge_campp_map(_p(self.gradient), _p(self.feature), _f32(wgt), _f32(cam), _f32(mm), N, h * w, Cc, self._MODE,
                                  st["dt"], _stream()), "dge_campp_map")
        mask = torch.empty((N, 1, st["H"], st["W"]), dtype=torch.float32, device=dev)
        check(lib().dge_cam_resize(_f32(cam), _f32(mm), _f32(mask), N, h, w, st["H"], st["W"], _stream()), "dge_cam_resize")
        return mask

    def __call__(self, inputs, index):
        logits, st = self.net.run(inputs)
        self.index, _ = self.net.select_target(logits, index)
        self.gradient = self.net.backward_to_last_conv(st, self.index)
        self.feature = st["acts"][-1]
        return self._mask(st)

    def with_input_gradient(self, inputs, index=None):
        """(mask, d target / d inputs) from ONE forward and ONE backward.  The script calls `grad_cam_plus_plus(imgs, None)`
        and `gbp(imgs_)` on the same images (E_mis_align_cropping_s1.py:159-168): two forward and two backward passes of
        the same network towards the same target, the first backward being a prefix of the second.  Needs the guided
        network (a GuidedBackPropagation object constructed on it), as the input gradient is the guided one."""
        if not self.net.guided:
            raise DgeError("with_input_gradient: construct GuidedBackPropagation(net) first (the input gradient is the guided one)")
        logits, st = self.net.run(inputs)
        self.index, _ = self.net.select_target(logits, index)
        self.gradient = self.net.backward_to_last_conv(st, self.index)
        self.feature = st["acts"][-1]
        return self._mask(st), self.net.backward_to_input(st, self.gradient)


class GradCamPlusPlus(GradCAM):
    """grad_cam.py:119-194 (`__call__` :157-194)."""
    _MODE = 1


class GuidedBackPropagation(object):
    """grad_cam.py:196-232.  Constructing it switches the network's ReLU backward to the guided form for every later
    backward through that network (the reference registers permanent hooks on the shared module)."""

    def __init__(self, net):
        if not isinstance(net, VGG16):
            raise DgeError("GuidedBackPropagation runs on dge_amd.grad_cam.VGG16")
        self.net = net
        net.guided = True
        self.net.eval()

    def __call__(self, inputs, index=None):
        logits, st = self.net.run(inputs)
        idx, _ = self.net.select_target(logits, index)
        return self.net.backward_to_input(st, self.net.backward_to_last_conv(st, idx))


def _jet_table():
    """OpenCV COLORMAP_JET as 256 (r, g, b) byte triples: 64 knots of the classic jet ramps, linearly interpolated."""
    up = [i / 16.0 for i in range(1, 17)]
    down = [1 - i / 16.0 for i in range(1, 17)]
    r = [0.0] * 24 + up + [1.0] * 16 + down[:8]
    g = [0.0] * 8 + up + [1.0] * 16 + down + [0.0] * 8
    b = up[8:] + [1.0] * 16 + down + [0.0] * 24
    x, xi = np.linspace(0.0, 1.0, 64), np.linspace(0.0, 1.0, 256)
    return np.rint(np.stack([np.interp(xi, x, np.array(c)) for c in (r, g, b)], axis=1) * 255.0).astype(np.int32)


_JET = {}


def mask2cam(mask, imgs):
    """grad_cam.py:234-251: mask [N,1,H,W], imgs [N,3,H,W] -> (heatmap, cam), both [N,3,H,W] f32 on the device."""
    N, _, H, W = imgs.shape
    dev = imgs.device
    if dev not in _JET:
        _JET[dev] = torch.from_numpy(_jet_table()).to(dev).contiguous()
    mask = mask.detach().to(torch.float32).contiguous()
    imgs = imgs.detach().to(torch.float32).contiguous()
    heat, cam = torch.empty_like(imgs), torch.empty_like(imgs)
    nblk = lib().dge_mask2cam_blocks(H * W)
    part = torch.empty((N, nblk, 3), dtype=torch.float32, device=dev)
    coef = torch.empty((N, 2), dtype=torch.float32, device=dev)
    check(lib().dge_mask2cam(_f32(mask), _f32(imgs), _p(_JET[dev]), _f32(heat), _f32(cam), _f32(part), _f32(coef), N, H * W,
                             _stream()), "dge_mask2cam")
    return heat, cam
