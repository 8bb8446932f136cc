"""ORACLE (test infrastructure only) - CPU restatement of the reference's Grad-CAM++ attention path
(SURVEY 8(f) row 1): metric/grad_cam.py `GradCamPlusPlus.__call__` :157-194, `GuidedBackPropagation`
:196-232, `mask2cam` :234-251, as E_mis_align_cropping_s1.py:99-106,159-170 wires them: ONE vgg16 carries both
the Grad-CAM++ hooks (forward/backward hook on the last Conv2d, `features.28`) and the guided-back-propagation
hooks (`clamp(grad_in, min=0)` on every nn.ReLU), so the Grad-CAM++ gradient is a *guided* gradient too.

Two facts of the reference that the restatement keeps (both probed by running the reference's classes):
  * torchvision's VGG uses nn.ReLU(inplace=True): the tensor the forward hook stored for `features.28` is
    overwritten by `features.29`, so `self.feature` holds the POST-ReLU activation;
  * the Grad-CAM++ weight sum(relu(g) * 1[g>0] / sum(relu(g))) is 1 for every channel with any positive
    gradient and 0 otherwise (up to float rounding).

Pinning: tools/gen_golden_gradcam.py runs the reference's own classes on `VGG16Ref` (below; torchvision's layout
with seeded stand-in weights - torchvision and its pretrained weights are absent from this image, so parity with
the real classifier is STRUCTURAL) and stores their outputs in tests/golden/gradcam.npz.  `cv2` is absent too:
`cv2.resize` (INTER_LINEAR) and `cv2.applyColorMap(COLORMAP_JET)` are restated here from OpenCV's published
definitions (imgproc/resize.cpp half-pixel centres with edge clamp; colormap.cpp Jet = 64-knot piecewise-linear
table interpolated to 256 entries) and handed to the reference as its `cv2` - that part is UNPINNED.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

VGG16_WIDTHS = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")


class VGG16Ref(nn.Module):
    """torchvision.models.vgg16 layout: features.{0..30} (Conv2d/ReLU(inplace)/MaxPool2d), avgpool =
    AdaptiveAvgPool2d(7), classifier.{0,3,6} Linear with ReLU(inplace)/Dropout between."""

    def __init__(self, widths=VGG16_WIDTHS, fc=4096, num_classes=1000):
        super().__init__()
        layers, cin = [], 3
        for w in widths:
            if w == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, w, 3, padding=1), nn.ReLU(inplace=True)]
                cin = w
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(nn.Linear(cin * 49, fc), nn.ReLU(True), nn.Dropout(),
                                        nn.Linear(fc, fc), nn.ReLU(True), nn.Dropout(), nn.Linear(fc, num_classes))

    def forward(self, x):
        x = self.features(x)
        x = self.avgpool(x)
        return self.classifier(torch.flatten(x, 1))


def seeded_state(shapes, seed=0):
    """He-scaled stand-in weights keyed like torchvision's state_dict."""
    from tests.golden import recipe as R
    out = {}
    for k, shp in shapes.items():
        if k.endswith(".bias"):
            out[k] = R.randn("vgg." + k, shp, seed, 0.05)
        else:
            fan_in = int(np.prod(shp[1:]))
            out[k] = R.randn("vgg." + k, shp, seed, float(np.sqrt(2.0 / fan_in)))
    return out


# ------------------------------------------------------------------------------------------ cv2 restatements
def cv2_resize_linear(src, dsize):
    """cv2.resize(src, (W, H)) with the default INTER_LINEAR for a single-channel float array.
    OpenCV: fx = (dx + 0.5) * (sw / dw) - 0.5; sx = floor(fx); fx -= sx; sx < 0 -> (0, 0);
    sx >= sw - 1 -> (sw - 1, 0); coefficients are float32 for float and double images alike."""
    src = np.asarray(src)
    W, H = dsize
    sh, sw = src.shape

    def taps(dn, sn):
        scale = sn / dn
        f = (np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5
        i0 = np.floor(f).astype(np.int64)
        a = (f - i0).astype(np.float32)
        lo = i0 < 0
        i0[lo] = 0
        a[lo] = 0
        hi = i0 >= sn - 1
        i0[hi] = sn - 1
        a[hi] = 0
        i1 = np.minimum(i0 + 1, sn - 1)
        return i0, i1, a

    y0, y1, ay = taps(H, sh)
    x0, x1, ax = taps(W, sw)
    s = src.astype(np.float64)
    ax = ax.astype(np.float64)[None, :]
    ay = ay.astype(np.float64)[:, None]
    top = s[y0][:, x0] * (1 - ax) + s[y0][:, x1] * ax
    bot = s[y1][:, x0] * (1 - ax) + s[y1][:, x1] * ax
    return (top * (1 - ay) + bot * ay).astype(src.dtype)


def jet_lut():
    """OpenCV COLORMAP_JET as a [256,3] uint8 table in B,G,R order."""
    up = [i / 16.0 for i in range(1, 17)]              # 0.0625 .. 1
    down = [1 - i / 16.0 for i in range(1, 17)]        # 0.9375 .. 0
    r = [0.0] * 24 + up + [1.0] * 16 + down[:8]
    g = [0.0] * 8 + up + [1.0] * 16 + down + [0.0] * 8
    b = up[8:] + [1.0] * 16 + down + [0.0] * 24
    x = np.linspace(0.0, 1.0, 64)
    xi = np.linspace(0.0, 1.0, 256)
    lut = np.stack([np.interp(xi, x, np.array(c)) for c in (b, g, r)], axis=1)
    return np.rint(lut * 255.0).astype(np.uint8)


COLORMAP_JET = 2


def cv2_apply_colormap(u8, colormap=COLORMAP_JET):
    assert colormap == COLORMAP_JET
    return jet_lut()[np.asarray(u8, dtype=np.uint8)]      # [..., 3] BGR


# ------------------------------------------------------------------------------------------ restatement
class _GuidedReLU(torch.autograd.Function):
    """nn.ReLU with GuidedBackPropagation.backward_hook (grad_cam.py:208-217): clamp(grad_in, min=0)."""

    @staticmethod
    def forward(ctx, x):
        y = x.clamp(min=0)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return (g * (y > 0)).clamp(min=0)


def _relu(x, guided):
    return _GuidedReLU.apply(x) if guided else F.relu(x)


def vgg_forward(sd, x, guided=True):
    """Returns (logits, pre-activation of the last conv, its post-ReLU value).  torchvision numbering: conv k,
    ReLU k+1, and a MaxPool2d at k+2 wherever the next conv sits at k+3 (and after the last conv)."""
    convs = sorted({int(k.split(".")[1]) for k in sd if k.startswith("features.")})
    pre = post = None
    for n, idx in enumerate(convs):
        pre = F.conv2d(x, sd[f"features.{idx}.weight"], sd[f"features.{idx}.bias"], padding=1)
        post = x = _relu(pre, guided)
        if n + 1 == len(convs) or convs[n + 1] == idx + 3:
            x = F.max_pool2d(x, 2, 2)
    x = F.adaptive_avg_pool2d(x, (7, 7)).flatten(1)
    x = _relu(F.linear(x, sd["classifier.0.weight"], sd["classifier.0.bias"]), guided)
    x = _relu(F.linear(x, sd["classifier.3.weight"], sd["classifier.3.bias"]), guided)
    return F.linear(x, sd["classifier.6.weight"], sd["classifier.6.bias"]), pre, post


def _target(logits, index):
    if index is None:
        index = np.argmax(logits.detach().numpy(), axis=1)           # grad_cam.py:166-167
    index_max = int(np.argmax(np.bincount(np.asarray(index))))       # :168
    return logits[:, index_max].mean(), index_max                   # :169-170


def grad_cam_pp(sd, inputs, index=None, guided=True, plain=False):
    """GradCamPlusPlus.__call__ (grad_cam.py:157-194); plain=True: GradCAM.__call__ (:84-116).  Returns (mask
    [N,1,H,W] f64, index_max, logits, gradient at the last conv [N,C,h,w], feature [N,C,h,w])."""
    logits, pre, post = vgg_forward(sd, inputs.detach().clone().requires_grad_(True), guided)
    target, index_max = _target(logits, index)
    (grad,) = torch.autograd.grad(target, pre)
    N, _, H, W = inputs.shape
    cam_all = np.zeros((N, H, W))
    for i in range(N):
        feature = post[i].detach().numpy()
        if plain:
            weight = np.mean(grad[i].numpy(), axis=(1, 2))                     # :101
            cam = np.maximum(np.sum(feature * weight[:, None, None], axis=0), 0)  # :103-105
        else:
            g = np.maximum(grad[i].numpy(), 0.0)                                # :177
            indicate = np.where(g > 0, 1.0, 0.0)
            norm = np.sum(g, axis=(1, 2))
            norm = np.where(norm > 0, 1.0 / np.where(norm > 0, norm, 1), 0.0).astype(np.float32)
            alpha = indicate * norm[:, None, None]
            weight = np.sum(g * alpha, axis=(1, 2))                             # :183
            cam = np.sum(feature * weight[:, None, None], axis=0)               # :185-186
        cam -= np.min(cam)
        cam /= np.max(cam)
        cam_all[i] = cv2_resize_linear(cam, (W, H))
    return torch.tensor(cam_all.reshape(N, 1, H, W)), index_max, logits.detach(), grad, post.detach()


def guided_backprop(sd, inputs, index=None):
    """GuidedBackPropagation.__call__ (grad_cam.py:219-232): d mean_n logits[n, index_max] / d inputs with every
    ReLU's input gradient clamped at zero."""
    x = inputs.detach().clone().requires_grad_(True)
    logits, _, _ = vgg_forward(sd, x, True)
    target, _ = _target(logits, index)
    (g,) = torch.autograd.grad(target, x)
    return g


def mask2cam(mask, imgs):
    """grad_cam.py:234-251, including its sequential normalisation: sample i is shifted by the minimum over the
    WHOLE `cam` array as it stands at that moment (samples < i already normalised, samples > i still the raw
    images) - `np.max(np.min(cam), 0)` is a no-op max over axis 0 of a scalar - then divided by its own maximum."""
    imgs = np.float32(imgs.detach().cpu().numpy())
    mask = mask.detach().cpu().numpy()
    heat = imgs.copy()
    cam = imgs.copy()
    for i in range(imgs.shape[0]):
        h = cv2_apply_colormap(np.uint8(255 * mask[i, 0]))
        h = (np.float32(h) / 255)[..., ::-1]
        h = np.transpose(h, (2, 0, 1))
        heat[i] = h
        cam[i] = h + imgs[i]
        cam[i] -= np.min(cam)
        cam[i] /= np.max(cam[i])
    return torch.tensor(heat), torch.tensor(cam)
