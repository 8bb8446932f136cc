"""ORACLE (test infrastructure only) - CPU restatement of LPIPS(net='vgg').

Third-party algorithm, NOT present under /root/reference: the reference calls
`lpips.LPIPS(net='vgg')` (E_align_s2.py:98; training_utils.py:93; requirements.txt:12, unpinned)
from richzhang/PerceptualSimilarity (v0.1 linear heads) on torchvision's VGG16.  Neither the
package nor any weights exist in this image, so PARITY IS UNPINNED here: this file restates
the published algorithm (Zhang et al., CVPR 2018: scaling layer, VGG16 relu1_2 / 2_2 / 3_3 /
4_3 / 5_3 taps, channel-unit-normalisation, squared difference, non-negative 1x1 `lin`, spatial
mean, sum over taps) and is exercised with seeded stand-in weights of the right shapes.
"""
import torch
import torch.nn.functional as F

VGG_CFG = [(3, 64), (64, 64), "M", (64, 128), (128, 128), "M", (128, 256), (256, 256), (256, 256), "M",
           (256, 512), (512, 512), (512, 512), "M", (512, 512), (512, 512), (512, 512)]
# torchvision `features` indices of the convs, and the conv index after which each tap sits
CONV_IDX = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]
SLICE_OF_CONV = [1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5]
TAP_AFTER_CONV = [1, 3, 6, 9, 12]
TAP_CH = [64, 128, 256, 512, 512]
SHIFT = (-0.030, -0.088, -0.188)
SCALE = (0.458, 0.448, 0.450)


def param_shapes():
    s = {"scaling_layer.shift": [1, 3, 1, 1], "scaling_layer.scale": [1, 3, 1, 1]}
    ci = 0
    for item in VGG_CFG:
        if item == "M":
            continue
        cin, cout = item
        s[f"net.slice{SLICE_OF_CONV[ci]}.{CONV_IDX[ci]}.weight"] = [cout, cin, 3, 3]
        s[f"net.slice{SLICE_OF_CONV[ci]}.{CONV_IDX[ci]}.bias"] = [cout]
        ci += 1
    for k, c in enumerate(TAP_CH):
        s[f"lin{k}.model.1.weight"] = [1, c, 1, 1]
    return s


def seeded_params(seed=0):
    """Stand-in weights (He-scaled convs, non-negative lins) - compute-identical to the real ones."""
    g = torch.Generator().manual_seed(1234 + seed)
    P = {}
    for k, shp in param_shapes().items():
        if k == "scaling_layer.shift":
            P[k] = torch.tensor(SHIFT).view(1, 3, 1, 1)
        elif k == "scaling_layer.scale":
            P[k] = torch.tensor(SCALE).view(1, 3, 1, 1)
        elif k.endswith(".bias"):
            P[k] = torch.randn(shp, generator=g) * 0.05
        elif k.startswith("lin"):
            P[k] = torch.randn(shp, generator=g).abs() * 0.1
        else:
            fan_in = shp[1] * 9
            P[k] = torch.randn(shp, generator=g) * (2.0 / fan_in) ** 0.5
    return P


def features(P, x):
    taps = []
    ci = 0
    for item in VGG_CFG:
        if item == "M":
            x = F.max_pool2d(x, 2, 2)
            continue
        name = f"net.slice{SLICE_OF_CONV[ci]}.{CONV_IDX[ci]}"
        x = F.relu(F.conv2d(x, P[name + ".weight"], P[name + ".bias"], padding=1))
        if ci in TAP_AFTER_CONV:
            taps.append(x)
        ci += 1
    return taps


def lpips(P, a, b):
    """Returns [B,1,1,1] like lpips.LPIPS.forward (normalize=False: inputs already in [-1,1])."""
    sa = (a - P["scaling_layer.shift"]) / P["scaling_layer.scale"]
    sb = (b - P["scaling_layer.shift"]) / P["scaling_layer.scale"]
    fa, fb = features(P, sa), features(P, sb)
    total = 0
    for k in range(5):
        na = fa[k] / (fa[k].pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        nb = fb[k] / (fb[k].pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        d = (na - nb) ** 2
        total = total + F.conv2d(d, P[f"lin{k}.model.1.weight"]).mean(dim=(2, 3), keepdim=True)
    return total
