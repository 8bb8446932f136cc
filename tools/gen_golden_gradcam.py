#!/usr/bin/env python3
"""Golden vectors for the Grad-CAM++ attention path from the reference's own classes.

Runs ONLY in the build container (needs /root/reference; a no-op elsewhere).  Imports metric/grad_cam.py of the
reference (never copies it) with `cv2` and `torchvision` stubbed - neither is installed here: `cv2.resize` and
`cv2.applyColorMap` are the restatements in oracle/gradcam_ref.py, the network is oracle.gradcam_ref.VGG16Ref
(torchvision's vgg16 layout, narrow widths, seeded weights).  Wiring follows E_mis_align_cropping_s1.py:99-106,
159-170: GradCamPlusPlus and GuidedBackPropagation share ONE network.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_gradcam.py
"""
import contextlib
import io
import os
import sys
import types

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
if not os.path.isdir(REF):
    print("reference not present; nothing to do")
    sys.exit(0)
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch
from torch import nn

from oracle import gradcam_ref as GR
from tests.golden import recipe as R

cv2 = types.ModuleType("cv2")
cv2.resize = lambda src, dsize: GR.cv2_resize_linear(src, dsize)
cv2.applyColorMap = GR.cv2_apply_colormap
cv2.COLORMAP_JET = GR.COLORMAP_JET
sys.modules["cv2"] = cv2
sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))

import warnings
warnings.filterwarnings("ignore")
from metric.grad_cam import GradCAM, GradCamPlusPlus, GuidedBackPropagation, mask2cam  # noqa: E402


def main():
    cfg = R.GRADCAM_CFG
    net = GR.VGG16Ref(cfg["widths"], cfg["fc"], cfg["classes"])
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    sd = GR.seeded_state(shapes, cfg["seed"])
    net.load_state_dict(sd)
    final_layer = None
    for name, m in net.named_modules():                     # E_mis_align_cropping_s1.py:101-104
        if isinstance(m, nn.Conv2d):
            final_layer = name
    gcpp = GradCamPlusPlus(net, final_layer)
    gbp = GuidedBackPropagation(net)
    gc_plain = GradCAM(net, final_layer)
    N, H, W = cfg["N"], cfg["H"], cfg["W"]
    imgs1 = R.gradcam_images("a", N, H, W)
    imgs2 = R.gradcam_images("b", N, H, W)
    out = {"final_layer": np.array(final_layer)}
    with contextlib.redirect_stdout(io.StringIO()):
        for tag, imgs in (("1", imgs1), ("2", imgs2)):
            mask = gcpp(imgs, None)                         # the network's own parameters require grad (as in the script)
            out["mask_" + tag] = mask.numpy().astype(np.float32)
            out["feature_" + tag] = gcpp.feature.detach().numpy().copy()
            out["gradient_" + tag] = gcpp.gradient.detach().numpy().copy()
            out["logits_" + tag] = net(imgs).detach().numpy()
            x_ = imgs.detach().clone()
            x_.requires_grad = True
            out["gbp_" + tag] = gbp(x_).detach().numpy().copy()
            heat, cam = mask2cam(mask, imgs)
            out["heat_" + tag] = heat.numpy()
            out["cam_" + tag] = cam.numpy()
            # explicit class index path (`index` given)
            idx = np.array([3] * N)
            out["mask_idx_" + tag] = gcpp(imgs, idx).numpy().astype(np.float32)
            out["mask_plain_" + tag] = gc_plain(imgs, None).numpy().astype(np.float32)
    # restatement vs the reference's classes (pins the oracle)
    for tag, imgs in (("1", imgs1), ("2", imgs2)):
        m, index_max, logits, grad, post = GR.grad_cam_pp(sd, imgs, None, guided=True)
        np.testing.assert_allclose(logits.numpy(), out["logits_" + tag], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(post.numpy(), out["feature_" + tag], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(grad.numpy(), out["gradient_" + tag], rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(m.numpy(), out["mask_" + tag], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(GR.grad_cam_pp(sd, imgs, None, plain=True)[0].numpy(), out["mask_plain_" + tag], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(GR.grad_cam_pp(sd, imgs, np.array([3] * N))[0].numpy(), out["mask_idx_" + tag], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(GR.guided_backprop(sd, imgs).numpy(), out["gbp_" + tag], rtol=1e-4, atol=1e-9)
        h, c = GR.mask2cam(torch.tensor(out["mask_" + tag]).double(), imgs)
        np.testing.assert_allclose(h.numpy(), out["heat_" + tag], atol=1e-6)
        np.testing.assert_allclose(c.numpy(), out["cam_" + tag], atol=1e-5)
        out["index_max_" + tag] = np.array(index_max)
        srt = np.sort(out["logits_" + tag], axis=1)
        print(tag, "index_max", index_max, "argmax", out["logits_" + tag].argmax(1), "margin", (srt[:, -1] - srt[:, -2]))
    for k in [k for k in out if k.startswith("feature_")]:
        del out[k]                                         # large; reproduced by the oracle, not needed as a fixture
    path = os.path.join(OUT, "gradcam.npz")
    np.savez_compressed(path, **out)
    print("wrote gradcam.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
