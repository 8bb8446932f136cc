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


class _NoiseFeeder:
    """torch.randn inside the reference encoder -> the recipe's tensors (as tools/gen_golden.py does)."""

    def __init__(self, prefix, seed):
        self.prefix, self.seed, self.i = prefix, seed, 0
        self.orig = torch.randn

    def __call__(self, *size, **kw):
        if len(size) == 1 and isinstance(size[0], (list, tuple)):
            size = tuple(size[0])
        t = R.randn(f"{self.prefix}.noise{self.i}", size, self.seed)
        self.i += 1
        return t

    def __enter__(self):
        torch.randn = self
        return self

    def __exit__(self, *a):
        torch.randn = self.orig


def gen_step_misalign():
    """Two iterations of the loop body of E_mis_align_cropping_s1.py:109-206 (mtype 2) with the reference's own modules at
    the reduced size of step_s2.npz (same G / E / z / noise recipes), the narrow VGG16 stand-in for Grad-CAM and the
    seeded LPIPS stand-in.  As shipped, the image phase backpropagates only into LPIPS's `lin` weights (every image-space
    loss is built from detached tensors, :175-191): E is trained by the latent phase alone."""
    from model.stylegan2_generator import StyleGAN2Generator
    import model.E.E as EE
    tv = sys.modules["torchvision"]
    tv.transforms = types.ModuleType("torchvision.transforms")
    tv.transforms.Compose = lambda x: None
    tv.transforms.ToTensor = lambda: None
    sys.modules["torchvision.transforms"] = tv.transforms
    import training_utils as TU
    from model.utils.custom_adam import LREQAdam
    from oracle import lpips_ref as LR
    from tests.helpers import s2_shapes, enc_shapes  # noqa: F401

    shapes_of = lambda sd: {k: list(v.shape) for k, v in sd.items()}
    G = StyleGAN2Generator(64, fmaps_base=2048, fmaps_max=128)
    G.load_state_dict(R.fill_s2(shapes_of(G.state_dict()), seed=11))
    E = EE.BE(startf=16, maxf=64, layer_count=5)
    E.load_state_dict(R.fill_encoder(shapes_of(E.state_dict()), seed=31))
    LP = LR.seeded_params(0)
    for k, v in LP.items():
        if k.startswith("lin"):
            v.requires_grad_(True)                  # lpips.LPIPS(net='vgg'): the lin layers are trainable parameters
    lp = lambda a, b: LR.lpips(LP, a, b)
    cfg = R.GRADCAM_CFG
    net = GR.VGG16Ref(cfg["widths"], cfg["fc"], cfg["classes"])
    net.load_state_dict(GR.seeded_state(shapes_of(net.state_dict()), cfg["seed"]))
    final_layer = [n for n, m in net.named_modules() if isinstance(m, nn.Conv2d)][-1]
    gcpp = GradCamPlusPlus(net, final_layer)
    gbp = GuidedBackPropagation(net)
    opt = LREQAdam([{"params": E.parameters()}], lr=0.0015, betas=(0.0, 0.99), weight_decay=0)
    out = {}
    B = 2
    new_z = R.randn("step.new_z", (B, 512), 1)
    orig_randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: new_z.clone()
    flat = lambda inf: [inf[0][0], inf[0][1], inf[0][2], inf[1], inf[2], inf[3], inf[4]]
    try:
        for it in range(2):
            np.random.seed(it)
            z = R.randn(f"step.z{it}", (B, 512), 1)
            with torch.no_grad():
                r = G(z, trunc_psi=0.7, trunc_layers=8, randomize_noise=False)
            imgs1, w1 = r["image"], r["wp"]
            with _NoiseFeeder(f"step.it{it}", 1):
                const2, w2 = E(imgs1)
            imgs2 = G.synthesis(w2)["image"]
            opt.zero_grad()
            with contextlib.redirect_stdout(io.StringIO()):
                mask_1 = gcpp(imgs1, None)
                mask_2 = gcpp(imgs2, None)
                imgs1_ = imgs1.detach().clone()
                imgs1_.requires_grad = True
                imgs2_ = imgs2.detach().clone()
                imgs2_.requires_grad = True
                grad_1 = gbp(imgs1_)
                grad_2 = gbp(imgs2_)
            heat_1, cam_1 = mask2cam(mask_1, imgs1)
            heat_2, cam_2 = mask2cam(mask_2, imgs2)
            l_grad, i_grad = TU.space_loss(grad_1, grad_2, lpips_model=lp)
            l_imgs, i_imgs = TU.space_loss(imgs1.detach().clone(), imgs2.detach().clone(), lpips_model=lp)
            m1, m2 = mask_1.float(), mask_2.float()
            l_mask, i_mask = TU.space_loss(m1.detach().clone(), m2.detach().clone(), lpips_model=lp)
            c1, c2 = cam_1.float(), cam_2.float()
            l_cam, i_cam = TU.space_loss(c1.detach().clone(), c2.detach().clone(), lpips_model=lp)
            loss_tsa = l_imgs + l_mask + l_cam
            opt.zero_grad()
            loss_tsa.backward(retain_graph=True)
            assert all(p.grad is None for p in E.parameters())        # the image phase does not reach E
            opt.step()
            l_w, i_w = TU.space_loss(w1, w2, image_space=False)
            loss_mtv = l_w * 0.01
            opt.zero_grad()
            loss_mtv.backward()
            opt.step()
            out[f"it{it}_w2"] = w2.detach()
            out[f"it{it}_mask_1"] = m1
            out[f"it{it}_mask_2"] = m2
            out[f"it{it}_losses"] = np.array([float(loss_tsa), float(l_imgs), float(l_mask), float(l_cam), float(l_grad), float(l_w)])
            out[f"it{it}_info"] = np.array([flat(i_imgs), flat(i_mask), flat(i_cam), flat(i_grad), flat(i_w)])
            out[f"it{it}_param_checksum"] = np.array(R.checksum(E.state_dict()))
            for k in ("decode_block.0.conv_1.weight", "decode_block.2.conv_2.weight", "decode_block.4.inver_mod2.weight",
                      "decode_block.1.bias_1", "FromRGB.from_rgb.weight"):
                out[f"it{it}_after:{k}"] = E.state_dict()[k].clone()
            print("it", it, out[f"it{it}_losses"])
    finally:
        torch.randn_like = orig_randn_like
    path = os.path.join(OUT, "step_misalign.npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print("wrote step_misalign.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    todo = sys.argv[1:] or ["maps", "step"]
    if "maps" in todo:
        main()
    if "step" in todo:
        gen_step_misalign()
