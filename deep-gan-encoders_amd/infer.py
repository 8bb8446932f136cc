"""Inference-side helpers of the reference on the HIP path (thin wrappers over the same forward kernels):

* `reconstruct`   - inferE.py:101-141 / rec_real_img.py / synthesized_IMG.py: G(z) -> E -> G(w2) once, no gradients;
* `edit_latent`   - embeded_img_edit.py:28-41: w[start:start+end] = (w + bonus*direction)[start:start+end] on a W+ code,
                    then one Gs.forward(w, lod);
* `image_metrics` - comparing-baseline.py:21-45 on device: PSNR / MSE on [0,255], cosine on [-1,1], LPIPS and the skimage SSIM of
                    that script (`ssim_skimage`: 7x7 uniform window, sample covariance, data_range 255 - a different statistic
                    from the training loss's pytorch_ssim, which `losses.space_loss` reports);
* `save_image`    - torchvision.utils.save_image(img*0.5+0.5, path) for a single image batch laid out in one row.
"""
import math

import torch

from . import ops
from ._lib import lib, check
from .ops import _f32, _p, _stream


@torch.no_grad()
def reconstruct(step, z=None, iteration=4):
    """One inversion round trip with the models of an `EAlignStep` (any --mtype): returns dict(imgs1, w1, const2, w2, imgs2)."""
    from .e_align import set_seed, _BigGANAdapter
    gen, E, B = step.gen, step.E, step.batch_size
    big = isinstance(gen, _BigGANAdapter)
    if z is None:
        set_seed(iteration)                                   # inferE.py:101-103 (seed 4)
        z = gen.draw(iteration, B, step.dev) if big else torch.randn(B, step.z_dim)
    z = z.to(step.dev)
    imgs1, w1 = gen.sample(z)
    const2, w2 = E(imgs1, gen.const1) if big else E(imgs1)
    return dict(imgs1=imgs1, w1=w1, const2=const2, w2=w2, imgs2=gen.synth(w2))


def edit_latent(w, direction, bonus=70.0, start=0, end=3):
    """embeded_img_edit.py:28-41.  w: [1,L,512] or [L,512] W+ code, direction: [1,512] (InterfaceGAN boundary).  Rows
    start .. start+end-1 move along the direction, the others keep the identity features.  Returns [1,L,512]."""
    w2 = w.detach().clone().float()
    w2 = w2.squeeze(0) if w2.dim() == 3 else w2
    d = torch.as_tensor(direction).float().to(w2.device).expand(w2.shape[0], w2.shape[1])
    w2[start:start + end] = (w2 + bonus * d)[start:start + end]
    return w2.reshape(1, w2.shape[0], w2.shape[1])


@torch.no_grad()
def image_metrics(img1, img2, lpips_model=None):
    """img1, img2: [B,3,H,W] in [-1,1] on the GPU -> dict of device scalars: psnr / mse on the [0,255] scale, cosine on
    [-1,1], lpips (mean over the batch) when a model is given.  One reduction pass (dge_loss_reduce)."""
    a = img1.detach().float().contiguous()
    b = img2.detach().float().contiguous()
    B, Cc, H, W = a.shape
    slots = ops.zeros((16, 8), a.device)
    check(lib().dge_loss_reduce(_f32(a), _f32(b), _p(slots), B, Cc, H, W, 0, 0, H, W, _stream()), "dge_loss_reduce")
    s = ops._sum_over_batch(slots)
    n = float(a.numel())
    mse255 = s[0] / n * (127.5 * 127.5)                      # x255 = (x + 1) * 127.5
    out = dict(mse=mse255, psnr=10.0 * torch.log10(255.0 * 255.0 / mse255), cosine=s[1] / torch.sqrt(s[2] * s[3]))
    out["ssim"] = ssim_skimage(a, b).mean()
    if lpips_model is not None:
        out["lpips"], _ = lpips_model.value_and_grad(a, b, need_grad=False)
    return out


@torch.no_grad()
def ssim_skimage(img1, img2):
    """skimage.measure.compare_ssim(x, y, data_range=255, multichannel=True) of comparing-baseline.py:25 for every image pair of
    a batch: img1, img2 [B,C,H,W] in [-1,1] on the GPU (taken to the script's [0,255] scale inside the kernel) -> [B]."""
    a = img1.detach().float().contiguous()
    b = img2.detach().float().contiguous()
    B, Cc, H, W = a.shape
    if H < 7 or W < 7:
        raise ValueError("win_size exceeds image extent (skimage raises the same for images smaller than 7x7)")
    sums = torch.zeros(B * Cc, dtype=torch.float32, device=a.device)
    check(lib().dge_ssim_box7(_f32(a), _f32(b), _p(sums), B * Cc, H, W, 127.5, 127.5, 255.0, _stream()), "dge_ssim_box7")
    return sums.view(B, Cc).sum(1) / float(Cc * (H - 6) * (W - 6))


def save_image(img, path):
    """img [B,3,H,W] in [-1,1] -> PNG (samples side by side)."""
    from PIL import Image
    x = (img.detach().float().cpu() * 0.5 + 0.5).clamp(0, 1)
    x = torch.cat(list(x), dim=2)                              # [3,H,B*W]
    Image.fromarray((x.permute(1, 2, 0) * 255.0 + 0.5).to(torch.uint8).numpy()).save(path)
