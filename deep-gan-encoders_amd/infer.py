"""Inference-side helpers of the reference on the HIP path (thin wrappers over the same forward kernels):

* `reconstruct`   - inferE.py:101-141 / rec_real_img.py / synthesized_IMG.py: G(z) -> E -> G(w2) once, no gradients;
* `edit_latent`   - embeded_img_edit.py:28-41: w[start:start+end] = (w + bonus*direction)[start:start+end] on a W+ code,
                    then one Gs.forward(w, lod);
* `image_metrics` - comparing-baseline.py:21-45 on device: PSNR / MSE on [0,255], cosine on [-1,1], LPIPS and the skimage SSIM of
                    that script (`ssim_skimage`: 7x7 uniform window, sample covariance, data_range 255 - a different statistic
                    from the training loss's pytorch_ssim, which `losses.space_loss` reports);
* `load_images` / `reconstruct_images` - rec_real_img.py:84-120: image files -> `[N,3,S,S]` batch in [-1,1] (PIL, the script's
                    Resize + ToTensor; `training_utils.imgPath2loader` with `bicubic=True`) -> E -> G, the "invert this image" path;
* `save_image`    - torchvision.utils.save_image(img*0.5+0.5, path) for a single image batch laid out in one row.
"""
import math

import torch

from . import ops
from ._lib import lib, check
from .ops import _f32, _p, _stream


@torch.no_grad()
def reconstruct(step, z=None, iteration=4, noises=None):
    """One inversion round trip with the models of an `EAlignStep` (any --mtype): returns dict(imgs1, w1, const2, w2, imgs2).
    `noises`: optional encoder noise tensors (the reference draws them on the CPU, model/E/E.py:60,73 - parity runs inject them)."""
    from .e_align import set_seed, _BigGANAdapter
    gen, E, B = step.gen, step.E, step.batch_size
    big = isinstance(gen, _BigGANAdapter)
    if z is None:
        set_seed(iteration)                                   # inferE.py:101-103 (seed 4)
        z = gen.draw(iteration, B, step.dev) if big else torch.randn(B, step.z_dim)
    z = z.to(step.dev)
    imgs1, w1 = gen.sample(z)
    const2, w2 = E(imgs1, gen.const1, noises=noises) if big else E(imgs1, noises=noises)
    return dict(imgs1=imgs1, w1=w1, const2=const2, w2=w2, imgs2=gen.synth(w2))


def load_images(paths, size, bicubic=False, device="cuda"):
    """rec_real_img.py:84-98: Image.open(p).convert('RGB') -> Resize((size, size)) (PIL bilinear, what torchvision's Resize does
    to a PIL image) -> ToTensor ([0,1], CHW) -> stacked, then the script's `* 2 - 1` (:101).  bicubic=True is
    training_utils.imgPath2loader (:11-15: `image.resize((size, size))`, PIL's default filter).  Host-side file I/O."""
    import numpy as np
    from PIL import Image
    out = []
    for p in paths:
        img = Image.open(p).convert("RGB")
        img = img.resize((size, size)) if bicubic else img.resize((size, size), Image.BILINEAR)
        out.append(torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0))
    return (torch.stack(out, dim=0) * 2 - 1).to(device)


@torch.no_grad()
def reconstruct_images(step, imgs1):
    """rec_real_img.py:100-112: real images [N,3,S,S] in [-1,1] -> E -> G, one image at a time as the script does (the
    encoder's instance statistics are per sample, so batching changes nothing but the noise draws).  Returns (w2, imgs2)."""
    from .e_align import _BigGANAdapter
    gen, E = step.gen, step.E
    if isinstance(gen, _BigGANAdapter):
        raise ValueError("BigGAN needs the class-conditional vector of the image (rec_real_img.py:104); use E(img, cond) directly")
    ws, outs = [], []
    for j in imgs1:
        _, w2 = E(j.unsqueeze(0))
        ws.append(w2)
        outs.append(gen.synth(w2))
    return torch.cat(ws), torch.cat(outs)


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


def save_image_grid(imgs, path, nrow=10):
    """torchvision.utils.save_image(imgs*0.5+0.5, path, nrow=nrow) without torchvision: `nrow` images per row, 2 px padding."""
    from PIL import Image
    x = (imgs.detach().float().cpu() * 0.5 + 0.5).clamp(0, 1)
    n, c, h, w = x.shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    pad = 2
    grid = torch.zeros(c, rows * (h + pad) + pad, cols * (w + pad) + pad)
    for i in range(n):
        r, q = divmod(i, cols)
        grid[:, pad + r * (h + pad):pad + r * (h + pad) + h, pad + q * (w + pad):pad + q * (w + pad) + w] = x[i]
    Image.fromarray((grid.permute(1, 2, 0) * 255.0 + 0.5).to(torch.uint8).numpy()).save(path)


def _step_from_args(args, device="cuda"):
    from .e_align import EAlignStep, load_models
    G, Gm, E, _ = load_models(args, device=device, lpips=False)
    G.eval()
    E.eval()
    return EAlignStep(G, E, None, batch_size=args.batch_size, z_dim=args.z_dim, mapping=Gm)


def main(argv=None):
    """Entry points of the reference's inference scripts on the HIP path (`python -m dge_amd.infer <command> ...`), every
    --mtype and its checkpoint container (e_align.load_models):

    * `infer` - inferE.py:101-141: seed 4, z -> G -> E -> G, writes <out>/v2ep<seed>.png (originals over reconstructions);
    * `rec`   - rec_real_img.py:84-127: every image of --img_dir -> E -> G, writes <out>/real/%05d_realimg.png and
                <out>/rec/%05d_mtv_rec.png;
    * `synth` - synthesized_IMG.py:97-145: for iteration in 30000 .. 30000 + --iterations: set_seed(iteration), z -> G -> E -> G,
                writes <out>/id<flag>_%05d.png (nrow 10: originals then reconstructions)."""
    import argparse
    import os
    from .e_align import add_model_args, set_seed, _BigGANAdapter
    parser = argparse.ArgumentParser(prog="dge_amd.infer", description=main.__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument("command", choices=("infer", "rec", "synth"))
    add_model_args(parser)
    parser.add_argument("--batch_size", type=int, default=5)
    parser.add_argument("--seed", type=int, default=4)
    parser.add_argument("--iterations", type=int, default=1)
    parser.add_argument("--img_dir", default=None, help="rec: directory of images (sorted by name)")
    parser.add_argument("--out", default="./result")
    args = parser.parse_args(argv)
    os.makedirs(args.out, exist_ok=True)
    step = _step_from_args(args)
    written = []
    if args.command == "infer":
        r = reconstruct(step, iteration=args.seed)
        path = os.path.join(args.out, "v2ep%d.png" % args.seed)
        save_image_grid(torch.cat((r["imgs1"][:args.batch_size], r["imgs2"][:args.batch_size])), path, nrow=args.batch_size)
        written.append(path)
    elif args.command == "rec":
        if not args.img_dir:
            parser.error("rec needs --img_dir")
        names = sorted(n for n in os.listdir(args.img_dir) if n.lower().endswith((".png", ".jpg", ".jpeg", ".bmp")))
        if not names:
            parser.error("rec: no images in " + args.img_dir)
        d1, d2 = os.path.join(args.out, "real"), os.path.join(args.out, "rec")
        os.makedirs(d1, exist_ok=True)
        os.makedirs(d2, exist_ok=True)
        imgs = load_images([os.path.join(args.img_dir, n) for n in names], args.img_size, device=step.dev)
        set_seed(args.seed)                                   # the encoder draws noise in every forward (model/E/E.py:62,72)
        _, rec = reconstruct_images(step, imgs)
        for i in range(imgs.shape[0]):
            a, b = os.path.join(d1, "%s_realimg.png" % str(i).rjust(5, "0")), os.path.join(d2, "%s_mtv_rec.png" % str(i).rjust(5, "0"))
            save_image_grid(imgs[i:i + 1], a, nrow=1)
            save_image_grid(rec[i:i + 1], b, nrow=1)
            written += [a, b]
    else:
        big = isinstance(step.gen, _BigGANAdapter)
        for iteration in range(30000, 30000 + args.iterations):
            r = reconstruct(step, iteration=iteration % 30000 if big else iteration)
            flag = getattr(step.gen, "flag", 0) if big else 0
            path = os.path.join(args.out, "id%s_%s.png" % (flag, str(iteration - 30000).rjust(5, "0")))
            save_image_grid(torch.cat((r["imgs1"], r["imgs2"])), path, nrow=10)
            written.append(path)
    for w in written:
        print(w)
    return written


if __name__ == "__main__":
    main()
