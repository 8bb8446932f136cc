"""The `E_mis_align_cropping_s1.py` iteration on the HIP path (reference :109-206).

What the shipped loop does, and what is kept: G -> E -> G as in E_align_s2; Grad-CAM++ masks of both image batches,
guided-back-propagation gradients, JET overlays (metric/grad_cam.py); four *logged* image-space `space_loss`
evaluations (gradients, images, masks, overlays).  Every one of them is built from detached tensors (:175-191), so
`loss_tsa.backward()` reaches only LPIPS's own `lin` weights and the first `E_optimizer.step()` finds no gradient:
the encoder is trained by the latent phase (`0.01 * loss_w`, :199-205) alone.  The second synthesis therefore runs
without a gradient tape here.  `loss_c` (:197, logged only) is not produced.
"""
import torch

from . import losses
from .e_align import EAlignStep, set_seed, _BigGANAdapter
from .grad_cam import GradCamPlusPlus, GuidedBackPropagation, mask2cam


class MisAlignStep(EAlignStep):
    def __init__(self, generator, E, lpips_model, vgg16, fused_attention=True, **kw):
        """`vgg16`: dge_amd.grad_cam.VGG16 (torchvision vgg16 layout), shared by Grad-CAM++ and guided back-propagation
        as in E_mis_align_cropping_s1.py:99-106."""
        super().__init__(generator, E, lpips_model, **kw)
        self.grad_cam_plus_plus = GradCamPlusPlus(vgg16, vgg16.final_layer)
        self.gbp = GuidedBackPropagation(vgg16)
        self.fused_attention = fused_attention

    def step(self, iteration, z=None, noises=None, gen_noises=(None, None), new_z=None):
        E = self.E
        B = self.batch_size
        from . import ops
        from .e_align import _StyleGAN2Adapter
        if isinstance(self.gen, _StyleGAN2Adapter):
            self.gen.new_z = new_z
        ops.zero_arena_begin(self.dev)
        if z is None or not z.is_cuda:
            set_seed(iteration % 30000)
        big = isinstance(self.gen, _BigGANAdapter)
        if z is None:
            zg = self.gen.draw(iteration, B * self.world, self.dev) if big else torch.randn(B * self.world, self.z_dim)
            z = zg[self.rank * B:(self.rank + 1) * B]
        z = self._upload(z)
        with torch.no_grad():
            imgs1, w1 = self.gen.sample(z, gen_noises[0])
        if noises is None and self.reference_noise:
            from .autograd_enc import draw_noises
            noises = [n.to(self.dev) for n in draw_noises(E, B, imgs1.shape[2], "cpu")]
        const2, w2 = E(imgs1, self.gen.const1, noises=noises) if big else E(imgs1, noises=noises)
        with torch.no_grad():
            imgs2 = self.gen.synth(w2.detach(), gen_noises[1])
            # attention maps (:159-170)
            if self.fused_attention:       # one forward + one backward per batch instead of two of each (same results)
                mask_1, grad_1 = self.grad_cam_plus_plus.with_input_gradient(imgs1)
                mask_2, grad_2 = self.grad_cam_plus_plus.with_input_gradient(imgs2)
            else:
                mask_1 = self.grad_cam_plus_plus(imgs1, None)
                mask_2 = self.grad_cam_plus_plus(imgs2, None)
                grad_1 = self.gbp(imgs1)
                grad_2 = self.gbp(imgs2)
            heat_1, cam_1 = mask2cam(mask_1, imgs1)
            heat_2, cam_2 = mask2cam(mask_2, imgs2)
            gctx = losses.GlobalBatch(self.world) if (self.dist_on and self.exact_ddp) else None
            # logged image-space terms (:172-191); values only
            _, info_grad = losses.space_loss(grad_1, grad_2, lpips_model=self.lpips, global_batch=gctx)
            l_imgs, info_imgs = losses.space_loss(imgs1, imgs2, lpips_model=self.lpips, global_batch=gctx)
            l_mask, info_mask = losses.space_loss(mask_1, mask_2, lpips_model=self.lpips, global_batch=gctx)
            l_cam, info_cam = losses.space_loss(cam_1, cam_2, lpips_model=self.lpips, global_batch=gctx)
            loss_tsa = l_imgs + l_mask + l_cam
        # latent phase (:199-205)
        gctx = losses.GlobalBatch(self.world) if (self.dist_on and self.exact_ddp) else None
        loss_w, info_w = losses.space_loss(w1, w2, image_space=False, global_batch=gctx)
        loss_mtv = loss_w * 0.01
        self.opt.zero_grad()
        loss_mtv.backward()
        gs = self._sync_grads()
        self.opt.step(grad_scale=gs)
        ops.zero_arena_end()
        self.last = dict(imgs1=imgs1, imgs2=imgs2, w1=w1, w2=w2.detach(), const2=const2, mask_1=mask_1, mask_2=mask_2,
                         grad_1=grad_1, grad_2=grad_2, heatmap_1=heat_1, heatmap_2=heat_2, cam_1=cam_1, cam_2=cam_2,
                         loss_tsa=loss_tsa, info_imgs=info_imgs, info_mask=info_mask, info_Gcam=info_cam, info_grad=info_grad,
                         loss_w=loss_w.detach(), info_w=info_w)
        return self.last
