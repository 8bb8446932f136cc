import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tests.test_sg1 import small_params, golden, R
from oracle import ref_torch as O
import dge_amd.stylegan1 as S
import dge_amd.autograd_sg1 as A
g = golden("sg1_small.npz"); gg = golden("sg1_grad.npz")
G = S.Generator(startf=32, maxf=64, layer_count=6, latent_size=512, compute_dtype="f32").cuda()
P = small_params()
G.load_state_dict(P)
tag, lod, prefix, nn_ = ("_lod3", 3, "sg1b", 8)
noises = [R.randn(f"{prefix}.noise{i}", tuple(s), 6) for i, s in enumerate(g["noise_shapes"].tolist()[:nn_])]
# oracle with taps
U = []
orig = O.sg1_style_mod
def tap(x, style):
    u = orig(x, style); u.retain_grad(); U.append(u); return u
O.sg1_style_mod = tap
st = R.randn("sg1.styles", (2, 12, 512), 6).requires_grad_(True)
img = O.sg1_generator(P, st, lod, noises)
gimg = R.randn("sg1.gimg" + tag, tuple(img.shape), 7)
(img * gimg).sum().backward()
A.DEBUG_TAP = {}
styles = R.randn("sg1.styles", (2, 12, 512), 6).cuda().requires_grad_(True)
img2 = G.forward(styles, lod, noises=noises)
print("img err", float((img2.cpu() - img).abs().max()))
(img2.float() * gimg.cuda()).sum().backward()
for k in sorted([q for q in A.DEBUG_TAP if isinstance(q, int)], reverse=True):
    mine = A.DEBUG_TAP[k].float().cpu().permute(0, 3, 1, 2)
    ref = U[k].grad
    if ref.shape[0] != mine.shape[0]: ref = ref.expand_as(mine)
    d = (mine - ref)
    print(k, "rel", float(d.norm() / ref.norm()), "border rel", float(d[:, :, 0].norm() / ref[:, :, 0].norm()), "inner rel",
          float(d[:, :, 1:-1, 1:-1].norm() / (ref[:, :, 1:-1, 1:-1].norm() + 1e-30)), tuple(mine.shape))
i = 2
dots = A.DEBUG_TAP[("dots1", i)].cpu(); rec = A.DEBUG_TAP[("rec", i)]
yv = rec["y"].float().cpu().permute(0, 3, 1, 2)
gu = U[2 * i].grad
print("dots0 rel", float((dots[..., 0] - (gu * yv).sum((2, 3))).abs().max() / (gu * yv).sum((2, 3)).abs().max()),
      "dots1 rel", float((dots[..., 1] - gu.sum((2, 3))).abs().max() / gu.sum((2, 3)).abs().max()))
m = yv.mean((2, 3)); var = yv.var((2, 3), unbiased=False)
print("sc err", float((rec["sc1"].cpu() - torch.rsqrt(var + 1e-8)).abs().max() / torch.rsqrt(var + 1e-8).abs().max()),
      "sh err", float((rec["sh1"].cpu() + m * torch.rsqrt(var + 1e-8)).abs().max()))
# y intermediate of the oracle: recompute via autograd the grad wrt y2 using the taps
from dge_amd import ops
coef, gs = ops.sg1_in_bwd_coef(A.DEBUG_TAP[("dots1", i)], rec["sc1"], rec["sh1"], rec["s1"], 16 * 16)
g_pre = ops.in_bwd(A.DEBUG_TAP[2 * i], rec["y"], coef, act=False).float().cpu().permute(0, 3, 1, 2)
yt = yv.clone().requires_grad_(True)
mm, vv = O.enc_stats(yt)
u = orig(O.inorm(yt, mm, vv), rec["s1"].cpu())
print("u vs U4", float((u.detach() - U[4].detach()).abs().max()))
(u * U[4].grad).sum().backward()
print("g_y2 rel", float((g_pre - yt.grad).norm() / yt.grad.norm()))
# where does the oracle's y2 differ? (oracle y2 = input of inorm in block 2)
blk = G.decode_block[2]
g_pre_m = ops.in_bwd(A.DEBUG_TAP[2 * i], rec["y"], coef, act=True)
g_t = ops.blur_noise_act(g_pre_m, None, None, None, blur=True, act=False)
g_hi = ops.conv2d(g_t, blk._packed(blk.conv_1, ops.F32, ops.PACK_DGRAD), 64, 3)
g_lo, _ = ops.nearest_up2_bwd(g_hi, A.DEBUG_TAP[("rec", 1)]["x"])
nchw = lambda t: t.float().cpu().permute(0, 3, 1, 2)
print("recomputed vs oracle U3", float((nchw(g_lo) - U[3].grad).norm() / U[3].grad.norm()))
print("recomputed vs tap3", float((nchw(g_lo) - nchw(A.DEBUG_TAP[3])).norm() / U[3].grad.norm()))
# torch chain from oracle-side y2 grad
ref_pre = yt.grad * torch.where(yv > 0, 1.0, 0.2)
ref_t = O.sg1_blur(ref_pre)
ref_hi = torch.nn.functional.conv_transpose2d(ref_t, P["decode_block.2.conv_1.weight"], padding=1)
ref_lo = 4 * torch.nn.functional.avg_pool2d(ref_hi, 2)
print("torch chain vs oracle U3", float((ref_lo - U[3].grad).norm() / U[3].grad.norm()))
import torch.nn.functional as F
U3d = U[3].detach().clone().requires_grad_(True)
t = F.conv2d(F.interpolate(U3d, scale_factor=2, mode="nearest"), P["decode_block.2.conv_1.weight"], padding=1)
y2 = F.leaky_relu(O.sg1_blur(t) + P["decode_block.2.noise_weight_1"] * noises[4] + P["decode_block.2.bias_1"], 0.2)
print("y2 vs mine", float((y2.detach() - yv).abs().max()))
(y2 * yt.grad).sum().backward()
print("autograd subchain vs oracle U3", float((U3d.grad - U[3].grad).norm() / U[3].grad.norm()))
print("autograd subchain vs torch chain", float((U3d.grad - ref_lo).norm() / U[3].grad.norm()))
import inspect; print(inspect.getsource(O.sg1_blur))
mm_ = ((yv > 0) != (y2.detach() > 0))
print("mask mismatches", int(mm_.sum()), "of", mm_.numel(), "rows", mm_.sum((0, 1, 3)).tolist())
print("|y| small count", int((yv.abs() < 1e-4).sum()), "exact zeros mine", int((yv == 0).sum()), "oracle", int((y2 == 0).sum()))
print("neg fraction", float((yv < 0).float().mean()))
