"""ORACLE (test infrastructure only) - CPU fp32 restatement of the E_align hot path.

This file is NOT part of the product.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import it, and only as the checker.  It restates,
from the math, what the reference computes, as plain functional
torch-fp32-on-CPU code over flat parameter dicts keyed like the reference's
state_dict.  It is pinned against tests/golden/*.npz, which were produced by running
the reference itself (tools/gen_golden.py).  Unpinned part: LPIPS (third-party `lpips`
package + VGG16 weights are absent from /root/reference and from this image) - the
structure is restated in oracle/lpips_ref.py and used with seeded stand-in weights.

Reference lines followed by each function are cited in its docstring
(paths relative to /root/reference).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


# ----------------------------------------------------------------------------- StyleGAN2
def s2_dense(x, weight, bias, lr_mul=1.0, additional_bias=0.0, act="lrelu"):
    """DenseBlock.forward, model/stylegan2_generator.py:990-996 (wscale :964-967)."""
    wscale = lr_mul / math.sqrt(weight.shape[1])
    y = x.reshape(x.shape[0], -1) @ (weight * wscale).t() + bias * lr_mul + additional_bias
    if act == "lrelu":
        y = F.leaky_relu(y, 0.2) * SQRT2
    return y


def s2_mapping(P, z, prefix="mapping."):
    """MappingModule.forward :246-278 with PixelNormLayer :550-553 (eps 1e-8)."""
    w = z / torch.sqrt(torch.mean(z * z, dim=1, keepdim=True) + 1e-8)
    i = 0
    while f"{prefix}dense{i}.weight" in P:
        w = s2_dense(w, P[f"{prefix}dense{i}.weight"], P[f"{prefix}dense{i}.bias"], lr_mul=0.01)
        i += 1
    return w


def s2_truncation(w_avg, w, num_layers, psi=None, layers=None):
    """TruncationModule.forward :311-333."""
    wp = w.unsqueeze(1).repeat(1, num_layers, 1) if w.ndim == 2 else w
    psi = 1.0 if psi is None else psi
    layers = 0 if layers is None else layers
    if psi < 1.0 and layers > 0:
        coef = torch.ones(1, num_layers, 1)
        coef[:, :layers] = psi
        wp = w_avg.view(1, 1, -1) + (wp - w_avg.view(1, 1, -1)) * coef
    return wp


def fir_kernel(gain=4.0):
    k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32)
    return torch.from_numpy(k / k.sum() * gain)


def s2_upsample_skip(img):
    """UpsamplingLayer.forward :603-615 with scale_factor=2: zero-insert, pad (2,1,2,1), 4x4 FIR (sum 4)."""
    B, C, H, W = img.shape
    z = torch.zeros(B, C, 2 * H, 2 * W, dtype=img.dtype)
    z[:, :, ::2, ::2] = img
    z = F.pad(z, (2, 1, 2, 1))
    k = fir_kernel().view(1, 1, 4, 4).repeat(C, 1, 1, 1)
    return F.conv2d(z, k, groups=C)


def s2_filter_after_up(x):
    """The `filter` of an up ModulateConvBlock (:802-807): pad (1,1,1,1) + 4x4 FIR (sum 4)."""
    C = x.shape[1]
    k = fir_kernel().view(1, 1, 4, 4).repeat(C, 1, 1, 1)
    return F.conv2d(F.pad(x, (1, 1, 1, 1)), k, groups=C)


def s2_style(P, name, w):
    """style = Dense(w) + 1, linear (:825-829)."""
    return s2_dense(w, P[name + ".style.weight"], P[name + ".style.bias"], additional_bias=1.0, act="linear")


def s2_modconv(P, name, x, w, up=False, demodulate=True, add_noise=True, act="lrelu", noise=None):
    """ModulateConvBlock.forward :855-922, restated in the *shared-weight* form
    (scale activations by s, divide by the norm afterwards - the algebra of :876-877,
    :908-909), which is what the HIP kernels implement."""
    weight = P[name + ".weight"]
    cout, cin, k, _ = weight.shape
    s = s2_style(P, name, w)                                     # [B, cin]
    wh = weight * (1.0 / math.sqrt(cin * k * k))
    xm = x * s.view(-1, cin, 1, 1)
    if up:
        # conv_transpose2d stride 2 with the flipped kernel (:879-895), then FIR (:896)
        y = F.conv_transpose2d(xm, wh.flip(2, 3).permute(1, 0, 2, 3), stride=2)
        y = s2_filter_after_up(y)
    else:
        y = F.conv2d(xm, wh, padding=k // 2)
    if demodulate:
        wsq = (wh * wh).sum(dim=(2, 3))                          # [cout, cin]
        d = torch.rsqrt((s * s) @ wsq.t() + 1e-8)                # [B, cout]
        y = y * d.view(-1, cout, 1, 1)
    if add_noise:
        nz = P[name + ".noise"] if noise is None else noise
        y = y + nz * P[name + ".noise_strength"].view(1, 1, 1, 1)
    y = y + P[name + ".bias"].view(1, -1, 1, 1)
    if act == "lrelu":
        y = F.leaky_relu(y, 0.2) * SQRT2
    return y, s


def s2_num_layers(P, prefix="synthesis."):
    n = 0
    while f"{prefix}layer{n}.weight" in P:
        n += 1
    return n + 1


def s2_synthesis(P, wp, prefix="synthesis.", collect=None):
    """SynthesisModule.forward :492-539 (architecture 'skip'); latent indexing :511-517."""
    B = wp.shape[0]
    x = P[prefix + "early_layer.const"].repeat(B, 1, 1, 1)
    nl = s2_num_layers(P, prefix)
    image = None
    for i in range(nl - 1):
        x, _ = s2_modconv(P, f"{prefix}layer{i}", x, wp[:, i], up=(i % 2 == 1))
        if collect is not None:
            collect[f"layer{i}"] = x
        if i % 2 == 0:
            rgb, _ = s2_modconv(P, f"{prefix}output{i // 2}", x, wp[:, i + 1], demodulate=False,
                                add_noise=False, act="linear")
            image = rgb if image is None else rgb + s2_upsample_skip(image)
    return image


def s2_generator_eval(P, z, psi=0.7, layers=8):
    w = s2_mapping(P, z)
    nl = s2_num_layers(P)
    wp = s2_truncation(P["truncation.w_avg"], w, nl, psi, layers)
    return w, wp, s2_synthesis(P, wp)


def s2_generator_train(P, z, new_z, u, cutoff, psi=0.7, layers=8, decay=0.995, mix_prob=0.9):
    """StyleGAN2Generator.forward in train mode :174-196 (quirk Q1): returns
    (wp, new w_avg).  `new_z`, `u`, `cutoff` are the RNG draws (:185,187,188)."""
    nl = s2_num_layers(P)
    w = s2_mapping(P, z)
    w_avg = P["truncation.w_avg"] * decay + w.mean(0) * (1 - decay)
    if mix_prob > 0:
        new_w = s2_mapping(P, new_z)
        if u < mix_prob:
            w = s2_truncation(w_avg, w, nl).clone()
            new_w = s2_truncation(w_avg, new_w, nl)
            w[:, :cutoff] = new_w[:, :cutoff]
    return s2_truncation(w_avg, w, nl, psi, layers), w_avg


# ----------------------------------------------------------------------------- encoder E.BE
def enc_stats(x):
    """mean / biased std without eps, model/E/E.py:51-52."""
    m = x.mean(dim=(2, 3))
    v = ((x - m[:, :, None, None]) ** 2).mean(dim=(2, 3))
    return m, v


def inorm(x, m, v, eps=1e-8):
    return (x - m[:, :, None, None]) * torch.rsqrt(v + eps)[:, :, None, None]


def enc_block(P, pre, x, n1, n2, last):
    """BEBlock.forward model/E/E.py:50-85.  n1/n2: injected N(0,1) noise [B,1,H,W]."""
    m1, v1 = enc_stats(x)
    w1 = torch.cat([m1, v1.sqrt()], 1) @ P[pre + "inver_mod1.weight"].t() + P[pre + "inver_mod1.bias"]
    res = x
    y = F.conv2d(inorm(x, m1, v1), P[pre + "conv_1.weight"], padding=1)
    y = F.leaky_relu(y + P[pre + "noise_weight_1"] * n1 + P[pre + "bias_1"], 0.2)
    m2, v2 = enc_stats(y)
    w2 = torch.cat([m2, v2.sqrt()], 1) @ P[pre + "inver_mod2.weight"].t() + P[pre + "inver_mod2.bias"]
    y = inorm(y, m2, v2)
    if not last:
        y = F.conv2d(y, P[pre + "conv_2.weight"], padding=1)
        y = F.leaky_relu(y + P[pre + "noise_weight_2"] * n2 + P[pre + "bias_2"], 0.2)
        y = F.avg_pool2d(y, 2, 2)
        res = F.avg_pool2d(res, 2, 2)
    if pre + "conv_3.weight" in P:
        res = F.conv2d(res, P[pre + "conv_3.weight"], P[pre + "conv_3.bias"])
    return 0.111 * y + 0.889 * res, w1, w2


def enc_forward(P, img, noises, collect=None):
    """BE.forward model/E/E.py:122-136 + FromRGB model/utils/net.py:231-240.
    `noises`: list in the reference's draw order (2 per block, 1 for the last)."""
    x = F.leaky_relu(F.conv2d(img, P["FromRGB.from_rgb.weight"], P["FromRGB.from_rgb.bias"]), 0.2)
    L = 0
    while f"decode_block.{L}.conv_1.weight" in P:
        L += 1
    ws, ni = [], 0
    for j in range(L):
        last = (j == L - 1)
        n1 = noises[ni]; ni += 1
        n2 = None
        if not last:
            n2 = noises[ni]; ni += 1
        x, w1, w2 = enc_block(P, f"decode_block.{j}.", x, n1, n2, last)
        if collect is not None:
            collect[j] = (x, w1, w2)
        ws = [w2, w1] + ws                      # E.py:130-134: later blocks go first
    return x, torch.stack(ws, dim=1)


def enc_noise_shapes(L, B, res):
    shp = []
    for j in range(L):
        r = res >> j
        shp.append((B, 1, r, r))
        if j != L - 1:
            shp.append((B, 1, r, r))
    return shp


# ----------------------------------------------------------------------------- losses
def ssim_window(size=11, sigma=1.5):
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / (2.0 * sigma ** 2)) for x in range(size)])
    return g / g.sum()


def ssim(a, b):
    """metric/pytorch_ssim.py:18-38 (11x11 Gaussian, zero pad 5, C1=1e-4, C2=9e-4), mean over all."""
    C = a.shape[1]
    g = ssim_window()
    w2 = (g[:, None] @ g[None, :]).float().view(1, 1, 11, 11).repeat(C, 1, 1, 1)
    f = lambda t: F.conv2d(t, w2, padding=5, groups=C)
    mu1, mu2 = f(a), f(b)
    s11, s22, s12 = f(a * a) - mu1 * mu1, f(b * b) - mu2 * mu2, f(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s11 + s22 + 9e-4))
    return m.mean()


def space_loss(a, b, image_space=True, lpips_fn=None):
    """training_utils.py:54-99.  Returns (loss, dict of logged terms)."""
    a, b = a.contiguous(), b.contiguous()
    mse = ((a - b) ** 2).mean()
    mse_mean = (a.mean() - b.mean()) ** 2
    mse_std = (a.std() - b.std()) ** 2
    # softmax with implicit dim (:67): dim 1 for 4-D, dim 0 for 3-D tensors
    dim = 1 if a.ndim == 4 else 0
    pa, pb = F.softmax(a, dim), F.softmax(b, dim)
    kl = (pa * (torch.log(pa) - torch.log(pb))).mean()          # KLDivLoss(reduction='mean') (:68)
    kl = torch.where(torch.isnan(kl), torch.zeros_like(kl), kl)
    kl = torch.where(torch.isinf(kl), torch.ones_like(kl), kl)
    fa, fb = a.view(-1), b.view(-1)
    cos = 1 - fa.dot(fb) / (fa.dot(fa).sqrt() * fb.dot(fb).sqrt())
    ssim_l = torch.tensor(0.0)
    lp = torch.tensor(0.0)
    if image_space:
        while a.shape[2] > 256:
            a, b = F.avg_pool2d(a, 2, 2), F.avg_pool2d(b, 2, 2)
        ssim_l = 1 - ssim(a, b)
        lp = lpips_fn(a, b).mean()
    loss = 5 * mse + 3 * cos + ssim_l + 2 * lp
    return loss, dict(mse=mse, mse_mean=mse_mean, mse_std=mse_std, kl=kl, cos=cos, ssim=ssim_l, lpips=lp)


def attention_crops(img):
    """AT1 / AT2 crops of E_align_s2.py:188-199."""
    H, W = img.shape[2], img.shape[3]
    at1 = img[:, :, :, W // 8: W - W // 8]
    oy, ox = H // 8 + H // 32, W // 8 + W // 32
    at2 = img[:, :, oy: H - oy, ox: W - ox]
    return at1, at2


# ----------------------------------------------------------------------------- optimiser
def lreq_adam_step(p, g, v, step, lr, beta2=0.99, eps=1e-8, coef=None):
    """LREQAdam.step model/utils/custom_adam.py:24-76 for one tensor; returns (p, v).
    `step` is the 1-based count for this parameter."""
    v = v * beta2 + (1 - beta2) * g * g
    denom = v.sqrt() + eps
    step_size = lr * math.sqrt(1 - beta2 ** step)
    if coef is not None and coef >= 0:
        step_size *= coef
    return p - step_size * g / denom, v


# ----------------------------------------------------------------------------- StyleGAN1
def sg1_blur(x):
    """Blur, model/stylegan1/net.py:48-58: depthwise [1,2,1]x[1,2,1]/16, zero pad 1."""
    C = x.shape[1]
    k = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16.0
    return F.conv2d(x, k.view(1, 1, 3, 3).repeat(C, 1, 1, 1), padding=1, groups=C)


def sg1_style_mod(x, style):
    """style_mod :32-34: y = s1 + x*(s0+1), style [B,2C] = [s0 | s1]."""
    C = x.shape[1]
    return x * (style[:, :C, None, None] + 1) + style[:, C:, None, None]


def sg1_generator(P, styles, lod, noises):
    """Generator.decode :331-336 + DecodeBlock.forward :141-169 (+ ToRGB :244-253)."""
    x = P["const"]
    ni = 0
    for i in range(lod + 1):
        pre = f"decode_block.{i}."
        if i != 0:
            w = P[pre + "conv_1.weight"]
            if (4 << i) >= 128:      # fused scale: ConvTranspose2d(3, s2, p1) with transform_kernel (lreq.py:129-131)
                wp = F.pad(w, (1, 1, 1, 1))
                w4 = wp[:, :, 1:, 1:] + wp[:, :, :-1, 1:] + wp[:, :, 1:, :-1] + wp[:, :, :-1, :-1]
                x = F.conv_transpose2d(x, w4, stride=2, padding=1)
            else:                    # upscale2d (nearest x2) + conv
                x = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
            x = sg1_blur(x)
        x = F.leaky_relu(x + P[pre + "noise_weight_1"] * noises[ni] + P[pre + "bias_1"], 0.2); ni += 1
        m, v = enc_stats(x)
        s1 = styles[:, 2 * i] @ P[pre + "style_1.weight"].t() + P[pre + "style_1.bias"]
        x = sg1_style_mod(inorm(x, m, v), s1)
        x = F.conv2d(x, P[pre + "conv_2.weight"], padding=1)
        x = F.leaky_relu(x + P[pre + "noise_weight_2"] * noises[ni] + P[pre + "bias_2"], 0.2); ni += 1
        m, v = enc_stats(x)
        s2 = styles[:, 2 * i + 1] @ P[pre + "style_2.weight"].t() + P[pre + "style_2.bias"]
        x = sg1_style_mod(inorm(x, m, v), s2)
    return F.conv2d(x, P[f"to_rgb.{lod}.to_rgb.weight"], P[f"to_rgb.{lod}.to_rgb.bias"])


def sg1_mapping(P, z, buffer1, coefs):
    """Mapping.forward :454-466 (pixel_norm :28, 8x lrelu(Linear)), lerp towards buffer1."""
    x = z * torch.rsqrt(torch.mean(z * z, dim=1, keepdim=True) + 1e-8)
    i = 1
    while f"block_{i}.fc.weight" in P:
        x = F.leaky_relu(x @ P[f"block_{i}.fc.weight"].t() + P[f"block_{i}.fc.bias"], 0.2)
        i += 1
    L = coefs.numel()
    x = x[:, None].repeat(1, L, 1)
    return buffer1 + (x - buffer1) * coefs.view(1, L, 1)


# ----------------------------------------------------------------------------- PGGAN
def pg_pixel_norm(x):
    return x / torch.sqrt(torch.mean(x * x, dim=1, keepdim=True) + 1e-8)


def pg_generator(P, z):
    """PGGANGenerator.forward (lod 0) model/pggan/pggan_generator.py:154-204 + ConvBlock.forward :319-339."""
    def block(name, x, up=False, k=3, pad=1, gain=math.sqrt(2.0), act=True):
        w = P[name + ".weight"]
        x = pg_pixel_norm(x)
        if up:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        y = F.conv2d(x, w * (gain / math.sqrt(w.shape[1] * k * k)), P[name + ".bias"], padding=pad)
        return F.leaky_relu(y, 0.2) if act else y
    x = pg_pixel_norm(z).view(z.shape[0], -1, 1, 1)
    # layer0: pixel norm is applied by forward() on z (:160) and again inside the ConvBlock (:320) - idempotent on [B,C,1,1]
    x = block("layer0", x, k=4, pad=3)
    k = 0
    while f"layer{2 * k + 1}.weight" in P:
        if k > 0:
            x = block(f"layer{2 * k}", x, up=True)
        x = block(f"layer{2 * k + 1}", x)
        k += 1
    return block(f"output{k - 1}", x, k=1, pad=0, gain=1.0, act=False)


# ----------------------------------------------------------------------------- encoder variants
def enc_blur_forward(P, img, noises, fused):
    """E_Blur.BE.forward (model/E/E_Blur.py:50-85,122-135): E.BE + blur before conv_2; blocks with
    fused[j] use conv2d(stride 2) with the transform_kernel weights (model/utils/lreq.py:145-147)."""
    x = F.leaky_relu(F.conv2d(img, P["FromRGB.from_rgb.weight"], P["FromRGB.from_rgb.bias"]), 0.2)
    L = len(fused)
    ws, ni = [], 0
    for j in range(L):
        pre = f"decode_block.{j}."
        last = (j == L - 1)
        m1, v1 = enc_stats(x)
        w1 = torch.cat([m1, v1.sqrt()], 1) @ P[pre + "inver_mod1.weight"].t() + P[pre + "inver_mod1.bias"]
        res = x
        y = F.conv2d(inorm(x, m1, v1), P[pre + "conv_1.weight"], padding=1)
        y = F.leaky_relu(y + P[pre + "noise_weight_1"] * noises[ni] + P[pre + "bias_1"], 0.2); ni += 1
        m2, v2 = enc_stats(y)
        w2 = torch.cat([m2, v2.sqrt()], 1) @ P[pre + "inver_mod2.weight"].t() + P[pre + "inver_mod2.bias"]
        y = inorm(y, m2, v2)
        if not last:
            y = sg1_blur(y)
            w = P[pre + "conv_2.weight"]
            if fused[j]:
                wp = F.pad(w, (1, 1, 1, 1))
                w4 = (wp[:, :, 1:, 1:] + wp[:, :, :-1, 1:] + wp[:, :, 1:, :-1] + wp[:, :, :-1, :-1]) * 0.25
                y = F.conv2d(y, w4, stride=2, padding=1)
            else:
                y = F.conv2d(y, w, padding=1)
            y = F.leaky_relu(y + P[pre + "noise_weight_2"] * noises[ni] + P[pre + "bias_2"], 0.2); ni += 1
            if not fused[j]:
                y = F.avg_pool2d(y, 2, 2)
            res = F.avg_pool2d(res, 2, 2)
        if pre + "conv_3.weight" in P:
            res = F.conv2d(res, P[pre + "conv_3.weight"], P[pre + "conv_3.bias"])
        x = 0.111 * y + 0.889 * res
        ws = [w2, w1] + ws
    return x, torch.stack(ws, dim=1)


def encpg_forward(P, img, noises, L):
    """E_PG.BE trunk + head (model/E/E_PG.py:73-108,150-164) with the evident-intent return."""
    x = F.leaky_relu(F.conv2d(img, P["FromRGB.from_rgb.weight"], P["FromRGB.from_rgb.bias"]), 0.2)
    ni = 0
    for j in range(L):
        pre = f"decode_block.{j}."
        res = x
        x = F.conv2d(inorm(x, *enc_stats(x)), P[pre + "conv_1.weight"], padding=1)
        x = F.leaky_relu(x + P[pre + "noise_weight_1"] * noises[ni] + P[pre + "bias_1"], 0.2); ni += 1
        if j == L - 1:
            break
        x = F.conv2d(inorm(x, *enc_stats(x)), P[pre + "conv_2.weight"], padding=1)
        x = x + P[pre + "noise_weight_2"] * noises[ni] + P[pre + "bias_2"]; ni += 1
        if pre + "conv_3.weight" in P:
            res = F.conv2d(res, P[pre + "conv_3.weight"], P[pre + "conv_3.bias"])
            res = inorm(res, *enc_stats(res)) * P[pre + "instance_norm_3.weight"].view(1, -1, 1, 1) + P[pre + "instance_norm_3.bias"].view(1, -1, 1, 1)
        x = F.avg_pool2d(F.leaky_relu(x + res, 0.2), 2, 2)
    z = x.reshape(x.shape[0], -1) @ P["new_final.weight"].t() + P["new_final.bias"] if "new_final.weight" in P else None
    return x, z


# ----------------------------------------------------------------------------- BigGAN-deep
def bg_sn_weight(P, name):
    """torch.nn.utils.spectral_norm in eval mode: weight_orig / (u . W v)."""
    w = P[name + ".weight_orig"]
    wm = w.reshape(w.shape[0], -1)
    return w / torch.dot(P[name + ".weight_u"], torch.mv(wm, P[name + ".weight_v"]))


def bg_sn_power_iteration(P, eps=1e-12, only=None):
    """One train-mode power iteration of torch.nn.utils.spectral_norm on every (selected) `*.weight_orig` of P, updating
    `weight_u` / `weight_v` in place under no_grad (torch/nn/utils/spectral_norm.py compute_weight; the reference's
    BigGAN / E_BIG stay in train mode).  After it, the eval-form `bg_sn_weight` equals the train-mode weight."""
    with torch.no_grad():
        for k in list(P):
            if not k.endswith(".weight_orig") or (only is not None and not only(k)):
                continue
            n = k[:-len(".weight_orig")]
            w = P[k].detach().reshape(P[k].shape[0], -1)
            v = F.normalize(torch.mv(w.t(), P[n + ".weight_u"]), dim=0, eps=eps)
            u = F.normalize(torch.mv(w, v), dim=0, eps=eps)
            P[n + ".weight_u"], P[n + ".weight_v"] = u, v


def bg_bn(P, name, x, trunc, cond, eps, n_stats=51):
    """BigGANBatchNorm.forward model/biggan_generator.py:127-150."""
    coef, idx = math.modf(trunc / (1.0 / (n_stats - 1)))
    idx = int(idx)
    rm, rv = P[name + ".running_means"], P[name + ".running_vars"]
    mean = rm[idx] * coef + rm[idx + 1] * (1 - coef) if coef != 0.0 else rm[idx]
    var = rv[idx] * coef + rv[idx + 1] * (1 - coef) if coef != 0.0 else rv[idx]
    xn = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + eps)
    if name + ".scale.weight_orig" in P:
        w = 1 + cond @ bg_sn_weight(P, name + ".scale").t()
        b = cond @ bg_sn_weight(P, name + ".offset").t()
        return xn * w[:, :, None, None] + b[:, :, None, None]
    return xn * P[name + ".weight"].view(1, -1, 1, 1) + P[name + ".bias"].view(1, -1, 1, 1)


def bg_generator(P, cfg, z, onehot, trunc):
    """BigGAN.forward :296-304 + Generator.forward :232-256 + GenBlock :175-203 + SelfAttn :75-97 (eval mode)."""
    eps = cfg["eps"]
    cond = torch.cat((z, onehot @ P["embeddings.weight"].t()), dim=1)
    ch = cfg["channel_width"]
    x = cond @ bg_sn_weight(P, "generator.gen_z").t() + P["generator.gen_z.bias"]
    x = x.view(-1, 4, 4, 16 * ch).permute(0, 3, 1, 2).contiguous()
    li = 0
    for i, (up, cin, cout) in enumerate(cfg["layers"]):
        if i == cfg["attention_layer_position"]:
            pre = f"generator.layers.{li}."
            B, C, H, W = x.shape
            theta = F.conv2d(x, bg_sn_weight(P, pre + "snconv1x1_theta")).view(B, C // 8, H * W)
            phi = F.max_pool2d(F.conv2d(x, bg_sn_weight(P, pre + "snconv1x1_phi")), 2).view(B, C // 8, H * W // 4)
            attn = torch.softmax(torch.bmm(theta.permute(0, 2, 1), phi), dim=-1)
            g = F.max_pool2d(F.conv2d(x, bg_sn_weight(P, pre + "snconv1x1_g")), 2).view(B, C // 2, H * W // 4)
            ag = torch.bmm(g, attn.permute(0, 2, 1)).view(B, C // 2, H, W)
            x = x + P[pre + "gamma"] * F.conv2d(ag, bg_sn_weight(P, pre + "snconv1x1_o_conv"))
            li += 1
        pre = f"generator.layers.{li}."
        x0 = x
        t = F.conv2d(F.relu(bg_bn(P, pre + "bn_0", x, trunc, cond, eps)), bg_sn_weight(P, pre + "conv_0"), P[pre + "conv_0.bias"])
        t = F.relu(bg_bn(P, pre + "bn_1", t, trunc, cond, eps))
        if up:
            t = F.interpolate(t, scale_factor=2, mode="nearest")
        t = F.conv2d(t, bg_sn_weight(P, pre + "conv_1"), P[pre + "conv_1.bias"], padding=1)
        t = F.conv2d(F.relu(bg_bn(P, pre + "bn_2", t, trunc, cond, eps)), bg_sn_weight(P, pre + "conv_2"), P[pre + "conv_2.bias"], padding=1)
        t = F.conv2d(F.relu(bg_bn(P, pre + "bn_3", t, trunc, cond, eps)), bg_sn_weight(P, pre + "conv_3"), P[pre + "conv_3.bias"])
        if cin != cout:
            x0 = x0[:, :x0.shape[1] // 2]
        if up:
            x0 = F.interpolate(x0, scale_factor=2, mode="nearest")
        x = t + x0
        li += 1
    x = F.relu(bg_bn(P, "generator.bn", x, trunc, None, eps))
    x = F.conv2d(x, bg_sn_weight(P, "generator.conv_to_rgb"), P["generator.conv_to_rgb.bias"], padding=1)
    return torch.tanh(x[:, :3]), cond


def encbig_forward(P, img, cond, noises, L, trunc=0.4):
    """E_BIG.BE.forward (model/E/E_BIG.py:129-169,212-227), eval-mode spectral norm."""
    x = F.leaky_relu(F.conv2d(img, P["FromRGB.from_rgb.weight"], P["FromRGB.from_rgb.bias"]), 0.2)
    ni = 0
    for j in range(L):
        pre = f"decode_block.{j}."
        res = x
        x = F.conv2d(bg_bn(P, pre + "batch_norm_1", x, trunc, cond, 1e-12), P[pre + "conv_1.weight"], padding=1)
        x = F.leaky_relu(x + P[pre + "noise_weight_1"] * noises[ni] + P[pre + "bias_1"], 0.2); ni += 1
        if j == L - 1:
            break
        x = F.conv2d(bg_bn(P, pre + "batch_norm_2", x, trunc, cond, 1e-12), P[pre + "conv_2.weight"], padding=1)
        x = F.leaky_relu(x + P[pre + "noise_weight_2"] * noises[ni] + P[pre + "bias_2"], 0.2); ni += 1
        if pre + "conv_3.weight" in P:
            res = F.conv2d(bg_bn(P, pre + "batch_norm_3", res, trunc, cond, 1e-12), P[pre + "conv_3.weight"], P[pre + "conv_3.bias"])
            x = F.leaky_relu(x, 0.2)
        x = F.avg_pool2d(x + res, 2, 2)
    c_v = x.reshape(x.shape[0], -1) @ P["new_final_1.weight"].t() + P["new_final_1.bias"]
    return x, c_v, c_v @ P["new_final_2.weight"].t() + P["new_final_2.bias"]
