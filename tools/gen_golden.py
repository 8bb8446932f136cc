#!/usr/bin/env python3
"""Generate golden vectors from the reference's PyTorch CPU path.

Runs ONLY in the build container (needs /root/reference; it is a no-op elsewhere).
Imports the reference modules (never copies them), feeds them seeded inputs built by
tests/golden/recipe.py and stores the reference's outputs under tests/golden/.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [section ...]

Sections: s2 enc loss adam step
"""
import json
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

torch.set_num_threads(8)
from tests.golden import recipe as R


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


# training_utils imports torchvision + PIL at module scope
_tv = _stub("torchvision")
_tv.transforms = _stub("torchvision.transforms", Compose=lambda x: None, ToTensor=lambda: None)
try:
    import PIL  # noqa
except Exception:
    _stub("PIL").Image = None


def shapes_of(sd):
    return {k: list(v.shape) for k, v in sd.items()}


def save_npz(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print("wrote", name, os.path.getsize(path) // 1024, "KiB")


# --------------------------------------------------------------------------- S2
def gen_s2():
    from model.stylegan2_generator import (StyleGAN2Generator, ModulateConvBlock,
                                           UpsamplingLayer)
    # key/shape lists of the real configurations (state_dict surface)
    keys = {}
    for res in (1024, 256):
        g = StyleGAN2Generator(res)
        keys[str(res)] = shapes_of(g.state_dict())
        del g
    with open(os.path.join(OUT, "s2_keys.json"), "w") as f:
        json.dump(keys, f)

    # -- single ModulateConvBlock cases (a1) and UpsamplingLayer (a2)
    cases = [  # in, out, res(out), up, k, demod, noise, act
        (128, 128, 8, False, 3),
        (128, 64, 16, True, 3),
        (64, 32, 32, True, 3),
        (32, 32, 32, False, 3),
        (32, 3, 32, False, 1),
        (512, 512, 4, False, 3),
    ]
    out = {}
    for ci, (cin, cout, res, up, k) in enumerate(cases):
        torgb = (k == 1)
        blk = ModulateConvBlock(cin, cout, res, 512, kernel_size=k, scale_factor=2 if up else 1,
                                demodulate=not torgb, add_noise=not torgb,
                                activation_type="linear" if torgb else "lrelu")
        sd = R.fill_s2(shapes_of(blk.state_dict()), seed=100 + ci)
        blk.load_state_dict(sd)
        rin = res // 2 if up else res
        x = R.randn(f"mc{ci}.x", (2, cin, rin, rin), 7)
        w = R.randn(f"mc{ci}.w", (2, 512), 7)
        with torch.no_grad():
            y, style = blk(x, w)
            # metamorphic: non-fused form (stylegan2_generator.py:876-877,908-909)
            blk.fused_modulate = False
            y2, _ = blk(x, w)
        out[f"c{ci}_y"] = y
        out[f"c{ci}_style"] = style
        out[f"c{ci}_y_nonfused"] = y2
        out[f"c{ci}_cfg"] = np.array([cin, cout, res, int(up), k])
    for mode, kw in (("up", dict()), ("filt", dict(scale_factor=1, extra_padding=-1, kernel_gain=2))):
        up = UpsamplingLayer(**kw)
        x = R.randn(f"ups.{mode}", (2, 3, 9 if mode == "filt" else 8, 9 if mode == "filt" else 8), 3)
        out[f"ups_{mode}_y"] = up(x)
    save_npz("s2_blocks.npz", **out)

    # -- whole generator, reduced width (128/128/128/64/32 channels @ 4..64)
    torch.manual_seed(0)
    G = StyleGAN2Generator(64, fmaps_base=2048, fmaps_max=128)
    sd = R.fill_s2(shapes_of(G.state_dict()), seed=11)
    G.load_state_dict(sd)
    out = {}
    # (i) synthesis(wp) with hooks on intermediates
    wp = R.randn("s2.wp", (2, G.num_layers, 512), 5)
    feats = {}
    hooks = []
    for name in ("layer0", "layer1", "layer2", "layer7", "layer8", "output0", "output4"):
        hooks.append(getattr(G.synthesis, name).register_forward_hook(
            lambda m, i, o, name=name: feats.__setitem__(name, o[0].detach().clone())))
    G.eval()
    with torch.no_grad():
        r = G.synthesis(wp)
    for h in hooks:
        h.remove()
    out["syn_image"] = r["image"]
    out["syn_style00"] = r["style00"]
    out["syn_output_style4"] = r["output_style4"]
    for k, v in feats.items():
        out["syn_" + k] = v
    # (ii) eval-mode full forward: mapping + truncation + synthesis
    z = R.randn("s2.z", (2, 512), 5)
    with torch.no_grad():
        r = G(z, trunc_psi=0.7, trunc_layers=8, randomize_noise=False)
    out["eval_w"] = r["w"]
    out["eval_wp"] = r["wp"]
    out["eval_image"] = r["image"]
    # (iii) train-mode forward as E_align_s2.py runs it (Q1): w_avg EMA + style mixing.
    G.train()
    new_z = R.randn("s2.new_z", (2, 512), 5)
    orig_randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: new_z.clone()
    try:
        for it, tag in ((3, "a"), (4, "b")):   # two iterations: seeds chosen to hit mix / no-mix
            np.random.seed(it)
            u = np.random.uniform()
            cutoff = np.random.randint(1, G.num_layers) if u < 0.9 else -1
            np.random.seed(it)
            w_avg_before = G.truncation.w_avg.clone()
            with torch.no_grad():
                r = G(z, trunc_psi=0.7, trunc_layers=8, randomize_noise=False)
            out[f"train_{tag}_u_cutoff"] = np.array([u, cutoff], dtype=np.float64)
            out[f"train_{tag}_w_avg_before"] = w_avg_before
            out[f"train_{tag}_w_avg_after"] = G.truncation.w_avg.clone()
            out[f"train_{tag}_wp"] = r["wp"]
            out[f"train_{tag}_image"] = r["image"]
    finally:
        torch.randn_like = orig_randn_like
    # force a no-mix case explicitly (style_mixing_prob = 0)
    G.truncation.w_avg.copy_(sd["truncation.w_avg"])
    with torch.no_grad():
        r = G(z, style_mixing_prob=0.0, trunc_psi=0.7, trunc_layers=8)
    out["train_nomix_wp"] = r["wp"]
    out["train_nomix_w_avg_after"] = G.truncation.w_avg.clone()
    out["state_checksum"] = np.array(R.checksum(sd))
    # (iv) gradient of an image functional w.r.t. wp (what the encoder receives, phase E)
    G.truncation.w_avg.copy_(sd["truncation.w_avg"])
    wp_g = wp.clone().requires_grad_(True)
    img = G.synthesis(wp_g)["image"]
    gimg = R.randn("s2.gimg", tuple(img.shape), 5, 1.0 / img.numel() ** 0.5)
    (img * gimg).sum().backward()
    out["grad_wp"] = wp_g.grad
    save_npz("s2_small.npz", **out)


# --------------------------------------------------------------------------- encoder
class _NoiseFeeder:
    """Replaces torch.randn inside the reference encoder so that noise tensors are the
    recipe's (reference draws them with the CPU generator, model/E/E.py:60,73 - Q6)."""

    def __init__(self, prefix, seed):
        self.prefix, self.seed, self.i, self.log = prefix, seed, 0, []
        self.orig = torch.randn

    def __call__(self, *size, **kw):
        if len(size) == 1 and isinstance(size[0], (list, tuple)):
            size = tuple(size[0])
        t = R.randn(f"{self.prefix}.noise{self.i}", size, self.seed)
        self.i += 1
        self.log.append(tuple(size))
        return t

    def __enter__(self):
        torch.randn = self
        return self

    def __exit__(self, *a):
        torch.randn = self.orig


class _LreluTap:
    """Records the input of every F.leaky_relu call of a reference forward, in call order (the reference modules call
    `F.leaky_relu` through the torch.nn.functional module object, so replacing the attribute reaches all of them)."""

    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.orig, self.inputs = F, F.leaky_relu, []
        F.leaky_relu = self
        return self

    def __call__(self, x, *a, **kw):
        self.inputs.append(x.detach().clone())
        return self.orig(x, *a, **kw)

    def __exit__(self, *a):
        self.F.leaky_relu = self.orig


KINK_MARGIN = 1e-4      # gradient fixtures: no leaky-relu pre-activation within KINK_MARGIN * max|pre-activation| of zero


def clear_kinks(runs, owners, names, margin=KINK_MARGIN):
    """Gradient parity at 1e-3 is meaningless when a pre-activation of the fixture sits within rounding distance of the leaky-relu
    kink: the slope (0.2 <-> 1) of that element is then decided by the last bit of whoever computes it, and a 64-element
    bias / noise-weight gradient moves by ~1e-2 per flipped element.  The reference adds a per-channel bias right in front of
    every leaky-relu (model/E/E.py:61-62,74-75; E_PG.py:83-84,97-102; E_BIG.py:142-143,157-158; stylegan1/net.py:150-152,162-164;
    model/utils/net.py:238-239), so the fixture's parameters are chosen such that this cannot happen: layer by layer, in forward order, each channel's
    bias is moved by the smallest amount (a few 1e-4 of the activation scale) that leaves every pre-activation of the channel
    at least 2 * margin * max away from zero.

    runs   : callables, each performs one reference forward (all of them see every nudge; call k of every run shares owners[k])
    owners : per leaky-relu call, the bias Parameter in front of it (None: the call re-applies lrelu to an lrelu output - same kink)
    names  : state_dict key of each owner (None for None)
    Returns ({key: nudged bias tensor}, achieved margin = min over calls of min|x| / max|x|)."""
    changed = {}

    def tap_all():
        recs = []
        for run in runs:
            with _LreluTap() as tap, torch.no_grad():
                run()
            assert len(tap.inputs) <= len(owners), (len(tap.inputs), len(owners))
            recs.append(tap.inputs)
        return recs

    for k, own in enumerate(owners):
        if own is None:
            continue
        xs = [r[k] for r in tap_all() if len(r) > k]
        C = own.numel()
        assert all(x.shape[1] == C for x in xs), (k, names[k], [tuple(x.shape) for x in xs], C)
        mx = max(float(x.abs().max()) for x in xs)
        mt = 2.0 * margin * mx
        v = torch.cat([x.transpose(0, 1).reshape(C, -1) for x in xs], dim=1).double().numpy()      # [C, values]
        delta = np.zeros(C)
        for c in range(C):
            u = np.sort(v[c])
            if np.min(np.abs(u)) > mt:
                continue
            # p = -delta must be >= mt away from every u: pick the admissible point closest to zero
            cand = [u[0] - mt, u[-1] + mt]
            gaps = np.nonzero(u[1:] - u[:-1] >= 2 * mt)[0]
            for i in gaps:
                lo, hi = u[i] + mt, u[i + 1] - mt
                cand.append(min(max(0.0, lo), hi))
            p = min(cand, key=abs)
            delta[c] = -p
        if np.any(delta != 0):
            with torch.no_grad():
                own.view(-1).add_(torch.from_numpy(delta).to(own.dtype))
            changed[names[k]] = own.detach().clone()
    # the assertion of the generator: every call of every run clears the margin with the final parameters
    worst = 1.0
    for r in tap_all():
        for k, x in enumerate(r):
            if owners[k] is None:
                continue
            rel = float(x.abs().min()) / float(x.abs().max())
            assert rel > margin, (k, names[k], rel)
            worst = min(worst, rel)
    print(f"  clear_kinks: {len(changed)} bias tensors nudged, min |pre-activation| / max = {worst:.2e} (margin {margin:g})")
    return changed, worst


def enc_kink_owners(E, second_attr, third=False):
    """leaky-relu call order of the reference encoders: FromRGB, then per block bias_1 [, bias_2 [, a second lrelu on x (E_BIG.py:163)]]"""
    owners, names = [E.FromRGB.from_rgb.bias], ["FromRGB.from_rgb.bias"]
    for j, b in enumerate(E.decode_block):
        owners.append(b.bias_1); names.append(f"decode_block.{j}.bias_1")
        if getattr(b, second_attr):
            owners.append(b.bias_2); names.append(f"decode_block.{j}.bias_2")
            if third and b.inputs != b.outputs:
                owners.append(None); names.append(None)
    return owners, names


def gen_enc():
    import model.E.E as EE
    keys = {}
    for tag, (sf, lc) in (("1024_16_9", (16, 9)), ("256_64_7", (64, 7))):
        e = EE.BE(startf=sf, maxf=512, layer_count=lc)
        keys[tag] = shapes_of(e.state_dict())
        keys[tag + "_lreq"] = {k: float(getattr(p, "lr_equalization_coef", -1.0))
                               for k, p in e.named_parameters()}
        del e
    with open(os.path.join(OUT, "enc_keys.json"), "w") as f:
        json.dump(keys, f)

    # small encoder: startf=16, maxf=64, 4 blocks, 32x32 input: ch 16->32->64->64(->64)
    E = EE.BE(startf=16, maxf=64, layer_count=4)
    sd = R.fill_encoder(shapes_of(E.state_dict()), seed=21)
    E.load_state_dict(sd)
    img = R.randn("enc.img", (2, 3, 32, 32), 9, 0.5)
    img.requires_grad_(True)

    def run():
        with _NoiseFeeder("enc", 9):
            E(img)
    nudged, margin = clear_kinks([run], *enc_kink_owners(E, "has_last_conv"))
    feats = {}
    hooks = [E.decode_block[j].register_forward_hook(
        lambda m, i, o, j=j: feats.__setitem__(j, [t.detach().clone() for t in o])) for j in range(4)]
    with _NoiseFeeder("enc", 9) as nf:
        x, w = E(img)
    for h in hooks:
        h.remove()
    out = {"x": x, "w": w, "noise_shapes": np.array([list(s) for s in nf.log])}
    for j, (xo, w1, w2) in feats.items():
        out[f"blk{j}_x"], out[f"blk{j}_w1"], out[f"blk{j}_w2"] = xo, w1, w2
    gw = R.randn("enc.gw", tuple(w.shape), 9, 0.05)
    (w * gw).sum().backward()
    out["grad_img"] = img.grad
    for k, p in E.named_parameters():
        if p.grad is not None:
            out["grad:" + k] = p.grad
        else:
            out["nograd:" + k] = np.zeros(1)
    out["state_checksum"] = np.array(R.checksum({k: v.detach() for k, v in E.state_dict().items()}))      # with the nudged biases
    out["kink_margin"] = np.array(margin)
    out.update({"param:" + k: v for k, v in nudged.items()})
    save_npz("enc_small.npz", **out)


# --------------------------------------------------------------------------- losses
def gen_loss():
    import training_utils as TU
    import metric.pytorch_ssim as PS
    out = {}
    a = R.randn("loss.a", (2, 3, 64, 64), 1, 0.5).clamp(-1, 1)
    b = (a + R.randn("loss.b", (2, 3, 64, 64), 1, 0.1)).clamp(-1, 1)
    out["ssim_64"] = PS.ssim(a, b)
    out["ssim_same"] = PS.ssim(a, a)
    a2 = R.randn("loss.a2", (1, 3, 40, 24), 1, 0.5)
    b2 = a2 * 0.7 + 0.1
    out["ssim_40x24"] = PS.ssim(a2, b2)

    # stand-in LPIPS: the real package/weights are absent (parity unpinned there)
    standin = lambda x, y: ((x - y) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    big_a = R.randn("loss.big_a", (2, 3, 512, 384), 2, 0.5)
    big_b = big_a + R.randn("loss.big_b", (2, 3, 512, 384), 2, 0.2)
    big_b.requires_grad_(True)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        l, info = TU.space_loss(big_a, big_b, lpips_model=standin)
        l.backward()
        out["img_loss"] = l.detach()
        out["img_info"] = np.array([info[0][0], info[0][1], info[0][2], info[1], info[2], info[3], info[4]])
        out["img_grad_b_sum"] = np.array([big_b.grad.double().sum().item(), big_b.grad.double().abs().sum().item()])
        out["img_grad_b_crop"] = big_b.grad[:, :, 100:116, 200:216].clone()
        w1 = R.randn("loss.w1", (2, 18, 512), 2)
        w2 = (w1 * 0.9 + R.randn("loss.w2", (2, 18, 512), 2, 0.3)).requires_grad_(True)
        l, info = TU.space_loss(w1, w2, image_space=False)
        l.backward()
        out["w_loss"] = l.detach()
        out["w_info"] = np.array([info[0][0], info[0][1], info[0][2], info[1], info[2], info[3], info[4]])
        out["w_grad"] = w2.grad
    save_npz("loss.npz", **out)


# --------------------------------------------------------------------------- adam
def gen_adam():
    from model.utils.custom_adam import LREQAdam
    import model.utils.lreq as ln
    lin = ln.Linear(12, 7)
    conv = ln.Conv2d(4, 6, 3, 1, 1, bias=False)
    plain = torch.nn.Parameter(torch.zeros(1, 6, 1, 1))
    params = {"lin.weight": lin.weight, "lin.bias": lin.bias, "conv.weight": conv.weight, "plain": plain}
    with torch.no_grad():
        for k, p in params.items():
            p.copy_(R.randn("adam.p." + k, tuple(p.shape), 0, 0.3))
    opt = LREQAdam([{"params": list(params.values())}], lr=0.0015, betas=(0.0, 0.99), weight_decay=0)
    out = {"coef": np.array([getattr(p, "lr_equalization_coef", -1.0) for p in params.values()])}
    import warnings
    for step in range(3):
        for k, p in params.items():
            p.grad = R.randn(f"adam.g{step}." + k, tuple(p.shape), 0, 0.01 * (step + 1))
        if step == 1:
            plain.grad = None    # skipped param must not advance its step counter
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            opt.step()
        for k, p in params.items():
            out[f"s{step}:{k}"] = p.detach().clone()
    save_npz("adam.npz", **out)


# --------------------------------------------------------------------------- two-phase step
def gen_step():
    """One full E_align_s2 iteration (mtype 2) of the REFERENCE at reduced size: G train-mode
    forward (Q1), E, G.synthesis, 3x space_loss, backward(retain_graph) + LREQAdam.step,
    latent loss, backward + step (Q3: second backward sees the updated weights)."""
    import warnings
    from model.stylegan2_generator import StyleGAN2Generator
    import model.E.E as EE
    import training_utils as TU
    from model.utils.custom_adam import LREQAdam
    from oracle import lpips_ref as LR

    G = StyleGAN2Generator(64, fmaps_base=2048, fmaps_max=128)
    G.load_state_dict(R.fill_s2(shapes_of(G.state_dict()), seed=11))
    E = EE.BE(startf=16, maxf=64, layer_count=5)
    E.load_state_dict(R.fill_encoder(shapes_of(E.state_dict()), seed=31))
    LP = LR.seeded_params(0)
    lp = lambda a, b: LR.lpips(LP, a, b)
    opt = LREQAdam([{"params": E.parameters()}], lr=0.0015, betas=(0.0, 0.99), weight_decay=0)
    out = {}
    B = 2
    new_z = R.randn("step.new_z", (B, 512), 1)
    orig_randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: new_z.clone()
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
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                l_i, i_i = TU.space_loss(imgs1, imgs2, lpips_model=lp)
                m1 = imgs1[:, :, :, imgs1.shape[3] // 8:-imgs1.shape[3] // 8]
                m2 = imgs2[:, :, :, imgs2.shape[3] // 8:-imgs2.shape[3] // 8]
                l_m, i_m = TU.space_loss(m1, m2, lpips_model=lp)
                o = imgs1.shape[2] // 8 + imgs1.shape[2] // 32
                s1, s2 = imgs1[:, :, o:-o, o:-o], imgs2[:, :, o:-o, o:-o]
                l_s, i_s = TU.space_loss(s1, s2, lpips_model=lp)
                loss_tsa = l_i + l_m * 5 + l_s * 9
                opt.zero_grad()
                loss_tsa.backward(retain_graph=True)
                gsum1 = sum(float(p.grad.double().abs().sum()) for p in E.parameters() if p.grad is not None)
                opt.step()
                out[f"it{it}_after_phase1:decode_block.0.conv_1.weight"] = E.decode_block[0].conv_1.weight.detach().clone()
                out[f"it{it}_after_phase1:decode_block.4.inver_mod2.weight"] = E.decode_block[4].inver_mod2.weight.detach().clone()
                l_w, i_w = TU.space_loss(w1, w2, image_space=False)
                loss_mtv = l_w * 0.01
                opt.zero_grad()
                loss_mtv.backward()
                gsum2 = sum(float(p.grad.double().abs().sum()) for p in E.parameters() if p.grad is not None)
                out[f"it{it}_grad2:decode_block.1.conv_2.weight"] = E.decode_block[1].conv_2.weight.grad.clone()
                out[f"it{it}_grad2:decode_block.0.conv_1.weight"] = E.decode_block[0].conv_1.weight.grad.clone()
                opt.step()
            flat = lambda inf: [inf[0][0], inf[0][1], inf[0][2], inf[1], inf[2], inf[3], inf[4]]
            out[f"it{it}_imgs1"] = imgs1
            out[f"it{it}_w1"] = w1
            out[f"it{it}_w2"] = w2.detach()
            out[f"it{it}_imgs2"] = imgs2.detach()
            out[f"it{it}_losses"] = np.array([float(loss_tsa), float(l_i), float(l_m), float(l_s), float(l_w)])
            out[f"it{it}_info"] = np.array([flat(i_i), flat(i_m), flat(i_s), flat(i_w)])
            out[f"it{it}_gradsums"] = np.array([gsum1, gsum2])
            out[f"it{it}_w_avg"] = G.truncation.w_avg.clone()
            out[f"it{it}_param_checksum"] = np.array(R.checksum(E.state_dict()))
            for k in ("decode_block.0.conv_1.weight", "decode_block.2.conv_2.weight", "decode_block.4.inver_mod2.weight",
                      "decode_block.1.bias_1", "decode_block.0.noise_weight_2", "FromRGB.from_rgb.weight"):
                out[f"it{it}_after_phase2:{k}"] = E.state_dict()[k].clone()
    finally:
        torch.randn_like = orig_randn_like
    save_npz("step_s2.npz", **out)


def gen_step_s1(legacy=False):
    """legacy=True: the same loop with `zero_grad(set_to_none=False)` - what torch < 2.0 (the reference pins torch >= 1.8 on
    python 3.7, i.e. <= 1.13) does by default: from the second iteration on the gradients are zero TENSORS at the script's first
    optimizer step, so LREQAdam advances every step counter and decays every second moment there (custom_adam.py:35-62) - three
    iterations, tests/golden/step_s1_legacy.npz.
    Two iterations of the STAGE-1 variant (E_align_cropping_s1.py:185-218) of the REFERENCE's modules at reduced size: the
    three image-space losses are computed on `.detach().clone()` inputs and summed unweighted (:185-203) - they carry no
    gradient to the encoder, so the first backward / optimizer step of the script changes nothing in E (gradients stay None
    after zero_grad) - and only the latent phase loss_w * 0.01 (:207-218) trains it."""
    import warnings
    from model.stylegan2_generator import StyleGAN2Generator
    import model.E.E as EE
    import training_utils as TU
    from model.utils.custom_adam import LREQAdam
    from oracle import lpips_ref as LR

    G = StyleGAN2Generator(64, fmaps_base=2048, fmaps_max=128)
    G.load_state_dict(R.fill_s2(shapes_of(G.state_dict()), seed=11))
    E = EE.BE(startf=16, maxf=64, layer_count=5)
    E.load_state_dict(R.fill_encoder(shapes_of(E.state_dict()), seed=31))
    LP = LR.seeded_params(0)
    lp = lambda a, b: LR.lpips(LP, a, b)
    opt = LREQAdam([{"params": E.parameters()}], lr=0.0015, betas=(0.0, 0.99), weight_decay=0)
    out = {}
    B = 2
    new_z = R.randn("step.new_z", (B, 512), 1)
    orig_randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: new_z.clone()
    try:
        for it in range(3 if legacy else 2):
            np.random.seed(it)
            z = R.randn(f"step.z{it}", (B, 512), 1)
            with torch.no_grad():
                r = G(z, trunc_psi=0.7, trunc_layers=8, randomize_noise=False)
            imgs1, w1 = r["image"], r["wp"]
            with _NoiseFeeder(f"step.it{it}", 1):
                const2, w2 = E(imgs1)
            imgs2 = G.synthesis(w2)["image"]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                l_i, i_i = TU.space_loss(imgs1.detach().clone(), imgs2.detach().clone(), lpips_model=lp)
                m1 = imgs1[:, :, :, imgs1.shape[3] // 8:-imgs1.shape[3] // 8].detach().clone()
                m2 = imgs2[:, :, :, imgs2.shape[3] // 8:-imgs2.shape[3] // 8].detach().clone()
                l_m, i_m = TU.space_loss(m1, m2, lpips_model=lp)
                o = imgs1.shape[2] // 8 + imgs1.shape[2] // 32
                s1, s2 = imgs1[:, :, o:-o, o:-o].detach().clone(), imgs2[:, :, o:-o, o:-o].detach().clone()
                l_s, i_s = TU.space_loss(s1, s2, lpips_model=lp)
                loss_tsa = l_i + l_m + l_s
                opt.zero_grad(set_to_none=not legacy)
                assert not loss_tsa.requires_grad          # nothing of E is reachable: the script's backward() only touches lpips' own layers
                opt.step()                                  # gradients None: nothing changes; legacy: zero tensors -> t += 1, v *= beta2
                l_w, i_w = TU.space_loss(w1, w2, image_space=False)
                loss_mtv = l_w * 0.01
                opt.zero_grad(set_to_none=not legacy)
                loss_mtv.backward()
                opt.step()
            flat = lambda inf: [inf[0][0], inf[0][1], inf[0][2], inf[1], inf[2], inf[3], inf[4]]
            out[f"it{it}_w2"] = w2.detach()
            out[f"it{it}_imgs2"] = imgs2.detach()
            out[f"it{it}_losses"] = np.array([float(loss_tsa), float(l_i), float(l_m), float(l_s), float(l_w)])
            out[f"it{it}_info"] = np.array([flat(i_i), flat(i_m), flat(i_s), flat(i_w)])
            out[f"it{it}_param_checksum"] = np.array(R.checksum(E.state_dict()))
            for k in ("decode_block.0.conv_1.weight", "decode_block.2.conv_2.weight", "decode_block.4.inver_mod2.weight",
                      "decode_block.1.bias_1", "FromRGB.from_rgb.weight"):
                out[f"it{it}_after_phase2:{k}"] = E.state_dict()[k].clone()
    finally:
        torch.randn_like = orig_randn_like
    save_npz("step_s1_legacy.npz" if legacy else "step_s1.npz", **out)


# --------------------------------------------------------------------------- StyleGAN1
def gen_sg1():
    import model.stylegan1.net as SG1
    keys = {}
    for tag, (sf, lc) in (("256_64_7", (64, 7)), ("1024_16_9", (16, 9))):
        g = SG1.Generator(startf=sf, maxf=512, layer_count=lc, latent_size=512, channels=3)
        keys["Gs_" + tag] = shapes_of(g.state_dict())
        del g
    m = SG1.Mapping(num_layers=14, mapping_layers=8, latent_size=512, dlatent_size=512, mapping_fmaps=512)
    keys["Gm"] = shapes_of(m.state_dict())
    with open(os.path.join(OUT, "sg1_keys.json"), "w") as f:
        json.dump(keys, f)
    # reduced generator: 6 blocks (4..128), channels 64,64,64,64,64,32; the last block uses the fused
    # ConvTranspose2d(3, stride 2) + transform_kernel path (resolution >= 128), the others upscale2d + conv
    G = SG1.Generator(startf=32, maxf=64, layer_count=6, latent_size=512, channels=3)
    sd = R.fill_encoder(shapes_of(G.state_dict()), seed=41)
    for k in sd:
        if k.endswith("blur.weight"):
            sd[k] = G.state_dict()[k].clone()
        if k == "const":
            sd[k] = R.randn("sg1.const", tuple(sd[k].shape), 41)
    G.load_state_dict(sd)
    styles = R.randn("sg1.styles", (2, 12, 512), 6)
    out = {}
    feats = {}
    hooks = [G.decode_block[j].register_forward_hook(lambda m_, i, o, j=j: feats.__setitem__(j, o.detach().clone())) for j in range(6)]
    with torch.no_grad(), _NoiseFeeder("sg1", 6) as nf:
        img = G.forward(styles, 5)
    for h in hooks:
        h.remove()
    out["image"] = img
    out["noise_shapes"] = np.array([list(s_) for s_ in nf.log])
    for j in (0, 1):
        out[f"blk{j}"] = feats[j]
    with torch.no_grad(), _NoiseFeeder("sg1b", 6):
        out["image_lod3"] = G.forward(styles, 3)
    out["state_checksum"] = np.array(R.checksum(sd))
    # mapping with truncation towards buffer1 (E_align_s2.py:32-41)
    M = SG1.Mapping(num_layers=12, mapping_layers=8, latent_size=512, dlatent_size=512, mapping_fmaps=512)
    msd = {k: R.randn("sg1m." + k, tuple(v.shape), 42, 0.05 if k.endswith("weight") else 0.01) for k, v in M.state_dict().items()}
    M.load_state_dict(msd)
    M.buffer1 = R.randn("sg1m.buffer1", (12, 512), 42, 0.5)
    layer_idx = torch.arange(12)[np.newaxis, :, np.newaxis]
    coefs = torch.where(layer_idx < 6, 0.7 * torch.ones(1, 12, 1), torch.ones(1, 12, 1))
    z = R.randn("sg1m.z", (3, 512), 42)
    with torch.no_grad():
        out["mapping_w"] = M(z, coefs_m=coefs)
    save_npz("sg1_small.npz", **out)


# --------------------------------------------------------------------------- PGGAN
def gen_pggan():
    import contextlib, io
    from model.pggan.pggan_generator import PGGANGenerator
    keys = {"256": shapes_of(PGGANGenerator(256).state_dict())}
    with open(os.path.join(OUT, "pggan_keys.json"), "w") as f:
        json.dump(keys, f)
    G = PGGANGenerator(32, fmaps_base=1024, fmaps_max=64)
    sd = {k: (R.randn("pg." + k, tuple(v.shape), 51, 0.2 if k.endswith("bias") else 1.0) if v.ndim else v.clone())
          for k, v in G.state_dict().items()}
    G.load_state_dict(sd)
    z = R.randn("pg.z", (2, 512), 51)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):    # the reference prints x.shape (:196)
        r = G(z)
    save_npz("pggan_small.npz", image=r["image"], z=r["z"], state_checksum=np.array(R.checksum(sd)))


# --------------------------------------------------------------------------- encoder variants
def gen_encvar():
    import model.E.E_Blur as EB
    import model.E.E_PG as EP
    keys = {"E_Blur_1024_16_9": shapes_of(EB.BE(startf=16, maxf=512, layer_count=9).state_dict()),
            "E_PG_256_64_7": shapes_of(EP.BE(startf=64, maxf=512, layer_count=7, pggan=True).state_dict())}
    with open(os.path.join(OUT, "encvar_keys.json"), "w") as f:
        json.dump(keys, f)
    # E_Blur: 6 blocks on a 128x128 input; blocks 0-3 take the stride-2 transform_kernel path (ctor resolution >= 128)
    E = EB.BE(startf=16, maxf=64, layer_count=6)
    sd = R.fill_encoder(shapes_of(E.state_dict()), seed=61)
    for k in sd:
        if k.endswith("blur.weight"):
            sd[k] = E.state_dict()[k].clone()
    E.load_state_dict(sd)
    img = R.randn("eb.img", (2, 3, 128, 128), 61, 0.5)
    with torch.no_grad(), _NoiseFeeder("eb", 61) as nf:
        x, w = E(img)
    save_npz("encblur_small.npz", x=x, w=w, noise_shapes=np.array([list(s_) for s_ in nf.log]), state_checksum=np.array(R.checksum(sd)),
             fused=np.array([int(b.fused_scale) for b in E.decode_block]))
    # E_PG: 5 blocks on 64x64 (startf 32 -> 512 channels at 4x4, as new_final expects 512*16 inputs)
    E = EP.BE(startf=32, maxf=512, layer_count=5, pggan=True)
    sd = R.fill_encoder(shapes_of(E.state_dict()), seed=62)
    for k in sd:
        if "instance_norm_3.weight" in k:
            sd[k] = R.randn("pg." + k, tuple(sd[k].shape), 62, 0.2, 1.0)
    E.load_state_dict(sd)
    img = R.randn("ep.img", (2, 3, 64, 64), 62, 0.5)
    feats = {}
    h1 = E.decode_block[4].register_forward_hook(lambda m_, i, o: feats.__setitem__("trunk", o[0].detach().clone()))
    h2 = E.new_final.register_forward_hook(lambda m_, i, o: feats.__setitem__("head", o.detach().clone()))
    with torch.no_grad(), _NoiseFeeder("ep", 62) as nf:
        r = E(img)
    h1.remove(); h2.remove()
    save_npz("encpg_small.npz", trunk=feats["trunk"], head=feats["head"], ret0=r[0], ret1=r[1],
             noise_shapes=np.array([list(s_) for s_ in nf.log]), state_checksum=np.array(R.checksum(sd)))


# --------------------------------------------------------------------------- BigGAN-deep
BIGGAN_SMALL_CFG = dict(output_dim=64, z_dim=128, class_embed_dim=128, channel_width=32, num_classes=1000,
                        layers=[[True, 16, 8], [False, 8, 8], [True, 8, 4], [True, 4, 2], [True, 2, 1]],
                        attention_layer_position=3, eps=1e-4, n_stats=51)
BIGGAN_DEEP256_CFG = dict(output_dim=256, z_dim=128, class_embed_dim=128, channel_width=128, num_classes=1000,
                          layers=[[False, 16, 16], [True, 16, 16], [False, 16, 16], [True, 16, 8], [False, 8, 8], [True, 8, 8],
                                  [False, 8, 8], [True, 8, 4], [False, 4, 4], [True, 4, 2], [False, 2, 2], [True, 2, 1]],
                          attention_layer_position=8, eps=1e-4, n_stats=51)


def gen_biggan():
    _stub("boto3"); _stub("botocore"); _stub("botocore.exceptions", ClientError=Exception)
    _stub("requests"); 
    from model.biggan_generator import BigGAN
    from model.utils.biggan_config import BigGANConfig
    keys = {"deep256": shapes_of(BigGAN(BigGANConfig.from_dict(BIGGAN_DEEP256_CFG)).state_dict())}
    with open(os.path.join(OUT, "biggan_keys.json"), "w") as f:
        json.dump(keys, f)
    G = BigGAN(BigGANConfig.from_dict(BIGGAN_SMALL_CFG))
    sd = R.fill_biggan(shapes_of(G.state_dict()), seed=71)
    G.load_state_dict(sd)
    G.eval()
    z = R.randn("bg.z", (2, 128), 71, 0.4)
    onehot = torch.zeros(2, 1000); onehot[:, 207] = 1.0
    with torch.no_grad():
        img, cond = G(z, onehot, 0.4)
    out = {"image": img, "cond": cond, "state_checksum": np.array(R.checksum(sd))}
    with torch.no_grad():
        out["image_t05"], _ = G(z, onehot, 0.5)
        out["image_t037"], _ = G(z, onehot, 0.37)      # interpolated statistics rows (coef != 0)
    # train mode: one power iteration per forward mutates weight_u / weight_v (quirk Q2)
    G.train()
    with torch.no_grad():
        out["image_train"], _ = G(z, onehot, 0.4)
    out["train_u_gen_z"] = G.state_dict()["generator.gen_z.weight_u"].clone()
    save_npz("biggan_small.npz", **out)


SECTIONS = {"s2": gen_s2, "enc": gen_enc, "loss": gen_loss, "adam": gen_adam, "step": gen_step, "sg1": gen_sg1, "pggan": gen_pggan,
            "encvar": gen_encvar, "biggan": gen_biggan}


def gen_encbig():
    import model.E.E_BIG as EBG
    keys = {"E_BIG_256_64_7": shapes_of(EBG.BE(startf=64, maxf=512, layer_count=7, biggan=True).state_dict())}
    with open(os.path.join(OUT, "encbig_keys.json"), "w") as f:
        json.dump(keys, f)
    E = EBG.BE(startf=32, maxf=512, layer_count=5, biggan=True)      # 64x64 input -> [B,512,4,4] -> 8192 features
    sd = R.fill_encbig(shapes_of(E.state_dict()), seed=81)
    E.load_state_dict(sd)
    E.eval()
    img = R.randn("ebg.img", (2, 3, 64, 64), 81, 0.5)
    cond = R.randn("ebg.cond", (2, 256), 81, 0.5)
    feats = {}
    h = E.decode_block[4].register_forward_hook(lambda m_, i, o: feats.__setitem__("trunk", o[0].detach().clone()))
    with torch.no_grad(), _NoiseFeeder("ebg", 81) as nf:
        c_v, z = E(img, cond)
    h.remove()
    save_npz("encbig_small.npz", trunk=feats["trunk"], c_v=c_v, z=z, noise_shapes=np.array([list(s_) for s_ in nf.log]),
             state_checksum=np.array(R.checksum(sd)))


SECTIONS["encbig"] = gen_encbig

def gen_sg1grad():
    """Gradient of a seeded linear functional of the StyleGAN1 image w.r.t. the styles (the quantity the
    E_align loop back-propagates through Gs, E_align_s2.py:158,204), same generator/inputs/noise as gen_sg1."""
    import model.stylegan1.net as SG1
    G = SG1.Generator(startf=32, maxf=64, layer_count=6, latent_size=512, channels=3)
    sd = R.fill_encoder(shapes_of(G.state_dict()), seed=41)
    for k in sd:
        if k.endswith("blur.weight"):
            sd[k] = G.state_dict()[k].clone()
        if k == "const":
            sd[k] = R.randn("sg1.const", tuple(sd[k].shape), 41)
    G.load_state_dict(sd)
    out = {}
    cases = (("", 5, "sg1"), ("_lod3", 3, "sg1b"))

    def runner(lod, prefix):
        def run():
            with _NoiseFeeder(prefix, 6):
                return G.forward(R.randn("sg1.styles", (2, 12, 512), 6), lod)
        return run
    owners, names = [], []
    for j, b in enumerate(G.decode_block):          # DecodeBlock.forward: lrelu behind bias_1 and behind bias_2 (stylegan1/net.py:150-152,162-164)
        owners += [b.bias_1, b.bias_2]; names += [f"decode_block.{j}.bias_1", f"decode_block.{j}.bias_2"]
    nudged, margin = clear_kinks([runner(lod, prefix) for _, lod, prefix in cases], owners, names)
    out["kink_margin"] = np.array(margin)
    out.update({"param:" + k: v for k, v in nudged.items()})
    for tag, lod, prefix in cases:
        styles = R.randn("sg1.styles", (2, 12, 512), 6).requires_grad_(True)
        with _NoiseFeeder(prefix, 6):
            img = G.forward(styles, lod)
        out["image" + tag] = img.detach()
        gimg = R.randn("sg1.gimg" + tag, tuple(img.shape), 7)
        (img * gimg).sum().backward()
        out["g_styles" + tag] = styles.grad
        out["loss" + tag] = (img * gimg).sum().detach()
    save_npz("sg1_grad.npz", **out)


SECTIONS["sg1grad"] = gen_sg1grad


def gen_step_sg1():
    """One full E_align_s2 iteration for --mtype 1 (StyleGAN1: Gm -> Gs.forward -> E -> Gs.forward, E_align_s2.py:27-46,
    105-108,157-158) of the REFERENCE at reduced size (res 64, 5 blocks), two iterations, every noise tensor captured."""
    import warnings
    import model.stylegan1.net as SG1
    import model.E.E as EE
    import training_utils as TU
    from model.utils.custom_adam import LREQAdam
    from oracle import lpips_ref as LR

    L = 5
    Gs = SG1.Generator(startf=16, maxf=64, layer_count=L, latent_size=512, channels=3)
    sd = R.fill_encoder(shapes_of(Gs.state_dict()), seed=43)
    for k in sd:
        if k.endswith("blur.weight"):
            sd[k] = Gs.state_dict()[k].clone()
        if k == "const":
            sd[k] = R.randn("sg1step.const", tuple(sd[k].shape), 43)
    Gs.load_state_dict(sd)
    Gm = SG1.Mapping(num_layers=2 * L, mapping_layers=8, latent_size=512, dlatent_size=512, mapping_fmaps=512)
    Gm.load_state_dict({k: R.randn("sg1step.m." + k, tuple(v.shape), 44, 0.05 if k.endswith("weight") else 0.01)
                        for k, v in Gm.state_dict().items()})
    Gm.buffer1 = R.randn("sg1step.buffer1", (2 * L, 512), 44, 0.5)
    Gm.eval()
    layer_idx = torch.arange(2 * L)[np.newaxis, :, np.newaxis]
    ones = torch.ones(layer_idx.shape, dtype=torch.float32)
    coefs = torch.where(layer_idx < L, 0.7 * ones, ones)
    E = EE.BE(startf=16, maxf=64, layer_count=L)
    E.load_state_dict(R.fill_encoder(shapes_of(E.state_dict()), seed=31))
    LP = LR.seeded_params(0)
    lp = lambda a, b: LR.lpips(LP, a, b)
    opt = LREQAdam([{"params": E.parameters()}], lr=0.0015, betas=(0.0, 0.99), weight_decay=0)
    out = {}
    B = 2
    lod = L - 1
    for it in range(2):
        z = R.randn(f"sg1step.z{it}", (B, 512), 1)
        with _NoiseFeeder(f"sg1step.it{it}", 1) as nf:
            with torch.no_grad():
                w1 = Gm(z, coefs_m=coefs)
                imgs1 = Gs.forward(w1, lod)
            const2, w2 = E(imgs1)
            imgs2 = Gs.forward(w2, lod)
        if it == 0:
            out["noise_shapes"] = np.array([list(s_) + [0] * (4 - len(s_)) for s_ in nf.log])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            l_i, i_i = TU.space_loss(imgs1, imgs2, lpips_model=lp)
            m1 = imgs1[:, :, :, imgs1.shape[3] // 8:-imgs1.shape[3] // 8]
            m2 = imgs2[:, :, :, imgs2.shape[3] // 8:-imgs2.shape[3] // 8]
            l_m, i_m = TU.space_loss(m1, m2, lpips_model=lp)
            o = imgs1.shape[2] // 8 + imgs1.shape[2] // 32
            s1, s2 = imgs1[:, :, o:-o, o:-o], imgs2[:, :, o:-o, o:-o]
            l_s, i_s = TU.space_loss(s1, s2, lpips_model=lp)
            loss_tsa = l_i + l_m * 5 + l_s * 9
            opt.zero_grad()
            loss_tsa.backward(retain_graph=True)
            opt.step()
            l_w, i_w = TU.space_loss(w1, w2, image_space=False)
            loss_mtv = l_w * 0.01
            opt.zero_grad()
            loss_mtv.backward()
            opt.step()
        flat = lambda inf: [inf[0][0], inf[0][1], inf[0][2], inf[1], inf[2], inf[3], inf[4]]
        out[f"it{it}_imgs1"] = imgs1
        out[f"it{it}_w1"] = w1
        out[f"it{it}_w2"] = w2.detach()
        out[f"it{it}_imgs2"] = imgs2.detach()
        out[f"it{it}_losses"] = np.array([float(loss_tsa), float(l_i), float(l_m), float(l_s), float(l_w)])
        out[f"it{it}_info"] = np.array([flat(i_i), flat(i_m), flat(i_s), flat(i_w)])
        out[f"it{it}_param_checksum"] = np.array(R.checksum(E.state_dict()))
        for k in ("decode_block.0.conv_1.weight", "decode_block.2.conv_2.weight", "decode_block.4.inver_mod2.weight",
                  "decode_block.1.bias_1", "FromRGB.from_rgb.weight"):
            out[f"it{it}_after_phase2:{k}"] = E.state_dict()[k].clone()
    save_npz("step_sg1.npz", **out)


SECTIONS["step_sg1"] = gen_step_sg1
SECTIONS["step_s1"] = gen_step_s1
SECTIONS["step_s1_legacy"] = lambda: gen_step_s1(legacy=True)

def gen_encblurgrad():
    """Gradients of the reference E_Blur.BE w.r.t. every parameter AND the input image for a seeded linear functional of
    both outputs (x, w) -- what embedding_img.py:86-127 back-propagates.  Same model / input / noise as encblur_small.npz."""
    import model.E.E_Blur as EB
    E = EB.BE(startf=16, maxf=64, layer_count=6)
    sd = R.fill_encoder(shapes_of(E.state_dict()), seed=61)
    for k in sd:
        if k.endswith("blur.weight"):
            sd[k] = E.state_dict()[k].clone()
    E.load_state_dict(sd)
    img = R.randn("eb.img", (2, 3, 128, 128), 61, 0.5).requires_grad_(True)

    def run():
        with _NoiseFeeder("eb", 61):
            return E(img)
    nudged, margin = clear_kinks([run], *enc_kink_owners(E, "has_last_conv"))
    x, w = run()
    gx, gw = R.randn("eb.gx", tuple(x.shape), 63), R.randn("eb.gw", tuple(w.shape), 63)
    loss = (x * gx).sum() + (w * gw).sum()
    loss.backward()
    out = {"loss": loss.detach(), "g_img": img.grad, "x": x.detach(), "w": w.detach(),
           "kink_margin": np.array(margin), **{"param:" + k: v for k, v in nudged.items()}}
    for k, p_ in E.named_parameters():
        if p_.grad is None:
            continue
        g = p_.grad
        out["norm:" + k] = g.norm()
        out["grad:" + k] = g if g.numel() <= 40000 else g.flatten()[:4096]
    save_npz("encblur_grad.npz", **out)


SECTIONS["encblurgrad"] = gen_encblurgrad

def gen_embed():
    """Two iterations of the reference's inversion loop body (embedding_img.py:84-127, optimizeE=True, batch 1) at reduced
    size: StyleGAN1 Gs (5 blocks, 64x64) + E_Blur (5 blocks), seeded stand-in LPIPS, every noise tensor captured."""
    import warnings
    import model.stylegan1.net as SG1
    import model.E.E_Blur as EB
    import training_utils as TU
    from model.utils.custom_adam import LREQAdam
    from oracle import lpips_ref as LR

    L = 5
    Gs = SG1.Generator(startf=16, maxf=64, layer_count=L, latent_size=512, channels=3)
    sd = R.fill_encoder(shapes_of(Gs.state_dict()), seed=43)
    for k in sd:
        if k.endswith("blur.weight"):
            sd[k] = Gs.state_dict()[k].clone()
        if k == "const":
            sd[k] = R.randn("sg1step.const", tuple(sd[k].shape), 43)
    Gs.load_state_dict(sd)
    E = EB.BE(startf=16, maxf=64, layer_count=L)
    esd = R.fill_encoder(shapes_of(E.state_dict()), seed=71)
    for k in esd:
        if k.endswith("blur.weight"):
            esd[k] = E.state_dict()[k].clone()
    E.load_state_dict(esd)
    LP = LR.seeded_params(0)
    lp = lambda a, b: LR.lpips(LP, a, b)
    opt = LREQAdam([{"params": E.parameters()}], lr=0.01, betas=(0.0, 0.99), weight_decay=0)
    imgs1 = torch.tanh(R.randn("embed.img", (1, 3, 64, 64), 72, 0.8))
    lod = L - 1
    out = {"imgs1": imgs1}
    for it in range(2):
        with _NoiseFeeder(f"embed.it{it}", 2) as nf:
            const2, w1 = E(imgs1)
            imgs2 = Gs.forward(w1, lod)
            const3, w2 = E(imgs2)
        if it == 0:
            out["noise_shapes"] = np.array([list(s_) for s_ in nf.log])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            loss_imgs, _ = TU.space_loss(imgs1, imgs2, lpips_model=lp)
            m1 = imgs1[:, :, :, imgs1.shape[3] // 8:-imgs1.shape[3] // 8].detach().clone()
            m2 = imgs2[:, :, :, imgs2.shape[3] // 8:-imgs2.shape[3] // 8].detach().clone()
            loss_medium, _ = TU.space_loss(m1, m2, lpips_model=lp)
            o = imgs1.shape[2] // 8 + imgs1.shape[2] // 32
            s1, s2 = imgs1[:, :, o:-o, o:-o].detach().clone(), imgs2[:, :, o:-o, o:-o].detach().clone()
            loss_small, _ = TU.space_loss(s1, s2, lpips_model=lp)
            opt.zero_grad()
            loss_msiv = loss_imgs + (loss_medium + loss_small) * 0.125
            loss_msiv.backward(retain_graph=True)
            out[f"it{it}_grad1:decode_block.0.conv_1.weight"] = E.decode_block[0].conv_1.weight.grad.clone()
            out[f"it{it}_grad1:decode_block.3.inver_mod2.weight"] = E.decode_block[3].inver_mod2.weight.grad.clone()
            opt.step()
            loss_w, _ = TU.space_loss(w1, w2, image_space=False)
            loss_c1, _ = TU.space_loss(const2, const3, image_space=False)
            opt.zero_grad()
            loss_mslv = (loss_w + loss_c1) * 0.01
            loss_mslv.backward()
            out[f"it{it}_grad2:decode_block.0.conv_1.weight"] = E.decode_block[0].conv_1.weight.grad.clone()
            out[f"it{it}_grad2:decode_block.2.conv_2.weight"] = E.decode_block[2].conv_2.weight.grad.clone()
            out[f"it{it}_grad2:decode_block.4.inver_mod1.weight"] = E.decode_block[4].inver_mod1.weight.grad.clone()
            out[f"it{it}_grad2:FromRGB.from_rgb.weight"] = E.FromRGB.from_rgb.weight.grad.clone()
            opt.step()
        out[f"it{it}_w1"] = w1.detach()
        out[f"it{it}_w2"] = w2.detach()
        out[f"it{it}_imgs2"] = imgs2.detach()
        out[f"it{it}_const2"] = const2.detach()
        out[f"it{it}_const3"] = const3.detach()
        out[f"it{it}_losses"] = np.array([float(loss_msiv), float(loss_imgs), float(loss_medium), float(loss_small), float(loss_w), float(loss_c1)])
        out[f"it{it}_param_checksum"] = np.array(R.checksum({k: v for k, v in E.state_dict().items() if not k.endswith("blur.weight")}))
    save_npz("embed_sg1.npz", **out)


SECTIONS["embed"] = gen_embed

def gen_pggangrad():
    """Gradient of a seeded linear functional of the PGGAN image w.r.t. z (same generator / z as pggan_small.npz)."""
    import contextlib, io
    from model.pggan.pggan_generator import PGGANGenerator
    G = PGGANGenerator(32, fmaps_base=1024, fmaps_max=64)
    sd = {k: (R.randn("pg." + k, tuple(v.shape), 51, 0.2 if k.endswith("bias") else 1.0) if v.ndim else v.clone())
          for k, v in G.state_dict().items()}
    G.load_state_dict(sd)
    z = R.randn("pg.z", (2, 512), 51).requires_grad_(True)
    with contextlib.redirect_stdout(io.StringIO()):
        img = G(z)["image"]
    gimg = R.randn("pg.gimg", tuple(img.shape), 52)
    loss = (img * gimg).sum()
    loss.backward()
    save_npz("pggan_grad.npz", g_z=z.grad, loss=loss.detach())


SECTIONS["pggangrad"] = gen_pggangrad

def gen_encpggrad():
    """Gradients of the reference E_PG.BE w.r.t. every parameter for a seeded linear functional of the head output
    new_final(x.view(B,-1)) -- the value the reference computes and then discards; it is taken live (with its
    autograd graph) from a forward hook on the reference's own module.  Same model / input / noise as encpg_small.npz."""
    import model.E.E_PG as EP
    E = EP.BE(startf=32, maxf=512, layer_count=5, pggan=True)
    sd = R.fill_encoder(shapes_of(E.state_dict()), seed=62)
    for k in sd:
        if "instance_norm_3.weight" in k:
            sd[k] = R.randn("pg." + k, tuple(sd[k].shape), 62, 0.2, 1.0)
    E.load_state_dict(sd)
    img = R.randn("ep.img", (2, 3, 64, 64), 62, 0.5)

    def run():
        with _NoiseFeeder("ep", 62):
            E(img)
    nudged, margin = clear_kinks([run], *enc_kink_owners(E, "has_second_conv"))
    feats = {}
    h = E.new_final.register_forward_hook(lambda m_, i, o: feats.__setitem__("head", o))
    run()
    h.remove()
    z = feats["head"]
    loss = (z * R.randn("ep.gz", tuple(z.shape), 64)).sum()
    loss.backward()
    out = {"loss": loss.detach(), "kink_margin": np.array(margin), **{"param:" + k: v for k, v in nudged.items()}}
    for k, p_ in E.named_parameters():
        if p_.grad is None:
            continue
        g = p_.grad
        out["norm:" + k] = g.norm()
        out["grad:" + k] = g if g.numel() <= 40000 else g.flatten()[:4096]
    save_npz("encpg_grad.npz", **out)


SECTIONS["encpggrad"] = gen_encpggrad

def gen_step_pg():
    """Two E_align_s2 iterations for --mtype 3 (PGGAN) with the reference's own modules, in the evident-intent form of
    SURVEY Q5 (the script as shipped crashes: E_PG returns (0, 0) and PGGANGenerator has no .synthesis): w2 = the live
    output of the reference's `new_final` (forward hook), imgs2 = generator(w2)['image'] (E_align_s2.py:134-138,153,160).
    Reduced size: PGGAN res 64, E_PG 5 blocks."""
    import contextlib, io, warnings
    from model.pggan.pggan_generator import PGGANGenerator
    import model.E.E_PG as EP
    import training_utils as TU
    from model.utils.custom_adam import LREQAdam
    from oracle import lpips_ref as LR

    G = PGGANGenerator(64, fmaps_base=1024, fmaps_max=64)
    G.load_state_dict({k: (R.randn("pgstep." + k, tuple(v.shape), 51, 0.2 if k.endswith("bias") else 1.0) if v.ndim else v.clone())
                       for k, v in G.state_dict().items()})
    E = EP.BE(startf=32, maxf=512, layer_count=5, pggan=True)
    sd = R.fill_encoder(shapes_of(E.state_dict()), seed=62)
    for k in sd:
        if "instance_norm_3.weight" in k:
            sd[k] = R.randn("pg." + k, tuple(sd[k].shape), 62, 0.2, 1.0)
    E.load_state_dict(sd)
    feats = {}
    E.new_final.register_forward_hook(lambda m_, i, o: feats.__setitem__("head", o))
    LP = LR.seeded_params(0)
    lp = lambda a, b: LR.lpips(LP, a, b)
    opt = LREQAdam([{"params": E.parameters()}], lr=0.0015, betas=(0.0, 0.99), weight_decay=0)
    out = {}
    B = 2
    for it in range(2):
        z = R.randn(f"pgstep.z{it}", (B, 512), 1)
        with _NoiseFeeder(f"pgstep.it{it}", 1) as nf, contextlib.redirect_stdout(io.StringIO()):
            with torch.no_grad():
                w1 = z
                imgs1 = G(w1)["image"]
            E(imgs1)
            w2 = feats["head"]
            imgs2 = G(w2)["image"]
        if it == 0:
            out["noise_shapes"] = np.array([list(s_) for s_ in nf.log])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            l_i, i_i = TU.space_loss(imgs1, imgs2, lpips_model=lp)
            m1 = imgs1[:, :, :, imgs1.shape[3] // 8:-imgs1.shape[3] // 8]
            m2 = imgs2[:, :, :, imgs2.shape[3] // 8:-imgs2.shape[3] // 8]
            l_m, i_m = TU.space_loss(m1, m2, lpips_model=lp)
            o = imgs1.shape[2] // 8 + imgs1.shape[2] // 32
            s1, s2 = imgs1[:, :, o:-o, o:-o], imgs2[:, :, o:-o, o:-o]
            l_s, i_s = TU.space_loss(s1, s2, lpips_model=lp)
            loss_tsa = l_i + l_m * 5 + l_s * 9
            opt.zero_grad()
            loss_tsa.backward(retain_graph=True)
            opt.step()
            l_w, i_w = TU.space_loss(w1, w2, image_space=False)
            loss_mtv = l_w * 0.01
            opt.zero_grad()
            loss_mtv.backward()
            opt.step()
        flat = lambda inf: [inf[0][0], inf[0][1], inf[0][2], inf[1], inf[2], inf[3], inf[4]]
        out[f"it{it}_imgs1"] = imgs1
        out[f"it{it}_w2"] = w2.detach()
        out[f"it{it}_imgs2"] = imgs2.detach()
        out[f"it{it}_losses"] = np.array([float(loss_tsa), float(l_i), float(l_m), float(l_s), float(l_w)])
        out[f"it{it}_info"] = np.array([flat(i_i), flat(i_m), flat(i_s), flat(i_w)])
        out[f"it{it}_param_checksum"] = np.array(R.checksum(E.state_dict()))
        for k in ("decode_block.0.conv_1.weight", "decode_block.2.conv_2.weight", "decode_block.1.conv_3.weight",
                  "decode_block.1.instance_norm_3.weight", "decode_block.1.bias_1", "FromRGB.from_rgb.weight", "new_final.bias"):
            out[f"it{it}_after_phase2:{k}"] = E.state_dict()[k].clone()
    save_npz("step_pg.npz", **out)


SECTIONS["step_pg"] = gen_step_pg

def gen_biggangrad():
    """Gradient of a seeded linear functional of the BigGAN-deep image w.r.t. z (same generator / inputs as biggan_small.npz,
    eval mode) -- what E_align_s2.py:162 back-propagates into the encoder."""
    _stub("boto3"); _stub("botocore"); _stub("botocore.exceptions", ClientError=Exception)
    _stub("requests")
    from model.biggan_generator import BigGAN
    from model.utils.biggan_config import BigGANConfig
    G = BigGAN(BigGANConfig.from_dict(BIGGAN_SMALL_CFG))
    G.load_state_dict(R.fill_biggan(shapes_of(G.state_dict()), seed=71))
    G.eval()
    z = R.randn("bg.z", (2, 128), 71, 0.4).requires_grad_(True)
    onehot = torch.zeros(2, 1000); onehot[:, 207] = 1.0
    img, cond = G(z, onehot, 0.4)
    loss = (img * R.randn("bg.gimg", tuple(img.shape), 72)).sum()
    loss.backward()
    save_npz("biggan_grad.npz", g_z=z.grad, loss=loss.detach())


SECTIONS["biggangrad"] = gen_biggangrad

def gen_encbiggrad():
    """Gradients of the reference E_BIG.BE w.r.t. every parameter for a seeded linear functional of both outputs (c_v, z),
    in TRAIN mode (the training script never calls .eval(): every conditional-BN linear runs one spectral-norm power
    iteration in the forward, and the gradient w.r.t. weight_orig goes through sigma).  Model / inputs of encbig_small.npz."""
    import model.E.E_BIG as EBG
    E = EBG.BE(startf=32, maxf=512, layer_count=5, biggan=True)
    E.load_state_dict(R.fill_encbig(shapes_of(E.state_dict()), seed=81))
    E.train()
    img = R.randn("ebg.img", (2, 3, 64, 64), 81, 0.5)
    cond = R.randn("ebg.cond", (2, 256), 81, 0.5)
    # every train-mode forward runs one spectral-norm power iteration IN PLACE on weight_u / weight_v: the buffers are put back
    # before each pass so that the nudging passes and the recorded pass all see the fixture's state (= one iteration from it)
    uv = {k: v.clone() for k, v in E.state_dict().items() if k.endswith(("weight_u", "weight_v"))}

    def run():
        with torch.no_grad():
            for k, v in E.state_dict().items():
                if k in uv:
                    v.copy_(uv[k])
        with _NoiseFeeder("ebg", 81):
            return E(img, cond)
    nudged, margin = clear_kinks([run], *enc_kink_owners(E, "has_second_conv", third=True))
    c_v, z = run()
    loss = (z * R.randn("ebg.gz", tuple(z.shape), 82)).sum() + (c_v * R.randn("ebg.gcv", tuple(c_v.shape), 82)).sum()
    loss.backward()
    out = {"loss": loss.detach(), "c_v": c_v.detach(), "z": z.detach(),
           "kink_margin": np.array(margin), **{"param:" + k: v for k, v in nudged.items()}}
    for k, p_ in E.named_parameters():
        if p_.grad is None:
            continue
        g = p_.grad
        out["norm:" + k] = g.norm()
        out["grad:" + k] = g if g.numel() <= 40000 else g.flatten()[:4096]
    save_npz("encbig_grad.npz", **out)


SECTIONS["encbiggrad"] = gen_encbiggrad


def gen_step_big():
    """Two E_align_s2 iterations for --mtype 4 (BigGAN-deep + E_BIG, E_align_s2.py:79-86,139-162) with the reference's own
    modules at reduced size (generator config BIGGAN_SMALL_CFG -> 64x64, E_BIG 5 blocks).  Both networks stay in train
    mode as in the script (one spectral-norm power iteration per forward: SURVEY Q2); `truncation` is the float32 tensor
    0.4 the script builds (tensor / step_size is a float32 division: exactly row 20, no interpolation); E_BIG's blocks use the
    Python float 0.4.  z comes from scipy's truncnorm and is stored as a fixture;
    the label index is the script's np.random.randint(1000) after set_seed (cast to an integer index: the script's float
    index array is rejected by torch.eye(...)[x,:])."""
    import warnings
    _stub("boto3"); _stub("botocore"); _stub("botocore.exceptions", ClientError=Exception)
    _stub("requests")
    from model.biggan_generator import BigGAN
    from model.utils.biggan_config import BigGANConfig
    import model.E.E_BIG as EBG
    import training_utils as TU
    from model.utils.custom_adam import LREQAdam
    from oracle import lpips_ref as LR

    G = BigGAN(BigGANConfig.from_dict(BIGGAN_SMALL_CFG))
    G.load_state_dict(R.fill_biggan(shapes_of(G.state_dict()), seed=71))
    E = EBG.BE(startf=32, maxf=512, layer_count=5, biggan=True)
    E.load_state_dict(R.fill_encbig(shapes_of(E.state_dict()), seed=81))
    LP = LR.seeded_params(0)
    lp = lambda a, b: LR.lpips(LP, a, b)
    opt = LREQAdam([{"params": E.parameters()}], lr=0.0015, betas=(0.0, 0.99), weight_decay=0)
    out = {}
    B = 2
    for it in range(2):
        TU.set_seed(it % 30000)
        z = TU.truncated_noise_sample(truncation=0.4, batch_size=B, seed=it % 30000)
        flag = np.random.randint(1000)
        label = TU.one_hot((flag * np.ones(B)).astype(np.int64))
        w1 = torch.tensor(z, dtype=torch.float)
        conditions = torch.tensor(label, dtype=torch.float)
        truncation = torch.tensor(0.4, dtype=torch.float)
        with _NoiseFeeder(f"bigstep.it{it}", 1) as nf:
            with torch.no_grad():
                imgs1, const1 = G(w1, conditions, truncation)
            const2, w2 = E(imgs1, const1)
            imgs2, _ = G(w2, conditions, truncation)
        if it == 0:
            out["noise_shapes"] = np.array([list(s_) for s_ in nf.log])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            l_i, i_i = TU.space_loss(imgs1, imgs2, lpips_model=lp)
            m1 = imgs1[:, :, :, imgs1.shape[3] // 8:-imgs1.shape[3] // 8]
            m2 = imgs2[:, :, :, imgs2.shape[3] // 8:-imgs2.shape[3] // 8]
            l_m, i_m = TU.space_loss(m1, m2, lpips_model=lp)
            o = imgs1.shape[2] // 8 + imgs1.shape[2] // 32
            s1, s2 = imgs1[:, :, o:-o, o:-o], imgs2[:, :, o:-o, o:-o]
            l_s, i_s = TU.space_loss(s1, s2, lpips_model=lp)
            loss_tsa = l_i + l_m * 5 + l_s * 9
            opt.zero_grad()
            loss_tsa.backward(retain_graph=True)
            opt.step()
            l_w, i_w = TU.space_loss(w1, w2, image_space=False)
            loss_mtv = l_w * 0.01
            opt.zero_grad()
            loss_mtv.backward()
            opt.step()
        flat = lambda inf: [inf[0][0], inf[0][1], inf[0][2], inf[1], inf[2], inf[3], inf[4]]
        out[f"it{it}_z"] = w1
        out[f"it{it}_flag"] = np.array(flag)
        out[f"it{it}_imgs1"] = imgs1
        out[f"it{it}_const1"] = const1
        out[f"it{it}_const2"] = const2.detach()
        out[f"it{it}_w2"] = w2.detach()
        out[f"it{it}_imgs2"] = imgs2.detach()
        out[f"it{it}_losses"] = np.array([float(loss_tsa), float(l_i), float(l_m), float(l_s), float(l_w)])
        out[f"it{it}_info"] = np.array([flat(i_i), flat(i_m), flat(i_s), flat(i_w)])
        out[f"it{it}_param_checksum"] = np.array(R.checksum({k: v for k, v in E.state_dict().items() if v.dtype.is_floating_point}))
        for k in ("decode_block.0.conv_1.weight", "decode_block.2.conv_2.weight", "decode_block.1.conv_3.weight",
                  "decode_block.1.batch_norm_1.scale.weight_orig", "decode_block.0.batch_norm_3.offset.weight_orig",
                  "decode_block.1.batch_norm_2.scale.weight_u", "decode_block.1.bias_1", "FromRGB.from_rgb.weight", "new_final_2.bias"):
            out[f"it{it}_after_phase2:{k}"] = E.state_dict()[k].clone()
    save_npz("step_big.npz", **out)


SECTIONS["step_big"] = gen_step_big

def gen_latent_fixtures():
    """The reference's shipped data files for the latent-editing / inversion demos (latent_code/directions/*.npy: InterfaceGAN
    boundaries [1,512] f64; latent_code/real_face_code/*.pt: W+ codes [1,18,512]) as one small fixture, plus the edited code
    embeded_img_edit.py:38-41 produces from them (its exact indexing: rows start..start+end of w + bonus*direction)."""
    out = {}
    root = os.path.join(REF, "latent_code")
    for name in ("age", "eyeglasses", "gender", "pose", "smile"):
        out["dir_" + name] = np.load(os.path.join(root, "directions", f"stylegan_ffhq_{name}_w_boundary.npy"))
    for name in ("i4_msk", "i5_ty"):
        out["w_" + name] = torch.load(os.path.join(root, "real_face_code", name + ".pt"), map_location="cpu").detach().clone().float()
    # the script's own arithmetic (embeded_img_edit.py:28-41) for two settings
    for tag, dname, bonus, start, end in (("a", "eyeglasses", 70, 0, 3), ("b", "smile", 100, 0, 4)):
        direction = torch.tensor(out["dir_" + dname]).float().expand(18, 512)
        w = out["w_i4_msk"].clone().squeeze(0)
        w[start:start + end] = (w + bonus * direction)[start:start + end]
        out["edit_" + tag] = w.reshape(1, 18, 512)
    save_npz("latent_fixtures.npz", **out)


SECTIONS["latent"] = gen_latent_fixtures

if __name__ == "__main__":
    todo = sys.argv[1:] or list(SECTIONS)
    for s_ in todo:
        print("==", s_)
        SECTIONS[s_]()
