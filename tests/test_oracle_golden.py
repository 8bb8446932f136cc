"""Pins the oracle (oracle/ref_torch.py) against the golden vectors produced by the
reference itself (tools/gen_golden.py).  CPU only."""
import json
import math
import os

import numpy as np
import pytest
import torch

from tests.conftest import golden, ROOT, with_fixture_params
from tests.golden import recipe as R
from oracle import ref_torch as O

T = torch.from_numpy


def close(a, b, rtol=2e-4, atol=2e-5):
    a = a.detach() if torch.is_tensor(a) else torch.as_tensor(a)
    b = torch.as_tensor(np.asarray(b))
    scale = b.abs().max().item() + 1e-30
    err = (a.double() - b.double()).abs().max().item()
    assert err <= atol + rtol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


def s2_small_params():
    from tests.helpers import s2_shapes
    shapes = s2_shapes(64, fmaps_base=2048, fmaps_max=128)
    return R.fill_s2(shapes, seed=11)


def test_s2_blocks():
    g = golden("s2_blocks.npz")
    from tests.helpers import modconv_shapes
    for ci in range(6):
        cin, cout, res, up, k = [int(v) for v in g[f"c{ci}_cfg"]]
        torgb = (k == 1)
        shapes = modconv_shapes(cin, cout, res, k, noise=not torgb, up=bool(up))
        P = {"L." + kk: v for kk, v in R.fill_s2(shapes, seed=100 + ci).items()}
        rin = res // 2 if up else res
        x = R.randn(f"mc{ci}.x", (2, cin, rin, rin), 7)
        w = R.randn(f"mc{ci}.w", (2, 512), 7)
        y, s = O.s2_modconv(P, "L", x, w, up=bool(up), demodulate=not torgb, add_noise=not torgb,
                            act="linear" if torgb else "lrelu")
        close(s, g[f"c{ci}_style"])
        close(y, g[f"c{ci}_y"])
        close(y, g[f"c{ci}_y_nonfused"])
    close(O.s2_upsample_skip(R.randn("ups.up", (2, 3, 8, 8), 3)), g["ups_up_y"])
    close(O.s2_filter_after_up(R.randn("ups.filt", (2, 3, 9, 9), 3)), g["ups_filt_y"])


def test_s2_synthesis_and_generator():
    g = golden("s2_small.npz")
    P = s2_small_params()
    assert abs(R.checksum(P) - float(g["state_checksum"])) < 1e-6 * float(g["state_checksum"])
    wp = R.randn("s2.wp", (2, 10, 512), 5)
    feats = {}
    img = O.s2_synthesis(P, wp, collect=feats)
    for name in ("layer0", "layer1", "layer2", "layer7", "layer8"):
        close(feats[name], g["syn_" + name])
    close(img, g["syn_image"])
    z = R.randn("s2.z", (2, 512), 5)
    w, wp2, img2 = O.s2_generator_eval(P, z)
    close(w, g["eval_w"]); close(wp2, g["eval_wp"]); close(img2, g["eval_image"])
    # layers >= trunc_layers carry coef 1: w_avg + (w - w_avg)*1 (not bit-identical to w, as in the reference)
    close(wp2[:, 8:], w[:, None].repeat(1, 2, 1), rtol=1e-6, atol=5e-7)


def test_s2_train_mode_quirk_q1():
    g = golden("s2_small.npz")
    P = s2_small_params()
    z = R.randn("s2.z", (2, 512), 5)
    new_z = R.randn("s2.new_z", (2, 512), 5)
    for tag in ("a", "b"):
        u, cutoff = g[f"train_{tag}_u_cutoff"]
        P["truncation.w_avg"] = T(g[f"train_{tag}_w_avg_before"])
        wp, w_avg = O.s2_generator_train(P, z, new_z, float(u), int(cutoff))
        close(w_avg, g[f"train_{tag}_w_avg_after"])
        close(wp, g[f"train_{tag}_wp"])
        P2 = dict(P); P2["truncation.w_avg"] = w_avg
        close(O.s2_synthesis(P2, wp), g[f"train_{tag}_image"])


def test_s2_grad_wp():
    g = golden("s2_small.npz")
    P = s2_small_params()
    wp = R.randn("s2.wp", (2, 10, 512), 5).requires_grad_(True)
    img = O.s2_synthesis(P, wp)
    gimg = R.randn("s2.gimg", tuple(img.shape), 5, 1.0 / img.numel() ** 0.5)
    (img * gimg).sum().backward()
    close(wp.grad, g["grad_wp"], rtol=1e-3)


def enc_small_params():
    from tests.helpers import enc_shapes
    return with_fixture_params(R.fill_encoder(enc_shapes(16, 64, 4), seed=21), golden("enc_small.npz"))


def test_encoder_forward_backward():
    g = golden("enc_small.npz")
    P = {k: v.requires_grad_(True) for k, v in enc_small_params().items()}
    assert abs(R.checksum(P) - float(g["state_checksum"])) < 1e-6 * float(g["state_checksum"])
    img = R.randn("enc.img", (2, 3, 32, 32), 9, 0.5).requires_grad_(True)
    shp = O.enc_noise_shapes(4, 2, 32)
    assert [list(s) for s in shp] == g["noise_shapes"].tolist()
    noises = [R.randn(f"enc.noise{i}", s, 9) for i, s in enumerate(shp)]
    feats = {}
    x, w = O.enc_forward(P, img, noises, collect=feats)
    close(x, g["x"]); close(w, g["w"])
    # bit-exact index map: w[:, 2(L-1-j)] = w2_j, w[:, 2(L-1-j)+1] = w1_j (E.py:130-134)
    for j in range(4):
        assert torch.equal(w[:, 2 * (3 - j)], feats[j][2]) and torch.equal(w[:, 2 * (3 - j) + 1], feats[j][1])
        close(feats[j][0], g[f"blk{j}_x"])
    gw = R.randn("enc.gw", tuple(w.shape), 9, 0.05)
    (w * gw).sum().backward()
    close(img.grad, g["grad_img"], rtol=1e-3)
    for k, p in P.items():
        if "grad:" + k in g.files:
            close(p.grad, g["grad:" + k], rtol=1e-3)
        else:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k


def test_losses():
    g = golden("loss.npz")
    a = R.randn("loss.a", (2, 3, 64, 64), 1, 0.5).clamp(-1, 1)
    b = (a + R.randn("loss.b", (2, 3, 64, 64), 1, 0.1)).clamp(-1, 1)
    close(O.ssim(a, b), g["ssim_64"]); close(O.ssim(a, a), g["ssim_same"])
    assert abs(float(O.ssim(a, a)) - 1.0) < 1e-6      # comparing-baseline.py:88 known answer
    a2 = R.randn("loss.a2", (1, 3, 40, 24), 1, 0.5)
    close(O.ssim(a2, a2 * 0.7 + 0.1), g["ssim_40x24"])
    standin = lambda x, y: ((x - y) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    big_a = R.randn("loss.big_a", (2, 3, 512, 384), 2, 0.5)
    big_b = (big_a + R.randn("loss.big_b", (2, 3, 512, 384), 2, 0.2)).requires_grad_(True)
    l, info = O.space_loss(big_a, big_b, lpips_fn=standin)
    l.backward()
    close(l, g["img_loss"])
    ref = g["img_info"]
    for i, k in enumerate(("mse", "mse_mean", "mse_std", "kl", "cos", "ssim", "lpips")):
        assert abs(float(info[k]) - ref[i]) <= 2e-4 * abs(ref[i]) + 1e-6, k
    close(big_b.grad[:, :, 100:116, 200:216], g["img_grad_b_crop"], rtol=1e-3)
    w1 = R.randn("loss.w1", (2, 18, 512), 2)
    w2 = (w1 * 0.9 + R.randn("loss.w2", (2, 18, 512), 2, 0.3)).requires_grad_(True)
    l, info = O.space_loss(w1, w2, image_space=False)
    l.backward()
    close(l, g["w_loss"]); close(w2.grad, g["w_grad"], rtol=1e-3)
    ref = g["w_info"]
    for i, k in enumerate(("mse", "mse_mean", "mse_std", "kl", "cos")):
        assert abs(float(info[k]) - ref[i]) <= 2e-4 * abs(ref[i]) + 1e-6, k


def test_lreq_adam():
    g = golden("adam.npz")
    names = ["lin.weight", "lin.bias", "conv.weight", "plain"]
    shapes = [(7, 12), (7,), (6, 4, 3, 3), (1, 6, 1, 1)]
    coef = g["coef"]
    p = [R.randn("adam.p." + k, s, 0, 0.3) for k, s in zip(names, shapes)]
    v = [torch.zeros(s) for s in shapes]
    cnt = [0] * 4
    for step in range(3):
        for i, k in enumerate(names):
            if step == 1 and k == "plain":
                continue
            gr = R.randn(f"adam.g{step}." + k, shapes[i], 0, 0.01 * (step + 1))
            cnt[i] += 1
            p[i], v[i] = O.lreq_adam_step(p[i], gr, v[i], cnt[i], 0.0015, coef=coef[i])
        for i, k in enumerate(names):
            close(p[i], g[f"s{step}:{k}"], rtol=1e-5, atol=1e-7)


def test_shape_tables_match_reference_state_dicts():
    from tests.helpers import s2_shapes, enc_shapes
    with open(os.path.join(ROOT, "tests", "golden", "s2_keys.json")) as f:
        s2 = json.load(f)
    for res in (1024, 256):
        mine = s2_shapes(res)
        assert list(mine.keys()) == list(s2[str(res)].keys())
        assert all(list(mine[k]) == s2[str(res)][k] for k in mine)
    assert len(s2["1024"]) == 165
    with open(os.path.join(ROOT, "tests", "golden", "enc_keys.json")) as f:
        ek = json.load(f)
    for tag, (sf, lc) in (("1024_16_9", (16, 9)), ("256_64_7", (64, 7))):
        mine = enc_shapes(sf, 512, lc)
        assert set(mine.keys()) == set(ek[tag].keys())
        assert all(list(mine[k]) == ek[tag][k] for k in mine)
    assert len(ek["1024_16_9"]) == 101 and len(ek["256_64_7"]) == 77


def test_c_oracle_is_clean_under_address_and_ub_sanitizers():
    """oracle/Makefile `san-check`: the C restatement compiled with -fsanitize=address,undefined (reports fatal) and driven
    over the parity shapes and the edge cases by oracle/san_driver.c."""
    import subprocess
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san-check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cases clean" in r.stdout


def test_c_oracle_modconv():
    """The plain-C restatement of the north-star kernel (oracle/modconv_oracle.c, the reference's
    fused per-sample-weight formulation) against the reference's own outputs."""
    import ctypes as C
    import subprocess
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(so)
    fp = C.POINTER(C.c_float)
    ptr = lambda t: t.contiguous().data_ptr()
    g = golden("s2_blocks.npz")
    from tests.helpers import modconv_shapes
    for ci in range(5):          # case 5 (512x512) is covered by the torch oracle; keep the C loop nests fast
        cin, cout, res, up, k = [int(v) for v in g[f"c{ci}_cfg"]]
        torgb = (k == 1)
        P = R.fill_s2(modconv_shapes(cin, cout, res, k, noise=not torgb, up=bool(up)), seed=100 + ci)
        rin = res // 2 if up else res
        x = R.randn(f"mc{ci}.x", (2, cin, rin, rin), 7)
        w = R.randn(f"mc{ci}.w", (2, 512), 7)
        style = torch.empty(2, cin)
        lib.orc_style(C.c_void_p(ptr(w)), C.c_void_p(ptr(P["style.weight"])), C.c_void_p(ptr(P["style.bias"])),
                      C.c_void_p(ptr(style)), 2, cin, 512)
        close(style, g[f"c{ci}_style"])
        y = torch.empty(2, cout, res, res)
        noise = P["noise"].reshape(-1) if not torgb else None
        lib.orc_modconv(C.c_void_p(ptr(x)), C.c_void_p(ptr(P["weight"])), C.c_void_p(ptr(style)), C.c_void_p(ptr(P["bias"])),
                        C.c_void_p(ptr(noise)) if noise is not None else None,
                        C.c_float(float(P["noise_strength"]) if not torgb else 0.0), C.c_void_p(ptr(y)),
                        2, cin, cout, res, k, int(up), int(not torgb), int(not torgb))
        close(y, g[f"c{ci}_y"])


def test_conv_ref_layer_oracle_pinned_on_reference_blocks():
    """oracle/conv_ref.py (the per-launch oracle of tests/test_fullsize_gpu.py) reproduces the outputs of the reference's own
    ModulateConvBlock (stride-1, up, toRGB) and its data gradients agree with autograd through the same lines."""
    import math
    from oracle import conv_ref as CR
    from tests.helpers import modconv_shapes
    g = golden("s2_blocks.npz")
    for ci in range(6):
        cin, cout, res, up, k = [int(v) for v in g[f"c{ci}_cfg"]]
        torgb = (k == 1)
        P = R.fill_s2(modconv_shapes(cin, cout, res, k, noise=not torgb, up=bool(up)), seed=100 + ci)
        rin = res // 2 if up else res
        x = R.randn(f"mc{ci}.x", (2, cin, rin, rin), 7)
        s = T(g[f"c{ci}_style"])
        wscale = 1.0 / math.sqrt(cin * k * k)
        wh = P["weight"] * wscale
        d = None if torgb else torch.rsqrt((s * s) @ (wh * wh).sum((2, 3)).t() + 1e-8)
        noise = None if torgb else P["noise"].reshape(1, res, res)
        ns = None if torgb else float(P["noise_strength"])
        kw = dict(gain=1.0, slope=1.0) if torgb else {}
        fn = CR.upconv_fir if up else CR.modconv
        y = fn(x, P["weight"], s, d, noise, ns, P["bias"], 1.0, wscale, **kw)
        close(y, g[f"c{ci}_y"])
    # adjoints used by the data-gradient launches: against autograd of the forward restatements
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 6, 6, generator=gen, requires_grad=True)
    w = torch.randn(12, 8, 3, 3, generator=gen)
    gy = torch.randn(2, 12, 6, 6, generator=gen)
    y = torch.nn.functional.conv2d(x, w * 0.3, padding=1)
    close(CR.conv_dgrad(gy, w, 0.3), torch.autograd.grad(y, x, gy)[0], rtol=1e-5)
    wv = (w * 0.3).clone().requires_grad_(True)
    y = torch.nn.functional.conv2d(x.detach(), wv, padding=1)
    close(CR.conv_wgrad(gy, x.detach(), 3), torch.autograd.grad(y, wv, gy)[0], rtol=1e-5)
    gu = torch.randn(2, 12, 12, 12, generator=gen)
    xu = x.detach().clone().requires_grad_(True)
    close(CR.up_dgrad(gu, w, 0.3, 6), torch.autograd.grad(CR.up_linear(xu, w, 0.3), xu, gu)[0], rtol=1e-5)


def test_conv_ref_folded_forms_equal_the_unfused_ones():
    """modconv_folded / enc_conv_folded (weight-side modulation, the reference's fused form :858-864) are the same functions
    as modconv / enc_conv when no storage rounding is applied."""
    from oracle import conv_ref as CR
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(2, 8, 9, 7, generator=gen)
    w = torch.randn(6, 8, 3, 3, generator=gen)
    s = 1 + 0.3 * torch.randn(2, 8, generator=gen)
    d = 0.5 + torch.rand(2, 6, generator=gen)
    nz = torch.randn(1, 9, 7, generator=gen)
    bias = torch.randn(6, generator=gen)
    close(CR.modconv_folded(x, w, s, d, nz, 0.3, bias, 1.0, 0.1), CR.modconv(x, w, s, d, nz, 0.3, bias, 1.0, 0.1), rtol=1e-5)
    sc = 0.5 + torch.rand(2, 8, generator=gen)
    sh = torch.randn(2, 8, generator=gen)
    nzb = torch.randn(2, 9, 7, generator=gen)
    nw = torch.randn(6, generator=gen)
    a = CR.enc_conv_folded(x, w, sc, sh, nzb, nw, bias)
    b = CR.enc_conv(x, w, sc, sh, nzb, nw, bias)
    for u, v in zip(a, b):
        close(u, v, rtol=2e-5)


def test_elem_ref_encoder_block_backward_equals_autograd_of_the_reference_restatement():
    """oracle/elem_ref.py's stage-wise backward (what tests/test_fullsize_elem_gpu.py holds the HIP kernels to), composed the way
    dge_amd/autograd_enc_bwd.py composes the kernels, against torch.autograd through oracle/ref_torch.enc_block - itself pinned
    on the reference's BEBlock outputs above (model/E/E.py:50-85)."""
    from oracle import elem_ref as ER, conv_ref as CR
    torch.manual_seed(3)
    C, C2, H = 8, 16, 12
    pre = "decode_block.0."
    P = {pre + "inver_mod1.weight": torch.randn(32, 2 * C) * 0.2, pre + "inver_mod1.bias": torch.randn(32) * 0.1,
         pre + "inver_mod2.weight": torch.randn(32, 2 * C) * 0.2, pre + "inver_mod2.bias": torch.randn(32) * 0.1,
         pre + "conv_1.weight": torch.randn(C, C, 3, 3) * 0.2, pre + "conv_2.weight": torch.randn(C2, C, 3, 3) * 0.2,
         pre + "conv_3.weight": torch.randn(C2, C, 1, 1) * 0.3, pre + "conv_3.bias": torch.randn(C2) * 0.1,
         pre + "noise_weight_1": torch.randn(1, C, 1, 1) * 0.3, pre + "bias_1": torch.randn(1, C, 1, 1) * 0.2,
         pre + "noise_weight_2": torch.randn(1, C2, 1, 1) * 0.3, pre + "bias_2": torch.randn(1, C2, 1, 1) * 0.2}
    P = {k: v.double().requires_grad_(True) for k, v in P.items()}
    x = torch.randn(1, C, H, H, dtype=torch.float64, requires_grad=True)
    n1, n2 = torch.randn(1, 1, H, H, dtype=torch.float64), torch.randn(1, 1, H, H, dtype=torch.float64)
    out, w1, w2 = O.enc_block(P, pre, x, n1, n2, last=False)
    g_out, g_w1, g_w2 = torch.randn_like(out), torch.randn_like(w1), torch.randn_like(w2)
    names = [pre + "bias_1", pre + "noise_weight_1", pre + "bias_2", pre + "noise_weight_2"]
    ref = torch.autograd.grad((out * g_out).sum() + (w1 * g_w1).sum() + (w2 * g_w2).sum(), [x] + [P[n] for n in names])
    # ---- the stage-wise composition (stored activations only, as the HIP path has them)
    with torch.no_grad():
        m1, v1 = O.enc_stats(x)
        x1 = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(O.inorm(x, m1, v1), P[pre + "conv_1.weight"], padding=1)
                                            + P[pre + "noise_weight_1"] * n1 + P[pre + "bias_1"], 0.2)
        m2, v2 = O.enc_stats(x1)
        a2 = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(O.inorm(x1, m2, v2), P[pre + "conv_2.weight"], padding=1)
                                            + P[pre + "noise_weight_2"] * n2 + P[pre + "bias_2"], 0.2)
    gp2, gb2, gn2 = ER.enc_act_pool_bwd(a2, g_out, n2[0, 0], 0.111 * 0.25)
    g_yn2 = CR.conv_dgrad(gp2, P[pre + "conv_2.weight"].detach())
    gms2 = (g_w2 @ P[pre + "inver_mod2.weight"].detach())[0]
    g_pre1, gb1, gn1 = ER.enc_in_bwd(x1, g_yn2, gms2[:C], gms2[C:], noise=n1[0, 0], act=True)
    g_yn1 = CR.conv_dgrad(g_pre1, P[pre + "conv_1.weight"].detach())
    extra = CR.conv_dgrad(g_out, P[pre + "conv_3.weight"].detach()) * 0.889
    gms1 = (g_w1 @ P[pre + "inver_mod1.weight"].detach())[0]
    g_x, _, _ = ER.enc_in_bwd(x.detach(), g_yn1, gms1[:C], gms1[C:], extra=extra, extra_scale=0.25)
    got = [g_x, gb1, gn1, gb2, gn2]
    for a, b, nm in zip(got, ref, ["x"] + names):
        assert torch.allclose(a.reshape(-1), b.reshape(-1), rtol=1e-9, atol=1e-10), nm


def test_elem_ref_generator_tails_equal_autograd_of_the_reference_restatement():
    """modconv tail / toRGB stages of oracle/elem_ref.py against autograd through oracle/ref_torch.s2_modconv (pinned on the
    reference's ModulateConvBlock outputs above; model/stylegan2_generator.py:855-922, :515-522)."""
    from oracle import elem_ref as ER, conv_ref as CR
    torch.manual_seed(4)
    Ci, Co, H = 8, 8, 10
    P = {"L.weight": torch.randn(Co, Ci, 3, 3), "L.bias": torch.randn(Co) * 0.3, "L.noise_strength": torch.tensor(0.4),
         "L.noise": torch.randn(1, 1, H, H), "L.style.weight": torch.randn(Ci, 512) * 0.5, "L.style.bias": torch.randn(Ci) * 0.2,
         "T.weight": torch.randn(3, Co, 1, 1), "T.bias": torch.randn(3) * 0.1, "T.style.weight": torch.randn(Co, 512) * 0.5,
         "T.style.bias": torch.randn(Co) * 0.2}
    x = torch.randn(1, Ci, H, H, requires_grad=True)
    wl = torch.randn(1, 512)
    y, s = O.s2_modconv(P, "L", x, wl)
    prev = torch.randn(1, 3, H // 2, H // 2)
    rgb, srgb = O.s2_modconv(P, "T", y, wl, demodulate=False, add_noise=False, act="linear")
    img = rgb + O.s2_upsample_skip(prev)
    gimg = torch.randn_like(img)
    (gx_ref,) = torch.autograd.grad((img * gimg).sum(), x)
    # stage-wise
    wscale_t = 1.0 / math.sqrt(Co)
    img2 = ER.torgb(y.detach(), P["T.weight"].reshape(3, Co), srgb[0].detach(), P["T.bias"], wscale_t, prev)
    assert torch.allclose(img2, img.detach(), rtol=1e-5, atol=1e-5)
    gy, gs = ER.torgb_bwd(y.detach(), P["T.weight"].reshape(3, Co), srgb[0].detach(), wscale_t, gimg)
    wscale = 1.0 / math.sqrt(9 * Ci)
    Wm = P["L.weight"] * wscale
    d = torch.rsqrt(((Wm[None] * s.detach()[:, None, :, None, None]) ** 2).sum(dim=(2, 3, 4)) + 1e-8)[0]
    g_yraw, R = ER.modconv_tail_bwd(y.detach(), gy, d, P["L.noise"][0, 0], math.sqrt(2.0))
    gx = CR.conv_dgrad(g_yraw.float(), Wm) * s.detach()[:, :, None, None]
    assert torch.allclose(gx, gx_ref, rtol=2e-4, atol=2e-5)
    # the noise-strength gradient is R[:,1] summed over channels; bias gradient R[:,2] (bscale 1)
    ns = P["L.noise_strength"].clone().requires_grad_(True)
    b = P["L.bias"].clone().requires_grad_(True)
    P2 = dict(P); P2["L.noise_strength"] = ns; P2["L.bias"] = b
    y2, _ = O.s2_modconv(P2, "L", x.detach(), wl)
    rgb2, _ = O.s2_modconv(P2, "T", y2, wl, demodulate=False, add_noise=False, act="linear")
    gns, gb = torch.autograd.grad((rgb2 * gimg).sum(), (ns, b))
    assert abs(float(R[:, 1].sum()) - float(gns)) < 2e-4 * abs(float(gns)) + 1e-5
    assert torch.allclose(R[:, 2].float(), gb, rtol=2e-4, atol=2e-5)


def test_step_ref_reproduces_the_reference_run():
    """oracle/step_ref.py - the composition the full-size step test and bench.py's cpu_baseline trust - against the reference's OWN
    two-iteration E_align_s2 run (tests/golden/step_s2.npz, made by tools/gen_golden.py:gen_step from the reference's modules):
    train-mode generator pass (w_avg EMA, style mixing with the reference's np.random draw order, :177-191), encoder, synthesis,
    the three space_loss terms, both backward phases with LREQAdam in between (quirk Q3), all in one call per iteration."""
    import numpy as np
    from oracle import step_ref
    from oracle import lpips_ref as LR
    from tests.golden import recipe as R
    from tests.helpers import s2_shapes, enc_shapes
    g = golden("step_s2.npz")
    PG = R.fill_s2(s2_shapes(64, fmaps_base=2048, fmaps_max=128), seed=11)
    PE = {k: v.clone().requires_grad_(True) for k, v in R.fill_encoder(enc_shapes(16, 64, 5), seed=31).items()}
    PL = LR.seeded_params(0)
    # lr-equalisation coefficients as the reference tags them (model/utils/lreq.py:60-62,118-120): taken from the module surface,
    # whose tags tests/test_step_gpu.py checks against the reference's own optimiser run
    from dge_amd.encoder import BE
    from tests.helpers import s2_shapes, enc_shapes
    coefs = {n: p.lr_equalization_coef for n, p in BE(startf=16, maxf=64, layer_count=5).named_parameters() if hasattr(p, "lr_equalization_coef")}
    state = {"_coef": coefs}
    new_z = R.randn("step.new_z", (2, 512), 1)
    nl = O.s2_num_layers(PG)
    for it in range(2):
        np.random.seed(it)
        u = np.random.uniform()
        cutoff = np.random.randint(1, nl) if u < 0.9 else 0
        z = R.randn(f"step.z{it}", (2, 512), 1)
        noises = [R.randn(f"step.it{it}.noise{i}", s, 1) for i, s in enumerate(O.enc_noise_shapes(5, 2, 64))]
        rec = {}
        r = step_ref.e_align_step(PG, PE, PL, z, noises, lr=0.0015, state=state, record=rec, train=dict(new_z=new_z, u=u, cutoff=cutoff))
        close(r["wp"], g[f"it{it}_w1"], rtol=1e-5)
        close(r["imgs1"], g[f"it{it}_imgs1"], rtol=2e-5)
        close(r["w2"], g[f"it{it}_w2"], rtol=2e-4)
        close(r["imgs2"], g[f"it{it}_imgs2"], rtol=2e-4)
        ref_l = g[f"it{it}_losses"]
        got = [r["loss_tsa"], *r["loss_parts"], r["loss_w"]]
        for a, b in zip(got, ref_l):
            assert abs(a - b) < 2e-4 * abs(b), (it, got, ref_l)
        close(PG["truncation.w_avg"], g[f"it{it}_w_avg"], rtol=1e-6)
        for key in g.files:
            if key.startswith(f"it{it}_grad2:"):
                close(rec["grad2"][key.split(":", 1)[1]], g[key], rtol=2e-3)
            if key.startswith(f"it{it}_after_phase2:"):
                close(PE[key.split(":", 1)[1]].detach(), g[key], rtol=2e-5)


def test_folded_up_layer_oracle_equals_the_reference_form():
    """oracle/conv_ref.py:upfold_weights / up_folded / dgrad_folded (the four-phase 3x3 form whose data gradient csrc/conv_pp.hip
    computes) against up_linear, the restatement of the reference's transposed conv + FIR (stylegan2_generator.py:879-896) that the
    golden blocks pin above."""
    from oracle import conv_ref as CR
    torch.manual_seed(5)
    w, x = torch.randn(6, 5, 3, 3), torch.randn(2, 5, 7, 9)
    a = CR.up_linear(x, w, 0.3)
    b = CR.up_folded(x.double(), CR.upfold_weights(w.double(), 0.3))
    assert (a - b).abs().max().item() < 2e-6 * a.abs().max().item()
    g = torch.randn(1, 6, 14, 18)
    xx = torch.zeros(1, 5, 7, 9, requires_grad=True)
    want = torch.autograd.grad(CR.up_linear(xx, w, 0.3), xx, g)[0]
    got = CR.dgrad_folded(g, w, 0.3, torch.ones(6), up=True)
    assert (got - want).abs().max().item() < 2e-6 * want.abs().max().item()
    d = 0.5 + torch.rand(6)
    want1 = CR.conv_dgrad(g * d[None, :, None, None], w, 0.3)
    got1 = CR.dgrad_folded(g, w, 0.3, d)
    assert (got1 - want1).abs().max().item() < 2e-6 * want1.abs().max().item()
