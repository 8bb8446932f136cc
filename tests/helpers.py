"""Shape tables of the reference's state_dicts, derived from the published formulas
(and checked against tests/golden/*_keys.json, which were dumped from the reference)."""
import math


def modconv_shapes(cin, cout, res, k=3, noise=True, up=False, wdim=512):
    s = {"weight": [cout, cin, k, k], "bias": [cout]}
    if noise:
        s["noise_strength"] = []
        s["noise"] = [1, 1, res, res]
    if up:
        s["filter.kernel"] = [1, 1, 4, 4]
    s["style.weight"] = [cin, wdim]
    s["style.bias"] = [cin]
    return s


def s2_shapes(resolution, fmaps_base=32 << 10, fmaps_max=512, wdim=512):
    nf = lambda r: min(fmaps_base // r, fmaps_max)
    s = {}
    for i in range(8):
        s[f"mapping.dense{i}.weight"] = [512, 512]
        s[f"mapping.dense{i}.bias"] = [512]
    s["truncation.w_avg"] = [wdim]
    s["synthesis.early_layer.const"] = [1, nf(4), 4, 4]
    rl = int(math.log2(resolution))
    for r2 in range(2, rl + 1):
        res = 1 << r2
        b = r2 - 2
        if res != 4:
            for k, v in modconv_shapes(nf(res // 2), nf(res), res, up=True).items():
                s[f"synthesis.layer{2 * b - 1}.{k}"] = v
        for k, v in modconv_shapes(nf(res), nf(res), res).items():
            s[f"synthesis.layer{2 * b}.{k}"] = v
        for k, v in modconv_shapes(nf(res), 3, res, k=1, noise=False).items():
            s[f"synthesis.output{b}.{k}"] = v
    s["synthesis.upsample.kernel"] = [1, 1, 4, 4]
    return s


def enc_shapes(startf, maxf, layer_count, latent=512, channels=3):
    s = {"FromRGB.from_rgb.weight": [startf, channels, 1, 1], "FromRGB.from_rgb.bias": [startf]}
    cin, cout = startf, startf * 2
    for j in range(layer_count):
        p = f"decode_block.{j}."
        last = (j + 1 == layer_count)
        s[p + "noise_weight_1"] = [1, cin, 1, 1]
        s[p + "bias_1"] = [1, cin, 1, 1]
        s[p + "noise_weight_2"] = [1, cout, 1, 1]
        s[p + "bias_2"] = [1, cout, 1, 1]
        s[p + "inver_mod1.weight"] = [latent, 2 * cin]
        s[p + "inver_mod1.bias"] = [latent]
        s[p + "conv_1.weight"] = [cin, cin, 3, 3]
        s[p + "inver_mod2.weight"] = [latent, 2 * cin]
        s[p + "inver_mod2.bias"] = [latent]
        if not last:
            s[p + "conv_2.weight"] = [cout, cin, 3, 3]
        if cin != cout:
            s[p + "conv_3.weight"] = [cout, cin, 1, 1]
            s[p + "conv_3.bias"] = [cout]
        cin, cout = min(maxf, cin * 2), min(maxf, cout * 2)
    return s
