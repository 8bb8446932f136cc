"""StyleGAN2 synthesis pipeline over the HIP ops (forward; the data-gradient path w.r.t. wp
is added by SynthesisFunction)."""
import torch

from . import ops
from .stylegan2_generator import _dt


def synthesis_forward(mod, wp, randomize_noise=False):
    """SynthesisModule.forward, reference model/stylegan2_generator.py:492-539: conv layer i is
    driven by wp[:, i]; the toRGB of block k by wp[:, 2k+1] (:511-517)."""
    dt = _dt(mod.compute_dtype)
    B = wp.shape[0]
    wp = wp.float().contiguous()
    results = {"wp": wp}
    x = ops.nchw_to_nhwc(mod.early_layer.const.detach(), B, dt)
    image = None
    for i in range(mod.num_layers - 1):
        x, style = getattr(mod, f"layer{i}")(x, wp[:, i], randomize_noise)
        results[f"style{i:02d}"] = style
        if i % 2 == 0:
            image, style = getattr(mod, f"output{i // 2}")(x, wp[:, i + 1], prev_image=image)
            results[f"output_style{i // 2}"] = style
    results["image"] = image
    return results
