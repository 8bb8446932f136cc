"""MI355X-native E_align hot path of disanda/Deep-GAN-Encoders.

Module surface mirrors the reference (class names, ctor/forward signatures, state_dict keys);
all device math runs in libdge_hip.so (hand-written gfx950 HIP kernels) through the C ABI in
include/dge_hip.h.  There is no CPU or stock-PyTorch fallback: using a module without the
library (or without a GPU) raises.
"""
from . import _lib  # noqa: F401
from .stylegan2_generator import StyleGAN2Generator  # noqa: F401
