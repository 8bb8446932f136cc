"""ctypes binding of libdge_hip.so (C ABI: include/dge_hip.h).  Fails loudly when missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DGE_LIB_PATH", os.path.join(_HERE, "libdge_hip.so"))   # override: A/B builds while tuning


class DgeError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w_packed", C.c_void_p), ("y", C.c_void_p), ("addend", C.c_void_p), ("dot_src", C.c_void_p),
        ("in_scale", C.c_void_p), ("in_shift", C.c_void_p), ("out_scale", C.c_void_p),
        ("bias", C.c_void_p), ("noise", C.c_void_p), ("noise_w", C.c_void_p), ("stats", C.c_void_p),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
        ("ksize", C.c_int), ("up", C.c_int), ("in_s2d", C.c_int), ("noise_batch", C.c_int), ("noise_w_per_channel", C.c_int),
        ("act", C.c_int), ("bias_scale", C.c_float), ("gain", C.c_float), ("add_scale", C.c_float),
        ("dtype", C.c_int), ("in_up2", C.c_int), ("in_relu", C.c_int), ("stats_slots", C.c_int), ("w_layout", C.c_int),
        ("prep", C.c_int), ("prep_gain", C.c_float), ("prep_noise", C.c_void_p), ("prep_ns", C.c_void_p), ("prep_noise_batch", C.c_int),
        ("prep_stats", C.c_void_p), ("mask_relu", C.c_int), ("in_t2d", C.c_int),
        ("rgb_w", C.c_void_p), ("rgb_style", C.c_void_p), ("rgb_bias", C.c_void_p), ("rgb_out", C.c_void_p), ("rgb_wscale", C.c_float),
        ("rgb_skip_y", C.c_int), ("pool_out", C.c_int), ("pool_mask", C.c_void_p),
        ("prefetch_w", C.c_void_p), ("prefetch_ntot", C.c_int), ("prefetch_cin", C.c_int), ("in_bwd_coef", C.c_void_p),
        ("in_bwd_extra", C.c_void_p), ("in_bwd_extra_scale", C.c_float), ("fr_img4", C.c_void_p), ("fr_out", C.c_void_p),
    ]


class ConvPPDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w_pp", C.c_void_p), ("y", C.c_void_p), ("w_bstride", C.c_longlong),
        ("out_scale", C.c_void_p), ("bias", C.c_void_p), ("noise", C.c_void_p), ("noise_w", C.c_void_p),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
        ("noise_batch", C.c_int), ("noise_w_per_channel", C.c_int), ("act", C.c_int), ("bias_scale", C.c_float), ("gain", C.c_float),
        ("dgrad", C.c_int), ("in_s2d", C.c_int), ("stats_slots", C.c_int), ("prep", C.c_int), ("prep_noise_batch", C.c_int), ("mask_relu", C.c_int),
        ("in_t2d", C.c_int), ("add_scale", C.c_float), ("prep_gain", C.c_float),
        ("dot_src", C.c_void_p), ("addend", C.c_void_p), ("stats", C.c_void_p), ("prep_stats", C.c_void_p), ("prep_noise", C.c_void_p), ("prep_ns", C.c_void_p),
    ]


class DenseLayer(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("I", C.c_int), ("O", C.c_int), ("wscale", C.c_float), ("bscale", C.c_float),
                ("add", C.c_float), ("act", C.c_int), ("gain", C.c_float)]


class S2GradEntry(C.Structure):
    _fields_ = [("P", C.c_void_p), ("st", C.c_void_p), ("d", C.c_void_p), ("s", C.c_void_p), ("bias", C.c_void_p), ("wsq", C.c_void_p),
                ("gs", C.c_void_p), ("wstyle", C.c_void_p),
                ("nslot_p", C.c_int), ("nslot_s", C.c_int), ("in_c", C.c_int), ("out_c", C.c_int), ("row", C.c_int), ("bscale", C.c_float)]


class SumPlanarEntry(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("out", C.c_void_p), ("nslot", C.c_int), ("C", C.c_int), ("NS", C.c_int), ("pad", C.c_int)]


_P, _I, _F = C.c_void_p, C.c_int, C.c_float
# name -> argtypes; every symbol declared in include/dge_hip.h must be listed here
SIGNATURES = {
    "dge_conv2d": [C.POINTER(ConvDesc), _P],
    "dge_fir_t2d": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dge_s2_style_grads": [C.POINTER(S2GradEntry), _I, _P, _I, _I, _I, _F, _P],
    "dge_conv_small_supported": [_I, _I, _I, _I, _I, _I, _I, _I],
    "dge_torgb_bwd_prep": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P],
    "dge_demod_bwd_prep": [_P, _I, _P, _P, _P, _I, _I, _F, _P],
    "dge_sum_slots": [_P, _P, _I, _I, _I, _P],
    "dge_sum_slots_planar": [_P, _P, _I, _I, _I, _P],
    "dge_set_deterministic": [_I],
    "dge_get_deterministic": [],
    "dge_packed_n": [_I],
    "dge_pack_conv_weight": [_P, _P, _I, _I, _I, _I, _I, _F, _P],
    "dge_pack_conv_weights_multi": [_P, _P, _I, _I, _P],
    "dge_pack_desc_bytes": [],
    "dge_weight_sumsq": [_P, _P, _I, _I, _I, _F, _P],
    "dge_linear": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _I, _F, _I, _P],
    "dge_pixelnorm": [_P, _P, _I, _I, _F, _P],
    "dge_truncation": [_P, _P, _P, _I, _I, _I, _F, _I, _I, _P],
    "dge_torgb": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "dge_conv_rgb_supported": [_I, _I, _I, _I, _I, _I, _I],
    "dge_conv_in_bwd_supported": [_I, _I, _I, _I, _I, _I, _I],
    "dge_conv_in_bwd_fromrgb_supported": [_I, _I, _I, _I, _I, _I, _I],
    "dge_conv_in_bwd_x_supported": [_I, _I, _I, _I, _I, _I, _I],
    "dge_conv_pool_supported": [_I, _I, _I, _I, _I, _I, _I],
    "dge_rgb_upsample_add": [_P, _P, _I, _I, _I, _P],
    "dge_nchw_to_nhwc": [_P, _P, _I, _I, _I, _I, _I, _P],
    "dge_nhwc_to_nchw": [_P, _P, _I, _I, _I, _I, _P],
    "dge_fromrgb": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dge_fromrgb2": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dge_stats_finalize": [_P, _P, _P, _P, _I, _I, _I, _F, _P],
    "dge_stats_finalize_slots": [_P, _I, _P, _P, _P, _I, _I, _I, _F, _P],
    "dge_blend": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _I, _P],
    "dge_loss_reduce": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "dge_crop_pool": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "dge_ssim_fwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "dge_ssim_box7": [_P, _P, _P, _I, _I, _I, _F, _F, _F, _P],
    "dge_ssim_bwd": [_P, _P, _P, _P, _I, _I, _I, _F, _I, _P],
    "dge_space_loss_finalize": [_P, _P, _P, _P, _F, _F, _I, _P],
    "dge_space_loss_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _I, _P],
    "dge_axpy_scalar": [_P, _P, _P, C.c_long, _F, _I, _P],
    "dge_lreq_adam_multi": [_I, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P],
    "dge_modconv_bwd_prep": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "dge_demod_bwd": [_P, _P, _P, _P, _P, _I, _I, _F, _P],
    "dge_linear_t": [_P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "dge_torgb_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P],
    "dge_up2_bwd": [_P, _P, _I, _I, _I, _P],
    "dge_conv_wgrad": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dge_act_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _I, _P],
    "dge_in_bwd_coef": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "dge_in_bwd_coef_slots": [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "dge_in_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "dge_chan_sum": [_P, _P, _I, _I, _I, _F, _I, _P],
    "dge_fromrgb_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dge_head_entry_size": [],
    "dge_heads_bwd": [_P, _I, _I, _P, _I, _P, _P, _P, _P, _I, _I, _P],
    "dge_dense_wgrad": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "dge_lpips_prep": [_P, _P, _I, _I, _I, _P, _P, _I, _P],
    "dge_lpips_prep_bwd": [_P, _P, _I, _I, _I, _P, _F, _I, _I, _P],
    "dge_maxpool2": [_P, _P, _I, _I, _I, _I, _I, _P],
    "dge_maxpool2_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dge_lpips_head": [_P, _P, _P, _P, _I, _I, _I, _F, _I, _P],
    "dge_mean": [_P, _P, _I, _P],
    "dge_blur_noise_act": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "dge_affine_compose": [_P, _P, _P, _P, _P, _I, _I, _P],
    "dge_lerp_layers": [_P, _P, _I, _P, _P, _I, _I, _I, _P],
    "dge_linear_rows": [_P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P],
    "dge_demod_rows": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "dge_fromrgb_dgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dge_nearest_up2": [_P, _P, _I, _I, _I, _I, _F, _I, _P],
    "dge_sg1_in_bwd_coef": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "dge_dot_stats": [_P, _P, _P, _I, _I, _I, _I, _P],
    "dge_nearest_up2_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dge_pixelnorm_nhwc": [_P, _P, C.c_long, _I, _F, _I, _P],
    "dge_pixelnorm_nhwc_bwd": [_P, _P, _P, C.c_long, _I, _F, _I, _P],
    "dge_affine_relu_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dge_slice_up_bwd": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dge_softmax_rows": [_P, C.c_long, _I, _I, _P],
    "dge_softmax_rows_bwd": [_P, _P, C.c_long, _I, _I, _P],
    "dge_rgb_tanh_bwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "dge_sn_group": [_P, _I, _I, _I, _P, _F, _I, _P],
    "dge_sn_entry_size": [],
    "dge_cbn_sn_wgrad_entry_size": [],
    "dge_cbn_sn_wgrad_group": [_P, _I, C.c_longlong, _I, _P, _I, _I, _P, _P],
    "dge_upconv_supported": [_I, _I, _I],
    "dge_pack_upconv_weight": [_P, _P, _I, _I, _F, _I, _P],
    "dge_upconv_fir": [_P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _F, _F, _I, _I, _I, _I, _I, _I, _I, _P],
    "dge_cbn_affine": [_P, _P, _I, _P, _P, _F, _P, _P, _I, _I, _P],
    "dge_slice_up": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dge_attention": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dge_rgb_tanh": [_P, _P, _I, _I, _I, _I, _P],
    "dge_guided_relu_bwd": [_P, _P, _P, C.c_long, _I, _I, _P],
    "dge_maxpool2_relu_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dge_maxpool2_bwd_relu": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dge_adaptive_pool7": [_P, _P, _I, _I, _I, _I, _I, _P],
    "dge_adaptive_pool7_bwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "dge_class_target": [_P, _P, _P, _P, _I, _I, _P],
    "dge_gather_row": [_P, _P, _P, _I, _I, _F, _P],
    "dge_campp_map": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dge_cam_resize": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dge_mask2cam_blocks": [_I],
    "dge_mask2cam": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "dge_randn": [_P, _I, _P, _P, _P, _P, C.c_ulonglong, _P, _P],
    "dge_version": [],
    "dge_env_reload": [],
    "dge_blend_pool_mask": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _I, _P],
    "dge_act_bwd_mask": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _I, _P],
    "dge_loss_reduce3": [_P, _P, _P, _I, _I, _I, _I, _P, _I, _P],
    "dge_crop_pool_multi": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dge_space_loss_bwd3": [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P],
    "dge_in_bwd_fromrgb": [_P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "dge_in_bwd_fused": [_P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "dge_sum_slots_planar_multi": [C.POINTER(SumPlanarEntry), _I, _P],
    "dge_heads_fwd": [_P, _I, _P, _P, _I, _I, _I, _P],
    "dge_conv_pp_supported": [_I, _I, _I, _I, _I, _I],
    "dge_pack_conv_pp": [_P, _P, _I, _I, _F, _P, _P, _F, _I, _I, _P],
    "dge_dense_chain": [_P, _I, C.POINTER(DenseLayer), _I, _P, _I, _I, _I, _F, _P],
    "dge_conv_wgrad_dots": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dge_pack_conv_pp_rows": [_P, _I, _P, _I, _I, _P, _I, _P, _F, _I, _I, _P],
    "dge_conv_pp": [C.POINTER(ConvPPDesc), _P],
    "dge_up_pp_supported": [_I, _I, _I, _I, _I, _I],
    "dge_pack_up_pp": [_P, _P, _I, _I, _P, _P, _F, _I, _P],
    "dge_up_pp": [_P, _P, C.c_longlong, _P, _P, _I, _P, _P, _F, _F, _I, _I, _I, _I, _I, _I, _P],
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DgeError(f"{LIB_PATH} is missing: build it with deep-gan-encoders_amd/csrc/build.sh "
                           "(__graft_entry__.build()). There is no fallback path.")
        _lib = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
        for name in ("dge_last_error", "dge_last_kernel"):
            getattr(_lib, name).restype = C.c_char_p
            getattr(_lib, name).argtypes = []
        # DGE_DETERMINISTIC=1: deterministic mode from the first launch on (same as ops.set_deterministic(True); needs a device)
        if os.environ.get("DGE_DETERMINISTIC") == "1":
            import torch
            if torch.cuda.is_available() and _lib.dge_set_deterministic(1) != 0:
                raise DgeError("DGE_DETERMINISTIC=1: " + _lib.dge_last_error().decode())
    return _lib


def last_kernel():
    """Name of the kernel instantiation the last conv-family call of this thread selected (include/dge_hip.h)."""
    return lib().dge_last_kernel().decode()


def check(rc, what=""):
    if rc != 0:
        raise DgeError(f"{what} failed ({rc}): {lib().dge_last_error().decode()}")
