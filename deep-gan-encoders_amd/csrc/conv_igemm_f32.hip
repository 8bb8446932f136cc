// Implicit-GEMM convolution, exact-f32 instantiations (v_mfma_f32_32x32x2_f32: the parity path); kernel in conv_igemm_impl.h.
#include "conv_igemm_impl.h"

int dge_conv_igemm_f32(const ConvParams& p, int ksize, hipStream_t s) {
    return ksize == 3 ? launch_t<float, 3>(p, s) : launch_t<float, 1>(p, s);
}
