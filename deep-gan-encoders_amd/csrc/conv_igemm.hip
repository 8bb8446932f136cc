// Implicit-GEMM 3x3 / 1x1 convolution for gfx950 - bf16 instantiations and the dispatch of the conv family
// (kernel and launch rules: conv_igemm_impl.h; f32 instantiations: conv_igemm_f32.hip).
#include "conv_igemm_impl.h"

int dge_conv_igemm_f32(const ConvParams& p, int ksize, hipStream_t s);

// N-tile width the packer pads to (the launcher uses this or a divisor of it).
extern "C" int dge_conv_ntile(int ntot) { return ntot >= 128 ? 128 : (ntot > 32 ? 64 : 32); }

int dge_conv_launch(const ConvParams& p, int dtype, int ksize, hipStream_t s) {
    if (p.w_frag) {                  // fragment-ordered weights: the low-resolution kernel (conv_small.hip) is the only reader
        DGE_CHECK(dge_conv_small_shape_ok(p.H, p.W, p.Cin, p.Ntot, ksize, p.in_s2d, p.in_up2, dtype),
                  "conv: w_layout = 1 (DGE_PACK_FRAG) but the launch %dx%d Cin=%d N=%d is not one dge_conv_small_supported() accepts",
                  p.H, p.W, p.Cin, p.Ntot);
        return dge_conv_small_launch(p, s);
    }
    if (dge_conv_stream_eligible(p, dtype, ksize)) return dge_conv_stream_launch(p, s);     // HBM-bound layers: conv_stream.hip
    if (dge_conv_pw_eligible(p, dtype, ksize)) return dge_conv_pw_launch(p, s);             // narrow 1x1 layers: conv_pw.hip
    const int esize = dtype == DGE_BF16 ? 2 : 4;
    DGE_CHECK(ksize == 1 || ksize == 3, "conv: ksize %d unsupported", ksize);
    DGE_CHECK(p.Cin % (32 / esize) == 0, "conv: Cin=%d must be a multiple of %d", p.Cin, 32 / esize);
    DGE_CHECK(p.Cout % (16 / esize) == 0, "conv: Cout=%d must be a multiple of %d", p.Cout, 16 / esize);
    DGE_CHECK(p.Ntot % dge_conv_ntile(p.Ntot) == 0, "conv: packed N=%d not padded to the N tile", p.Ntot);
    DGE_CHECK(!p.up || p.Cout % 16 == 0, "conv: up mode needs Cout %% 16 == 0 (got %d)", p.Cout);
    if (dtype == DGE_BF16) return ksize == 3 ? launch_t<bf16_t, 3>(p, s) : launch_t<bf16_t, 1>(p, s);
    return dge_conv_igemm_f32(p, ksize, s);            // conv_igemm_f32.hip (the exact-f32 parity path: its own translation unit)
}
