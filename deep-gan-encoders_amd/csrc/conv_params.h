// Kernel-side description of one implicit-GEMM convolution launch (see conv_igemm.hip).
#pragma once
#include <hip/hip_runtime.h>

struct ConvParams {
    const void* x;            // [B,H,W,Cin]  T
    const void* w;            // packed [KS*KS][Ntot][Cin] T  (dge_pack_conv_weight)
    void* y;                  // [B,OH,OW,Cout] T   (OH = 2H in up mode)
    const void* addend;       // optional [B,OH,OW,Cout] T, y += add_scale*addend
    const void* dot_src;      // optional [B,OH,OW,Cout] T: stats = (sum acc*dot_src, sum acc) of the raw accumulator
    const float* in_scale;    // optional [B,Cin]
    const float* in_shift;    // optional [B,Cin]
    const float* out_scale;   // optional [B,Cout]
    const float* bias;        // optional [Cout]
    const float* noise;       // optional [nB,OH,OW]
    const float* noise_w;     // [Cout] (stride 1) or scalar (stride 0)
    float* stats;             // optional [slots][B,Cout,2] (sum, sum of squares), atomically accumulated
    int stats_slots;          // workgroups spread their atomics over this many copies (contention)
    int B, H, W, Cin, Cout;
    int Ntot;                 // packed N (padded to the N tile); = 4*Cout (+pad) in up mode
    int Ntot_valid;           // unpadded N
    int up;                   // 1: depth-to-space x2 store of the 4 folded phases
    int in_s2d;               // 1: x is [B,2H,2W,Cin/4], read space-to-depth (adjoint of `up`)
    int in_relu;              // 1: ReLU after the prologue affine (BigGAN: BN -> ReLU -> conv)
    int in_up2;               // 1: x is [B,H/2,W/2,Cin], read through a nearest-neighbour x2 upsample
    int noise_bstride;        // 0 (shared noise) or OH*OW
    int noise_w_stride;       // 0 or 1
    int act;
    float bias_scale, gain, add_scale;
    int tiles_x, tiles_y;     // filled by the launcher
    int dbg;                  // ablation switches for tuning (DGE_CONV_DBG), 0 in production
    int w_frag;               // 1: `w` is in MFMA-fragment order (DGE_PACK_FRAG): the launch must go to conv_small.hip
    // Fused backward of the PRODUCING layer's tail (data gradients of the StyleGAN2 synthesis chain, needs dot_src): dot_src is the
    // stored activation x = lrelu(z)*prep_gain of the layer below, z = yraw*d + noise*ns + bias (stylegan2_generator.py:908-921).
    // The conv result g = acc*out_scale + addend is the gradient w.r.t. x; with prep the kernel stores g_z = g * prep_gain * lrelu'(x)
    // instead and adds per (b, c): prep_stats[.., 0] += sum g_z * (z - ns*noise), prep_stats[.., 1] += sum g_z  (the two sums the
    // demodulation / bias gradient of that layer needs; dge_modconv_bwd_prep computes them in a pass of its own).
    int prep;
    float prep_gain;
    const float* prep_noise;  // [nB,OH,OW] noise plane of the layer below, or null
    const float* prep_ns;     // its noise strength (device scalar), or null
    int prep_noise_bstride;   // 0 (shared plane) or OH*OW
    float* prep_stats;        // [slots][B,Cout,2], pre-zeroed (same slot count as `stats`)
    int mask_relu;            // 1: result *= [dot_src > 0] (ReLU backward of the layer below), no dot statistics
    int in_t2d;               // 1: x is [B,H+1,W+1,Cin] (dge_fir_t2d); only the taps (dy,dx) in {1,2}^2 are computed
    // toRGB of the result fused into the epilogue (conv_stream only; stylegan2_generator.py:515-522, :465-474): rgb_out[b][k][y][x] =
    // rgb_bias[k] + sum_c rgb_w[k][c] * rgb_wscale * rgb_style[b][c] * y[b][y][x][c]  (the bf16-stored y, f32 weights as hi + lo bf16)
    const float* rgb_w;       // [3][Cout] or null
    const float* rgb_style;   // [B][Cout]
    const float* rgb_bias;    // [3]
    float* rgb_out;           // [B][3][H][W] f32
    float rgb_wscale;
    int rgb_skip_y;           // 1: y is not stored (its only reader was the toRGB)
    // 2x2 average pool of the result taken in the epilogue (conv_stream only; model/E/E.py:75-76): y is [B,H/2,W/2,Cout]; pool_mask
    // (optional) receives the signs of the full-resolution values, [B, H/2*W/2, Cout/8] words (dge_blend_pool_mask's layout)
    int pool_out;
    unsigned* pool_mask;
    // conv_small.hip only: fragment-ordered weights of the NEXT low-resolution launch of the stream (or null).  Every workgroup
    // touches its share of the slice its XCD's workgroups will read, so that launch finds the 590 KB per N tile in L2 instead of
    // pulling it from HBM behind a cold miss (measured 22-24 us cold against 13-17 us hot per launch at 4^2 .. 16^2).
    const void* pf_w;
    int pf_ntot, pf_cin;
    // conv_stream.hip only (FL_DOT_IN): coefficients [B][Cout][3] = (A, Bc, Cc) of the instance-norm backward of the layer's input
    // x = dot_src (dge_in_bwd_coef): the launch stores g_pre = (A*acc + Bc*x + Cc) * lrelu'(x) and adds (sum g_pre, sum g_pre*noise)
    // to prep_stats (noise = prep_noise: the plane of the layer that produced x)
    const float* in_coef;
    // FL_DOT_FR (with in_coef): g_x = A*acc + Bc*x + Cc + in_extra_scale * in_extra[parent pixel] is not stored; fr_out [slots][B][Cout][4]
    // += sum g_x*lrelu'(x) * (fr_img4[b][p][0..3]);  in_extra [B][H/2][W/2][Cout] bf16 or null, fr_img4 [B][H][W][4] f32
    const void* in_extra; float in_extra_scale; const float* fr_img4; float* fr_out;
};

int dge_conv_launch(const ConvParams& p, int dtype, int ksize, hipStream_t s);
extern "C" int dge_conv_ntile(int ntot);

// conv_stream.hip: the streaming kernel for the HBM-bound small-channel 3x3 layers
bool dge_conv_stream_eligible(const ConvParams& p, int dtype, int ksize);
int dge_conv_stream_launch(const ConvParams& p, hipStream_t s);
bool dge_conv_rgb_ok(const ConvParams& p, int dtype, int ksize);
bool dge_conv_pool_ok(const ConvParams& p, int dtype, int ksize);
// conv_small.hip: the low-resolution 3x3 layers (whole-Cin halo tile resident in LDS, weights streamed straight into registers)
bool dge_conv_small_shape_ok(int H, int W, int cin, int ntot, int ksize, int in_s2d, int in_up2, int dtype);
int dge_conv_small_launch(const ConvParams& p, hipStream_t s);
// conv_pw.hip: 1x1 convolution of the narrow high-resolution layers (encoder skip branch conv_3 and its data gradient), no LDS
bool dge_conv_pw_eligible(const ConvParams& p, int dtype, int ksize);
int dge_conv_pw_launch(const ConvParams& p, hipStream_t s);
