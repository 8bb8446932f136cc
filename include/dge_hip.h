/* dge_hip.h - C ABI of libdge_hip.so: the MI355X (gfx950) kernels behind the E_align hot path
 * of disanda/Deep-GAN-Encoders.
 *
 * The reference has no FFI: its hot path sits behind torch.nn.Module objects that call
 * torch.nn.functional ops.  Each entry point below replaces the op sequence of
 * the cited reference lines (paths relative to the reference root).  The Python modules in
 * deep-gan-encoders_amd/ keep the reference's class names / forward signatures / state_dict keys
 * and call these functions through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless named host_*.  No allocation happens inside;
 *     outputs and scratch are caller provided.  No global mutable state except the last-error
 *     string (thread local).  Every call is asynchronous on `stream`.
 *   - Activations: NHWC, element type selected by `dtype` (DGE_F32 = 0 exact-f32 parity path,
 *     DGE_BF16 = 1 bf16 storage with f32 accumulation).  Images and latents: NCHW / row-major f32
 *     exactly as the reference's tensors.
 *   - Return 0 on success, <0 on error (-1 invalid argument / unsupported shape, -3 launch
 *     failure); dge_last_error() returns the message.
 */
#ifndef DGE_HIP_H
#define DGE_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* dge_stream_t;   /* == hipStream_t */

#define DGE_DTYPE_F32 0
#define DGE_DTYPE_BF16 1
#define DGE_ACT_LINEAR 0
#define DGE_ACT_LRELU02 1
#define DGE_ACT_RELU_ 2
#define DGE_LIN_RSQRT 3
#define DGE_PACK_FWD 0      /* [tap][Cout][Cin]                                   */
#define DGE_PACK_UPFOLD 1   /* [tap][4*Cout][Cin]: conv_transpose(s2)+FIR folded  */
#define DGE_PACK_DGRAD 2    /* [tap][Cin][Cout], taps flipped                     */
#define DGE_PACK_UPFOLD_DGRAD 3 /* [tap][Cin][4*Cout]: adjoint of DGE_PACK_UPFOLD  */
#define DGE_PACK_SG1_UP 4    /* w is [Cin][Cout][3][3] (ConvTranspose2d 3,s2,p1 + transform_kernel, lreq.py:129-131) -> [tap][4*Cout][Cin] */
#define DGE_PACK_SG1_UP_DGRAD 5 /* same parameter layout -> [tap][Cin][4*Cout]: adjoint of DGE_PACK_SG1_UP */
#define DGE_PACK_UPT2D_DGRAD 6 /* [tap][Cin][4*Cout]: adjoint of the up layer in phase form, read from dge_fir_t2d output (dge_conv_desc.in_t2d) */

const char* dge_last_error(void);
int dge_version(void);

/* One fused convolution launch.  Replaces, depending on the fields set:
 *   - ModulateConvBlock.forward, model/stylegan2_generator.py:855-922 (style modulation =
 *     in_scale, demodulation = out_scale, noise :911-916, bias :918-920, lrelu*sqrt2 :921;
 *     `up` = conv_transpose2d stride 2 + filter :879-896 via DGE_PACK_UPFOLD weights);
 *   - BEBlock convs, model/E/E.py:58-62,69-74,81-84 (instance norm apply = in_scale/in_shift,
 *     noise_weight/bias/leaky_relu epilogue, residual blend via addend, statistics of the
 *     output for the next mean/std :64-65 via `stats`). */
typedef struct dge_conv_desc {
    const void* x;            /* [B,H,W,Cin]                       */
    const void* w_packed;     /* from dge_pack_conv_weight          */
    void* y;                  /* [B,OH,OW,Cout], OH = up ? 2H : H   */
    const void* addend;       /* optional, same shape as y          */
    const void* dot_src;      /* optional, same shape as y: stats = (sum acc*dot_src, sum acc) of the raw conv result */
    const float* in_scale;    /* optional [B,Cin]                   */
    const float* in_shift;    /* optional [B,Cin]                   */
    const float* out_scale;   /* optional [B,Cout]                  */
    const float* bias;        /* optional [Cout]                    */
    const float* noise;       /* optional [noise_batch,OH,OW]       */
    const float* noise_w;     /* [Cout] or [1]                      */
    float* stats;             /* optional [stats_slots][B,Cout,2], pre-zeroed */
    int B, H, W, Cin, Cout;
    int ksize;                /* 1 or 3 (stride 1, pad ksize/2)     */
    int up;                   /* 0 / 1                              */
    int in_s2d;               /* 1: x is [B,2H,2W,Cin/4] read space-to-depth (adjoint of up; DGE_PACK_UPFOLD_DGRAD) */
    int noise_batch;          /* 1 = shared noise plane, else B     */
    int noise_w_per_channel;  /* 0 scalar strength, 1 per channel   */
    int act;                  /* DGE_ACT_*                          */
    float bias_scale, gain, add_scale;
    int dtype;
    int in_up2;               /* 1: x is [B,H/2,W/2,Cin] read through a nearest x2 upsample (upscale2d, model/stylegan1/net.py:37-43) */
    int in_relu;              /* 1: ReLU after the prologue affine (BigGAN BN -> ReLU -> conv, model/biggan_generator.py:178-196) */
    int stats_slots;          /* >=1: workgroups spread their statistics atomics over this many copies (combine with dge_sum_slots) */
    int w_layout;             /* 0: [tap][N][K] rows; 1: MFMA-fragment order (pack mode | DGE_PACK_FRAG) - only where
                                 dge_conv_small_supported() says so: the low-resolution layers (4^2 .. 16^2 at 512 channels:
                                 stylegan2_generator.py:488-490 layers 0-4, E.py blocks at <= 16^2, LPIPS conv5_x) */
    /* Fused tail backward of the layer BELOW (data gradients of the synthesis chain; needs dot_src = that layer's stored
     * activation x = lrelu(yraw*d + noise*ns + bias)*gain, stylegan2_generator.py:908-921): instead of g = acc*out_scale + addend
     * (the gradient w.r.t. x) the launch stores g_z = g * prep_gain * lrelu'(x) and adds, per (b, c),
     * prep_stats[..][0] += sum g_z*(z - ns*noise), prep_stats[..][1] += sum g_z - what dge_modconv_bwd_prep does in a pass of
     * its own, minus the demodulation factor d (hand it to the NEXT data-gradient launch as in_scale).  With in_s2d, in_scale /
     * in_shift are per physical channel [B, Cin/4].  Not available in deterministic mode. */
    int prep;                 /* 0 / 1 */
    float prep_gain;          /* activation gain of the layer below (sqrt 2) */
    const float* prep_noise;  /* its noise plane [prep_noise_batch, OH, OW] or NULL */
    const float* prep_ns;     /* its noise strength, device scalar, or NULL */
    int prep_noise_batch;     /* 1 = shared plane, else B */
    float* prep_stats;        /* [stats_slots][B,Cout,2], pre-zeroed */
    /* ReLU backward of the layer BELOW fused into a data-gradient launch (LPIPS VGG16 backward, third-party lpips algorithm at the
     * call site training_utils.py:93): dot_src = that layer's stored activation a = relu(pre); the stored result is
     * (acc*out_scale + addend) * [a > 0].  No dot-product statistics in this mode (stats may be NULL). */
    int mask_relu;            /* 0 / 1 */
    /* Phase-form adjoint of the up layer (stylegan2_generator.py:879-896 differentiated): x is the output of dge_fir_t2d,
     * [B,H+1,W+1,Cin] with Cin = 4*Cout_up, weights DGE_PACK_UPT2D_DGRAD; the launch computes y[m] = sum over the taps
     * (dy,dx) in {1,2}^2 of x[m + (dy-1,dx-1)] * W[tap] (4 of 9 taps: 16 tap-units per pixel against the 36 of in_s2d). */
    int in_t2d;               /* 0 / 1 */
    /* toRGB of the result fused into the epilogue (SynthesisModule.forward :515-522, ModulateConvBlock k = 1 :465-474), offered where
     * dge_conv_rgb_supported() says so: rgb_out[b][k][y][x] = rgb_bias[k] + sum_c rgb_w[k][c]*rgb_wscale*rgb_style[b][c] * y[b][y][x][c]
     * (y as stored, i.e. rounded to dtype); the 2x-upsampled previous image is added by dge_rgb_upsample_add.  rgb_skip_y = 1: y
     * itself is not stored (inference: the toRGB is its only reader). */
    const float* rgb_w;       /* [3][Cout] f32 or NULL */
    const float* rgb_style;   /* [B][Cout] */
    const float* rgb_bias;    /* [3] */
    float* rgb_out;           /* [B][3][H][W] f32, NULL = no fused toRGB */
    float rgb_wscale;
    int rgb_skip_y;
    /* 2x2 average pool of the result taken in the epilogue (BEBlock: downscale2d after conv_2, model/E/E.py:75-76), offered where
     * dge_conv_pool_supported() says so: y is [B,H/2,W/2,Cout] (the mean of the four f32 results, rounded once); pool_mask (or NULL)
     * receives the signs of the four full-resolution values in dge_blend_pool_mask's layout ([B, H/2*W/2, Cout/8] words, byte q =
     * position (2oy, 2ox), (2oy, 2ox+1), (2oy+1, 2ox), (2oy+1, 2ox+1), bit e = channel e of the 8-channel chunk). */
    int pool_out;             /* 0 / 1 */
    void* pool_mask;
    /* Low-resolution launches (w_layout = 1) only, optional: w_packed / packed N / Cin of the NEXT such launch on this stream.  The
     * launch then warms the L2 slices that one will read (its weights are 4.7 MB per layer and arrive HBM-cold otherwise).  Pure
     * hint: no effect on results; NULL = none. */
    const void* prefetch_w;
    int prefetch_ntot, prefetch_cin;
    /* Data-gradient launches where dge_conv_in_bwd_supported() says so (conv_2 of the first encoder blocks, model/E/E.py:50-85
     * differentiated): the instance-norm backward of the layer's INPUT x = dot_src applied in the epilogue.  in_bwd_coef [B][Cout][3]
     * = (A, Bc, Cc) from dge_in_bwd_coef_slots (its sums from dge_conv_wgrad_dots, i.e. before this launch); the launch stores
     * g_pre = (A*acc + Bc*x + Cc) * lrelu'(x) - what dge_in_bwd_fused(act = 1) made of the stored data gradient in a pass of its
     * own - and adds (sum g_pre, sum g_pre*noise) per (sample, channel) to prep_stats [stats_slots][B][Cout][2] (the bias /
     * noise-weight gradients of the layer that produced x; noise = prep_noise [prep_noise_batch][H][W]).  stats / prep: none.
     * Without prep_stats (where dge_conv_in_bwd_x_supported() says so: conv_1 of a block, Cin = Cout): the block-input form, y = A*acc +
     * Bc*x + Cc + in_bwd_extra_scale * in_bwd_extra[parent pixel] (dge_in_bwd_fused(act = 0) with the pooled skip gradient, E.py:77-84). */
    const float* in_bwd_coef;
    /* The last data gradient of the encoder backward (conv_1 of block 0; where dge_conv_in_bwd_fromrgb_supported() says so), with
     * in_bwd_coef: x = dot_src is the FromRGB output (model/utils/net.py:231-240).  g_x = A*acc + Bc*x + Cc + in_bwd_extra_scale *
     * in_bwd_extra[parent pixel] (in_bwd_extra [B][H/2][W/2][Cout] bf16 or NULL: the pooled skip gradient, E.py:77-84) is NOT stored (y
     * is ignored): fr_out [stats_slots][B][Cout][4] (pre-zeroed) += sum_p g_x*lrelu'(x) * fr_img4[b][p][0..3] - the FromRGB weight /
     * bias gradients per sample, what dge_in_bwd_fromrgb computed in a pass of its own.  fr_img4 [B][H][W][4] f32 = (r, g, b, 1) per
     * pixel (dge_fromrgb's optional second output).  prep_stats / prep_noise unused. */
    const void* in_bwd_extra;
    float in_bwd_extra_scale;
    const float* fr_img4;
    float* fr_out;
} dge_conv_desc;
int dge_conv2d(const dge_conv_desc* d, dge_stream_t stream);
/* 1 when a 3x3 stride-1 launch of this shape runs on the low-resolution kernel (csrc/conv_small.hip) and therefore wants its
 * weights packed with DGE_PACK_FRAG; H, W = the conv's input grid, cin / ntot = packed K and N (4*Cout in up mode). */
int dge_conv_small_supported(int H, int W, int cin, int ntot, int ksize, int in_s2d, int in_up2, int dtype);
#define DGE_PACK_FRAG 0x100   /* OR-ed into the pack mode: fragment-ordered output for dge_conv_desc.w_layout = 1 */
/* Test / tuning hook: the kernel instantiation the calling thread's last dge_conv2d / dge_upconv_fir / dge_conv_wgrad call
 * selected, e.g. "conv_igemm<bf16,16,16,128,32,3,2,2>" (pixel tile TH x TW, N tile, K chunk, kernel size, wave grid),
 * "upconv_fir<bf16>", "conv_wgrad_tr<3,16>".  The parity tests assert by name that the configurations which carry the
 * benchmark (dispatch rules of csrc/conv_igemm.hip: launch_t) are the ones compared with the oracle. */
const char* dge_last_kernel(void);
/* The DGE_* tuning / test switches are read from the environment once, at the first launch; call this after changing one. */
void dge_env_reload(void);
int dge_sum_slots(const float* partial, float* out, int nslot, int n, int accumulate, dge_stream_t stream);
/* the same sum over slots of [C][NS] partials, written planar: out[k*C + c] */
int dge_sum_slots_planar(const float* partial, float* out, int nslot, int C, int NS, dge_stream_t stream);
/* up to 32 of them in one launch (deferred parameter-gradient reductions of an encoder backward) */
typedef struct dge_sum_planar_entry { const float* partial; float* out; int nslot, C, NS, pad; } dge_sum_planar_entry;
int dge_sum_slots_planar_multi(const dge_sum_planar_entry* entries, int n, dge_stream_t stream);

/* Weight preparation (once per weight update).  w_oihw: [Cout][Cin][k][k] f32 as stored by the
 * reference (model/stylegan2_generator.py:814-819; model/utils/lreq.py:107-110).  `out` holds
 * k*k * dge_packed_n(N) * K elements of `dtype` (N,K per mode above). */
/* Every style vector of a synthesis pass in one launch (the 17 + 9 `style` DenseBlocks of :825-829, 465-474): row r of the
 * row-concatenated weights w [R,K] / bias [R] reads the latent row x[b*ldx_b + row_xoff[r] .. +K) and writes
 * y[row_ybase[r] + b*row_ybstride[r]] = wscale*<x,w_r> + bias[r]*bscale + add, i.e. per-layer contiguous [B,C] blocks. */
int dge_linear_rows(const float* x, int ldx_b, const int* row_xoff, const float* w, const float* bias, float* y,
                    const int* row_ybase, const int* row_ybstride, int B, int R, int K, float wscale, float bscale, float add,
                    dge_stream_t stream);
/* Every demodulation factor of a synthesis pass in one launch (:867-870): row r = an output channel of some layer,
 * d = rsqrt(sum_c s[b,c]^2 * wsq[c] + eps) over that layer's style block s_all[row_sbase[r] + b*cin ..) and its
 * row wsq_cat[row_woff[r] ..) of dge_weight_sumsq values. */
int dge_demod_rows(const float* s_all, const float* wsq_cat, const int* row_woff, const int* row_sbase, const int* row_cin,
                   float* d_all, const int* row_dbase, const int* row_dbstride, int B, int R, float eps, dge_stream_t stream);
/* Deterministic mode (the reference pins torch.backends.cudnn.deterministic = True, training_utils.py:51): every reduction that
 * ends in same-address f32 atomics (conv statistics, weight-gradient flush, per-channel sums of the streaming backward kernels,
 * loss sums) goes through per-contributor slots and an ordered sum by the last contributor instead.  Synchronises the device. */
int dge_set_deterministic(int on);
int dge_get_deterministic(void);
int dge_packed_n(int n_valid);
int dge_pack_conv_weight(const float* w_oihw, void* out, int cout, int cin, int ksize, int mode, int dtype,
                         float scale, dge_stream_t stream);
/* All packed copies of a module's conv weights in one launch (the encoder's ~35 weights change at every optimizer step,
 * model/utils/custom_adam.py:88-102, and each is needed in forward and data-gradient layout).  table_host: n rows of
 * 8 x int64 on the HOST {w_oihw, out, cout | cin << 32, ksize | mode << 32, dtype, float bits of scale, 0, 0}; descs_dev:
 * device scratch of n * dge_pack_desc_bytes() bytes, written when upload != 0 (upload = 0: it still holds the descriptors
 * of an earlier call with the same table). */
int dge_pack_conv_weights_multi(const long long* table_host, void* descs_dev, int n, int upload, dge_stream_t stream);
int dge_pack_desc_bytes(void);
/* wsq[o][i] = scale^2 * sum_taps W^2 : the demodulation norm of :867-870 for the shared-weight form */
int dge_weight_sumsq(const float* w_oihw, float* wsq, int cout, int cin, int ksize, float scale, dge_stream_t stream);

/* y[b][o] = act((sum_i f(x[b][i]) W[o][i])*wscale + bias[o]*bscale + add)*gain, f = id or square.
 * DenseBlock.forward :990-996 (mapping :267-270, style :825-829) and ln.Linear (lreq.py:69-75);
 * with square_input=1, act=DGE_LIN_RSQRT it evaluates d[b][o] = rsqrt(s^2 . wsq + eps) (:867-870). */
int dge_linear(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int I, int O,
               float wscale, float bscale, float add, int act, float gain, int square_input, dge_stream_t stream);
int dge_pixelnorm(const float* x, float* y, int B, int D, float eps, dge_stream_t stream);          /* :550-553 */
/* A chain of up to 8 dense layers (dge_linear's arithmetic, bit for bit) in one launch, optionally behind dge_pixelnorm: the mapping
 * network z -> w (MappingModule.forward :262-278: pixel norm + 8 DenseBlocks).  Layer l: y = act((x W^T)*wscale + bias*bscale + add)*gain,
 * w [O][I] f32, I of layer l = O of layer l - 1, widths <= 1024.  x [B] rows of stride ldx, y [B] rows of stride ldy. */
typedef struct dge_dense_layer { const float* w; const float* bias; int I, O; float wscale, bscale, add; int act; float gain; } dge_dense_layer;
int dge_dense_chain(const float* x, int ldx, const dge_dense_layer* layers, int n, float* y, int ldy, int B, int pixelnorm, float eps,
                    dge_stream_t stream);
int dge_truncation(const float* w, const float* w_avg, float* wp, int B, int L, int D, float psi, int layers,
                   int w_is_wp, dge_stream_t stream);                                                /* :311-333 */

/* toRGB (1x1 modulated conv, no demod, linear) + bias + 2x FIR-upsampled previous image:
 * SynthesisModule.forward :515-522, ModulateConvBlock k=1 :465-474, UpsamplingLayer :603-615.
 * x [B,H,W,Cin] dtype; wrgb [3][Cin] f32; style [B,Cin]; prev [B,3,H/2,W/2] f32 or NULL; img [B,3,H,W] f32. */
int dge_torgb(const void* x, const float* wrgb, const float* style, const float* bias, const float* prev, float* img,
              int B, int H, int W, int cin, float wscale, int dtype, dge_stream_t stream);

int dge_conv_rgb_supported(int B, int H, int W, int Cin, int Cout, int ksize, int dtype);
int dge_conv_in_bwd_supported(int B, int H, int W, int Cin, int Cout, int ksize, int dtype);
int dge_conv_in_bwd_fromrgb_supported(int B, int H, int W, int Cin, int Cout, int ksize, int dtype);
int dge_conv_in_bwd_x_supported(int B, int H, int W, int Cin, int Cout, int ksize, int dtype);
int dge_conv_pool_supported(int B, int H, int W, int Cin, int Cout, int ksize, int dtype);
/* img[b][c][y][x] += up2(prev)[b][c][y][x]: the skip connection of SynthesisModule.forward :517-522 (UpsamplingLayer :603-615:
 * zero-insert, pad (2,1), 4x4 FIR == per-axis taps {.25,.75} / {.75,.25}) for an image whose toRGB term is already in img
 * (dge_conv_desc.rgb_out).  img [B,3,H,W] f32, prev [B,3,H/2,W/2] f32. */
int dge_rgb_upsample_add(float* img, const float* prev, int BC, int H, int W, dge_stream_t stream);
int dge_nchw_to_nhwc(const float* src, void* dst, int B, int C, int HW, int src_B, int dtype, dge_stream_t stream);
int dge_nhwc_to_nchw(const void* src, float* dst, int B, int C, int HW, int dtype, dge_stream_t stream);

/* ---- encoder E.BE (model/E/E.py) streaming kernels ---------------------------------------- */
/* FromRGB: 1x1 conv 3->C + bias + leaky_relu(0.2), model/utils/net.py:231-240.  img NCHW f32,
 * w [C][3], y NHWC dtype; optional per-(b,c) (sum, sumsq) of y into stats [B,C,2] (pre-zeroed). */
int dge_fromrgb(const float* img, const float* w, const float* bias, void* y, float* stats, int B, int HW, int C,
                int dtype, dge_stream_t stream);
/* the same, plus (img4 != NULL) the image in pixel-major form [B][HW][4] f32 = (r, g, b, 1): the operand of the fused FromRGB
 * parameter-gradient reduction of the backward (dge_conv_desc.fr_img4) */
int dge_fromrgb2(const float* img, const float* w, const float* bias, void* y, float* stats, float* img4, int B, int HW, int C,
                 int dtype, dge_stream_t stream);
/* (sum, sumsq) -> musig [B,2C] = [mean | biased std] (E.py:51-53,64-66) and the instance-norm
 * affine sc = rsqrt(var+eps), sh = -mean*sc (nn.InstanceNorm2d eps=1e-8, E.py:57,68). */
int dge_stats_finalize(const float* stats, float* musig, float* sc, float* sh, int B, int C, int npix, float eps,
                       dge_stream_t stream);
/* the same with `nslot` copies of the sums ([nslot][B][C][2], the layout dge_conv2d's stats_slots writes) added on the fly */
int dge_stats_finalize_slots(const float* stats, int nslot, float* musig, float* sc, float* sh, int B, int C, int npix, float eps,
                             dge_stream_t stream);
/* y = alpha * P(x*sc+sh) + beta * z with P = identity (pool=0) or avg_pool2d(2) (pool=1; y,z at
 * OHxOW, x at 2OHx2OW); sc/sh/z/stats optional.  E.py:75-78,84 (downscale2d, residual blend). */
int dge_blend(const void* x, const void* z, void* y, const float* sc, const float* sh, float* stats, int B, int OH,
              int OW, int C, int pool, float alpha, float beta, int dtype, dge_stream_t stream);

/* ---- space_loss (training_utils.py:54-99) and SSIM (metric/pytorch_ssim.py:18-38) ---------- */
/* One pass over a crop window of a,b [B,C,H,W] f32: sums7 = [16][8] slot copies (pre-zeroed; the caller adds the 16 slots,
 * e.g. dge_sum_slots, before dge_space_loss_finalize) += { sum (a-b)^2, a.b, a.a, b.b, sum a, sum b,
 * sum softmax_C(a)*(log softmax_C(a)-log softmax_C(b)) }  -> mse :63, mean/std terms :64-65, KL :67-71, cosine :73-75. */
int dge_loss_reduce(const float* a, const float* b, float* sums7, int B, int C, int H, int W, int y0, int x0, int h, int w,
                    dge_stream_t stream);
/* crop + k x k mean (the `while H > 256: avg_pool2d(2)` loop of :81-84 collapsed) on BC planes */
int dge_crop_pool(const float* src, float* dst, int BC, int H, int W, int y0, int x0, int h, int w, int k, dge_stream_t stream);
/* ssim_sum [32] (pre-zeroed slot copies; their total is the sum, dge_space_loss_finalize adds them) += sum of the SSIM map of a,b [BC,h,w]; dmap (optional, [3][BC][h][w]) receives dS/dmu2,
 * dS/dE[b^2], dS/dE[ab] for dge_ssim_bwd, which writes g = scale * dSum/db. */
int dge_ssim_fwd(const float* a, const float* b, float* ssim_sum, float* dmap, int BC, int h, int w, dge_stream_t stream);
/* skimage compare_ssim of comparing-baseline.py:25 (7x7 uniform window, sample covariance, 3-pixel border cropped): sums[bc]
 * (pre-zeroed) += sum of the SSIM map over the interior of plane bc of a,b [BC,h,w]; pixels enter as x*scale + shift, data range R */
int dge_ssim_box7(const float* a, const float* b, float* sums, int BC, int h, int w, float scale, float shift, float data_range,
                  dge_stream_t stream);
int dge_ssim_bwd(const float* a, const float* b, const float* dmap, float* g, int BC, int h, int w, float scale,
                 int accumulate, dge_stream_t stream);
/* out8 = { 5*mse + 3*cos + (1-ssim) + 2*lpips (:97), mse, mse(mean), mse(std), kl, cos, 1-ssim, lpips } on device */
int dge_space_loss_finalize(const float* sums7, const float* ssim_sum, const float* lpips, float* out8, float n,
                            float n_pooled, int image_space, dge_stream_t stream);
/* g[window] (+)= weight * d(5*mse + 3*cos)/db + weight * unpool_k(g_pooled)   (g_pooled optional) */
int dge_space_loss_bwd(const float* a, const float* b, const float* sums7, const float* g_pooled, float* g, int BC, int H,
                       int W, int y0, int x0, int h, int w, int k, float n, float weight, int accumulate, dge_stream_t stream);
/* The same three kernels for the nested attention windows of E_align_s2.py:185-203 (full image, AT1, AT2), every pixel touched
 * once: wins = nwin x (y0, x0, h, w), all inside window 0.  dge_loss_reduce3: sums [nwin][16][8] slot copies (pre-zeroed; add
 * them with dge_sum_slots; not offered in deterministic mode).  dge_crop_pool_multi: n <= 6 (source plane set, window, pooling
 * factor) entries.  dge_space_loss_bwd3: g over window 0 is WRITTEN with sum_k weight[k] * gradient of window k. */
int dge_loss_reduce3(const float* a, const float* b, float* sums, int B, int C, int H, int W, const int* wins, int nwin, dge_stream_t stream);
int dge_crop_pool_multi(const float* const* src, float* const* dst, const int* wins, const int* ks, int n, int BC, int H, int W,
                        dge_stream_t stream);
int dge_space_loss_bwd3(const float* a, const float* b, const float* const* sums7, const float* const* g_pooled, float* g, int BC, int H,
                        int W, const int* wins, const int* ks, const float* n, const float* weight, int nwin, dge_stream_t stream);
/* y (+)= x * scalar[0] * extra  (scalar may be NULL) */
int dge_axpy_scalar(const float* x, const float* scalar, float* y, long n, float extra, int accumulate, dge_stream_t stream);

/* ---- LREQAdam.step, model/utils/custom_adam.py:24-76, all tensors in one launch ------------ */
/* host_* are HOST arrays of `ntensors` device pointers / sizes / per-tensor step sizes
 * (= lr * sqrt(1 - beta2^t) * lr_equalization_coef, :62-72).  gscale: optional device scalar the
 * gradients are multiplied by first (e.g. 1/world_size after an all-reduce-sum).  step_mult: optional DEVICE scalar
 * multiplied into every step size (the sqrt(1 - beta2^t) factor when the step is replayed from a captured hipGraph). */
int dge_lreq_adam_multi(int ntensors, float* const* host_p, const float* const* host_g, float* const* host_v,
                        const long* host_n, const float* host_step, float beta2, float eps, const float* gscale,
                        const float* step_mult, dge_stream_t stream);

/* ---- data gradient of the StyleGAN2 synthesis network w.r.t. wp (E_align_s2.py:160,204) ---- */
/* Backward through noise/bias/lrelu*gain/demodulation of ModulateConvBlock (:905-921):
 * gy = gx*gain*lrelu'(x)*d[b,c];  R[b,c,:] (pre-zeroed) += { sum g_z*z, sum g_z*noise, sum g_z }. */
int dge_modconv_bwd_prep(const void* gx, const void* x, const float* d, const float* noise, void* gy, float* R, int B,
                         int HW, int C, int noise_batch, float gain, int dtype, dge_stream_t stream);
/* t[b,o] = -(R0 - ns*R1 - bias[o]*bscale*R2) * d^2 : demodulation gradient factor (:867-870) */
int dge_demod_bwd(const float* R, const float* d, const float* bias, const float* noise_strength, float* t, int B, int C,
                  float bscale, dge_stream_t stream);
/* y[b*ldy + k*incy] (+)= scale * mul[b,k] * sum_o x[b*ldx + o*incx] * W[o,k]   (transposed dense layer) */
int dge_linear_t(const float* x, int ldx, int incx, const float* w, const float* mul, float* y, int ldy, int incy, int B,
                 int O, int K, float scale, int accumulate, dge_stream_t stream);
/* toRGB backward (:465-474): gx[b,p,i] = s[b,i]*wscale*sum_c g[b,c,p] Wrgb[c,i]; gs[b,i] (pre-zeroed) += sum_p (..)*x */
int dge_torgb_bwd(const float* gimg, const void* x, const float* wrgb, const float* style, void* gx, float* gs, int B,
                  int HW, int C, float wscale, int dtype, dge_stream_t stream);
/* Fused forms of the two calls above for the top of the synthesis backward and for the per-layer demodulation gradient when the
 * tail backward ran inside the data-gradient launches (dge_conv_desc.prep):
 * dge_torgb_bwd_prep = dge_torgb_bwd followed by dge_modconv_bwd_prep on the same layer (its output feeds only the last toRGB,
 * stylegan2_generator.py:515-522,908-921): gz [B,HW,C] = g_z, gs [B,C] += toRGB style gradient, P [B,C,2] += (sum g_z*(z - ns*noise),
 * sum g_z), all pre-zeroed.  dge_demod_bwd_prep: t[b,o] = -(P0 - bias[o]*bscale*P1) * d^2 with `nslot` copies of P added. */
int dge_torgb_bwd_prep(const float* gimg, const void* x, const float* wrgb, const float* style, const float* noise,
                       const float* noise_strength, int noise_batch, void* gz, float* gs, float* P, int B, int HW, int C,
                       float wscale, float gain, int dtype, dge_stream_t stream);
/* First half of the phase-form adjoint of the up layer: Z[b,m,(py,px),c] = scale[b,c] * (FIR^T g)[b, 2m + p, c] on the
 * (2H+1)^2 grid of the transposed conv (FIR = outer([1,3,3,1]/4), pad 1, :603-615; entries beyond the grid are zero).
 * g [B,2H,2W,C] -> z [B,H+1,W+1,4C]; scale optional [B,C] (the demodulation factor that multiplies the gradient). */
int dge_fir_t2d(const void* g, const float* scale, void* z, int B, int H, int W, int C, int dtype, dge_stream_t stream);
/* Every style gradient of a synthesis backward in one launch (stylegan2_generator.py:858-864,908-909 differentiated; replaces
 * dge_demod_bwd_prep + 2 x dge_linear_t per conv block and one dge_linear_t per toRGB block).  Conv block: P = fused tail sums
 * [nslot_p][B,out_c,2], st = data-gradient statistics [nslot_s][B,in_c,2], d [B,out_c], s [B,in_c], bias [out_c], wsq [out_c][in_c];
 * toRGB block: P = NULL, gs [B,in_c].  wstyle [in_c][K] = the block's style DenseBlock weight, row = its row of wp.
 * g_wp [B,nrows,K] must be pre-zeroed; n <= 32, channels <= 512. */
typedef struct dge_s2_grad_entry {
    const float* P; const float* st; const float* d; const float* s; const float* bias; const float* wsq;
    const float* gs; const float* wstyle;
    int nslot_p, nslot_s, in_c, out_c, row;
    float bscale;
} dge_s2_grad_entry;
int dge_s2_style_grads(const dge_s2_grad_entry* entries, int n, float* g_wp, int B, int nrows, int K, float wscale, dge_stream_t stream);
int dge_demod_bwd_prep(const float* P, int nslot, const float* d, const float* bias, float* t, int B, int C, float bscale,
                       dge_stream_t stream);
/* adjoint of the skip-branch 2x FIR upsample (:603-615): g [BC,2h,2w] -> gprev [BC,h,w] */
int dge_up2_bwd(const float* g, float* gprev, int BC, int h, int w, dge_stream_t stream);

/* ---- encoder backward (model/E/E.py:50-85 differentiated; conv data gradients use dge_conv2d) */
/* dw[o][i][tap] (f32, OIHW like the parameter, pre-zeroed) += sum_{b,p} g[b,p,o] * (x*in_scale+in_shift)[b,p+tap,i] */
int dge_conv_wgrad(const void* g, const void* x, const float* in_scale, const float* in_shift, float* dw, int B, int H, int W,
                   int cout, int cin, int ksize, int dtype, dge_stream_t stream);
/* dge_conv_wgrad (3x3, bf16) that also leaves, per (sample, input channel), the two sums of the layer's data gradient g_x =
 * conv^T(g, w) that the instance-norm backward of the layer's input needs - dots[slot][b][i] += (sum_p g_x*x, sum_p g_x), the
 * statistics a dge_conv2d data-gradient launch with dot_src = x produces - out of the weight-gradient correlations: they are
 * known BEFORE the data gradient runs, whose epilogue can then apply that backward itself (dge_conv_desc.in_bwd_coef).
 * w [cout][cin][3][3] f32: the layer's weight (read rounded to bf16, as the data gradient reads it); dots [dots_slots][B][cin][2]
 * pre-zeroed.  Returns 1 and launches NOTHING where the streaming weight-gradient kernel does not cover the shape (or in
 * deterministic mode). */
int dge_conv_wgrad_dots(const void* g, const void* x, const float* in_scale, const float* in_shift, float* dw, const float* w,
                        float* dots, int dots_slots, int B, int H, int W, int cout, int cin, int dtype, dge_stream_t stream);
/* gpre = scale * gup[q(p)] * (a > 0 ? 1 : slope) (q = 2x2 pooling parent when pool);
   red[b,c,red_cols] (PER-SAMPLE partial sums, pre-zeroed: workgroups of different samples never contend on an address; the
   caller adds them over b, e.g. dge_sum_slots) += {sum gpre, sum gpre*noise [, sum over the gup grid of gup]}   (red_cols = 2 or 3; the third column is the
   bias gradient of a parallel 1x1 branch fed by the same gup, reference E.py conv_3) */
int dge_act_bwd(const void* gup, const void* a, const float* noise, void* gpre, float* red, int red_cols, int B, int H, int W, int C,
                int pool, float scale, float slope, int dtype, dge_stream_t stream);
/* The pooled case without the activation tensor: dge_blend_pool_mask (the forward's avg-pool blend, model/E/E.py:76-84) leaves the
 * signs of its input as mask [B, (H/2)*(W/2), C/ep] words (bit q*ep + e: child q of the 2x2 block, channel e of the 16-byte chunk);
 * dge_act_bwd_mask = dge_act_bwd(pool = 1) reading that mask: 1 bit per element instead of the stored activation. */
int dge_blend_pool_mask(const void* x, const void* z, void* y, float* stats, unsigned* mask, int B, int OH, int OW, int C, float alpha,
                        float beta, int dtype, dge_stream_t stream);
int dge_act_bwd_mask(const void* gup, const unsigned* mask, const float* noise, void* gpre, float* red, int red_cols, int B, int H, int W,
                     int C, float scale, float slope, int dtype, dge_stream_t stream);
/* FromRGB data gradient (model/utils/net.py:231-240 differentiated w.r.t. the image; embedding_img.py:88 feeds a generated,
 * gradient-carrying image into the encoder): gimg [B,3,HW] f32 = sum_c w[c][k] * gx[b,p,c] * lrelu'(x0[b,p,c]) */
int dge_fromrgb_dgrad(const void* gx, const void* x0, const float* w, float* gimg, int B, int HW, int C, int dtype, dge_stream_t stream);
/* upscale2d materialised (nearest x2, model/stylegan1/net.py:37-43) with a scale: y [B,2H,2W,C] = scale * x[B,H,W,C][y/2,x/2];
 * with scale 0.25 it is the adjoint of avg_pool2d(2) (downscale2d / the stride-2 transform_kernel conv of E_Blur.py:35) */
int dge_nearest_up2(const void* x, void* y, int B, int H, int W, int C, float scale, int dtype, dge_stream_t stream);
/* coefficients (A,Bc,Cc)[B,C,3] of the instance-norm + (mean,std) backward; see DESIGN.md */
int dge_in_bwd_coef(const float* dots, const float* gms, const float* musig, const float* sc, const float* sh, float* coef,
                    int B, int C, int npix, dge_stream_t stream);
int dge_in_bwd_coef_slots(const float* dots, int nslot, const float* gms, const float* musig, const float* sc, const float* sh,
                          float* coef, int B, int C, int npix, dge_stream_t stream);      /* dots: [nslot][B][C][2] */
/* gout = A*gy + Bc*x + Cc + extra_scale*extra[q(p)], then optional lrelu' of x with bias/noise reductions into
 * red [B,C,2] (per-sample partial sums, pre-zeroed; summed over b by the caller) */
int dge_in_bwd(const void* gy, const void* x, const float* coef, const void* extra, const float* noise, void* gout, float* red,
               int B, int H, int W, int C, int extra_pool, float extra_scale, int act, int dtype, dge_stream_t stream);
/* dge_in_bwd_coef_slots followed by dge_in_bwd as ONE launch (C <= 512): the workgroups compute their sample's coefficients from
 * (dots [nslot][B,C,2] or NULL, gms [B,2C] or NULL, musig [B,2C], sc, sh [B,C], npix) themselves. */
int dge_in_bwd_fused(const void* gy, const void* x, const float* dots, int nslot, const float* gms, const float* musig,
                     const float* sc, const float* sh, int npix, const void* extra, const float* noise, void* gout, float* red,
                     int B, int H, int W, int C, int extra_pool, float extra_scale, int act, int dtype, dge_stream_t stream);
/* out [B,C] (per-sample partial sums, pre-zeroed) += scale * sum_p x[b,p,c] */
int dge_chan_sum(const void* x, float* out, int B, int HW, int C, float scale, int dtype, dge_stream_t stream);
/* out4[b][o][0..2] += sum g_pre*img[c], out4[b][o][3] += sum g_pre with g_pre = gx*lrelu'(x0)  (FromRGB, net.py:231-240);
 * per-sample partial sums [B,C,4], pre-zeroed, summed over b by the caller */
int dge_fromrgb_bwd(const void* gx, const void* x0, const float* img, float* out4, int B, int HW, int C, int dtype,
                    dge_stream_t stream);
/* dge_in_bwd_fused (act = 0) followed by dge_fromrgb_bwd as ONE launch - the last step of the encoder backward (E.py:124 FromRGB,
 * net.py:231-240): x0 is the FromRGB output, the gradient w.r.t. x0 is reduced into out4 [B,C,4] (pre-zeroed) and never stored */
int dge_in_bwd_fromrgb(const void* gy, const void* x0, const float* dots, int nslot, const float* gms, const float* musig,
                       const float* sc, const float* sh, int npix, const void* extra, const float* img, float* out4,
                       int B, int H, int W, int C, int extra_pool, float extra_scale, int dtype, dge_stream_t stream);
/* Forward of every inver_mod head of the encoder in one launch (E.py:51-53,64-66): w[b, gcol_l + o] = musig_l[b,:] . W_l[o,:] + bias_l[o]
 * over the same entry table as dge_heads_bwd (entries carry the bias pointer); w [B, ldw]. */
int dge_heads_fwd(const void* dev_entries, int n, const float* musig_all, float* w, int ldw, int B, int O, dge_stream_t stream);
/* Backward of all `inver_mod` heads of the encoder (E.py:51-53,66-68: w_l = Linear(mean/std statistics)) in two launches.
 * dev_entries: n records {const float* W [O][I]; long moff, woff; int I, gcol, boff, pad} in DEVICE memory
 * (dge_head_entry_size() bytes); g [B][ldg] holds the gradient of head l at columns gcol .. gcol+O;
 * gms_all[moff + b*I + k] = sum_o g*W, gw_all[woff + o*I + i] = sum_b g*musig_all[moff + b*I + i], gb_all[boff + o] = sum_b g */
int dge_head_entry_size(void);
int dge_heads_bwd(const void* dev_entries, int n, int max_I, const float* g, int ldg, const float* musig_all, float* gms_all,
                  float* gw_all, float* gb_all, int B, int O, dge_stream_t stream);
/* gw[o][i] (+)= sum_b gy[b][o]*x[b][i]; gb[o] (+)= sum_b gy[b][o]   (ln.Linear parameter gradients) */
int dge_dense_wgrad(const float* gy, int ldgy, const float* x, int ldx, float* gw, float* gb, int B, int O, int I,
                    int accumulate, dge_stream_t stream);

/* ---- LPIPS(net='vgg') glue (third-party algorithm called at training_utils.py:93; VGG convs use dge_conv2d) */
int dge_lpips_prep(const float* img, void* x, int B, int HW, int cpad, const float* host_shift3, const float* host_scale3,
                   int dtype, dge_stream_t stream);                 /* ScalingLayer + NCHW f32 -> NHWC (channels padded) */
int dge_lpips_prep_bwd(const void* gx, float* gimg, int B, int HW, int cpad, const float* host_scale3, float factor,
                       int accumulate, int dtype, dge_stream_t stream);
int dge_maxpool2(const void* x, void* y, int B, int H, int W, int C, int dtype, dge_stream_t stream);   /* MaxPool2d(2,2) */
int dge_maxpool2_bwd(const void* gy, const void* x, const void* addend, void* gx, int B, int H, int W, int C, int dtype,
                     dge_stream_t stream);
/* one tap: feat [2B,h,w,C] (a then b); val[b] += spatial mean of sum_c lin_c (n_a - n_b)^2; g1 = gscale * dval/dfeat_b */
int dge_lpips_head(const void* feat, const float* lin, float* val, void* g1, int B, int HW, int C, float gscale, int dtype,
                   dge_stream_t stream);
int dge_mean(const float* v, float* out, int n, dge_stream_t stream);

/* ---- StyleGAN1 (model/stylegan1/net.py) streaming kernels ---------------------------------- */
/* y = act(blur3x3(x) + noise_w[c]*noise[b,p] + bias[c]) (act: 0 none, 1 lrelu 0.2; blur/noise/bias optional) with optional per-(b,c) statistics:
 * Blur :48-58 ([1,2,1]^2/16, zero pad) + the noise/bias/leaky_relu of DecodeBlock.forward :146-152,158-164. */
int dge_blur_noise_act(const void* x, const float* noise, const float* noise_w, const float* bias, void* y, float* stats,
                       int B, int H, int W, int C, int do_blur, int noise_batch, int act, int dtype, dge_stream_t stream);
/* instance norm + style_mod as one per-(b,c) affine: a = sc*(s0+1), b = sh*(s0+1) + s1 with style [B,2C] = [s0 | s1]
 * (style_mod :32-34 after InstanceNorm2d :153-156) */
int dge_affine_compose(const float* sc, const float* sh, const float* style, float* a, float* b, int B, int C, dge_stream_t stream);
/* out[b,l,:] = avg[l*avg_stride + :] + (w[b,:] - avg[...]) * coefs[l]   (Mapping.forward lerp, :459-466) */
int dge_lerp_layers(const float* w, const float* avg, int avg_stride, const float* coefs, float* out, int B, int L, int D,
                    dge_stream_t stream);

/* StyleGAN1 synthesis data gradient (DecodeBlock.forward :141-169 differentiated w.r.t. the styles; the generator's own
 * parameters get no gradients, E_align_s2.py trains the encoder only):
 * coefficients (A,Bc,Cc)[B,C,3] of g_y = A*g_u + Bc*y + Cc for u = style_mod(InstanceNorm(y)) (:153-156, :32-34) from
 * dots [B,C,2] = (sum g_u*y, sum g_u), and the style gradient gstyle [B,2C] = [g_s0 | g_s1]. */
int dge_sg1_in_bwd_coef(const float* dots, const float* sc, const float* sh, const float* style, float* coef, float* gstyle,
                        int B, int C, int npix, dge_stream_t stream);
/* stats [B,C,2] (pre-zeroed) += (sum_p g*x, sum_p g) for NHWC g, x */
int dge_dot_stats(const void* g, const void* x, float* stats, int B, int HW, int C, int dtype, dge_stream_t stream);
/* adjoint of upscale2d (nearest x2, :37-43): glow [B,H,W,C] = 2x2 block sums of ghi [B,2H,2W,C]; optional dot statistics
 * of glow against x [B,H,W,C] as in dge_dot_stats */
int dge_nearest_up2_bwd(const void* ghi, const void* x, void* glow, float* stats, int B, int H, int W, int C, int dtype,
                        dge_stream_t stream);

/* torch.nn.utils.spectral_norm (snconv2d / snlinear, biggan_generator.py:28-56) for a GROUP of weights in five launches:
 * train mode = one power iteration (v = normalize(W^T u), u = normalize(W v), written into the module's buffers), sigma = u.(W v),
 * W_eff = W / sigma.  `entries`: device array of n records {const float* W; float* u, *v, *t, *s, *weff, *usnap, *vsnap; int O, K}
 * (dge_sn_entry_size() bytes each; t pre-zeroed in train mode; usnap / vsnap receive the u, v used for sigma). */
int dge_sn_group(const void* entries, int n, int maxO, int maxK, float* sigma, float eps, int training, dge_stream_t stream);
int dge_sn_entry_size(void);
/* Weight gradients of all conditional-batch-norm linears of a backward in two launches (biggan BigGANBatchNorm :141-144 differentiated, through the
 * spectral norm with u, v constant): entry = {dots [B][C][2] = (dL/da, dL/db) of the norm's affine, mean [C], rstd [C] (scale linear; NULL for the
 * offset linear: kind 1), weight_orig [C][K] (live), u [C], v [K], sigma [1], out [C][K], C, kind, row0 = rows before this entry};
 * out = (gw - <gw, W> / sigma * u v^T) / sigma with gw[o][k] = sum_b gy[b][o] cond[b][k].  rowdot: scratch of `rows` floats. */
int dge_cbn_sn_wgrad_entry_size(void);
int dge_cbn_sn_wgrad_group(const void* entries, int n, long long rows, int maxC, const float* cond, int B, int K, float* rowdot, dge_stream_t stream);

/* ---- counter-based normal noise (Philox4x32-10 + Box-Muller; csrc/rng_kernels.hip) ----
 * Replaces the torch.randn draws of the training step (model/E/E.py:60,73 encoder noise; stylegan2_generator.py:187 new_z,
 * :911-913 randomize_noise; model/stylegan1/net.py noise) with draws that are a pure function of (seed, subseq, element index):
 * segment i fills out[start[i] .. start[i]+count[i]) with elements goff[i].. of draw subseq[i], so a data-parallel rank
 * generates exactly its rows of the global-batch tensor.  seed_dev (optional device scalar) overrides `seed` (hipGraph replay). */
int dge_randn(float* out, int nseg, const long long* start, const long long* count, const unsigned long long* goff,
              const unsigned int* subseq, unsigned long long seed, const unsigned long long* seed_dev, dge_stream_t stream);

/* ---- StyleGAN2 up layer at algorithmic cost (stylegan2_generator.py:879-896 conv_transpose2d + 4x4 FIR, :911-921) ----
 * Transposed conv in phase form on the MFMAs (9 tap-MACs per input pixel instead of the 36 of the folded 3x3-per-phase
 * form of dge_conv2d(up=1)), FIR + demodulation / noise / bias / activation from LDS in the same kernel.
 * w_packed: dge_pack_upconv_weight ([9 (phase,tap) units][Cout][Cin]).  Supported when dge_upconv_supported(). */
int dge_upconv_supported(int Cin, int Cout, int dtype);
int dge_pack_upconv_weight(const float* w, void* out, int Cout, int Cin, float scale, int dtype, dge_stream_t stream);
int dge_upconv_fir(const void* x, const void* w_packed, void* y, const float* in_scale, const float* out_scale,
                   const float* noise, int noise_bstride, const float* noise_w, int noise_w_stride, const float* bias,
                   float bias_scale, float gain, int act, int B, int H, int W, int Cin, int Cout, int dtype, dge_stream_t stream);

/* ---- The same up layer as a ping-pong implicit GEMM for the MFMA-bound layers (csrc/up_pp.hip; Cin >= 128: layers 7 / 9 / 11 / 13
 * of the 1024^2 generator).  stylegan2_generator.py:879-896 conv_transpose2d + FIR in the reference's FUSED-modulation form
 * (:858-875): dge_pack_up_pp folds style, demodulation and gain into one weight image per sample,
 * W'[b][unit][o][i] = bf16(w_units[unit][o][i] * in_scale[b][i] * out_scale[b][o] * gain) with w_units = dge_pack_upconv_weight's
 * [9][Cout][Cin] bf16 tensor (9*Cin*Cout elements per copy; nb = 1 shared copy or B copies); dge_up_pp runs
 * y[b] = act(FIR(conv_transpose2d(x[b], W'[b])) + noise*noise_w*gain + bias*bias_scale*gain)   (:908-921), x [B,H,W,Cin] ->
 * y [B,2H,2W,Cout], bf16.  w_bstride: bf16 elements between the samples' images (0 = shared).  Shapes per dge_up_pp_supported(). */
int dge_up_pp_supported(int B, int H, int W, int Cin, int Cout, int dtype);
int dge_pack_up_pp(const void* w_units, void* out, int Cout, int Cin, const float* in_scale, const float* out_scale, float gain,
                   int nb, dge_stream_t stream);
int dge_up_pp(const void* x, const void* w_img, long long w_bstride, void* y, const float* noise, int noise_bstride,
              const float* noise_w, const float* bias, float bias_scale, float gain, int act, int B, int H, int W, int Cin, int Cout,
              dge_stream_t stream);

/* ---- Ping-pong implicit GEMM for the MFMA-bound 3x3 stride-1 layers (csrc/conv_pp.hip) -----------------------------------
 * ModulateConvBlock.forward, stride-1 branch, in the reference's FUSED-modulation form (stylegan2_generator.py:858-875: the
 * style multiplies the weight, the demodulation divides it, one weight per sample; :898-904 conv; :911-921 noise, bias,
 * lrelu*sqrt2) for the layers with >= 128 output channels at 64^2 .. 256^2, and plain conv + bias + ReLU (LPIPS' VGG16).
 * dge_pack_conv_pp writes the LDS image the kernel streams: per (sample, 128-wide N tile, 32-channel K chunk, tap) one 8 KiB
 * block; `nb` = 1 shared copy or B per-sample copies with in_scale [nb][K] / out_scale [nb][N] / gain folded in:
 * W'[b][n][k] = dtype((w*wscale) * (in_scale[b][k] * (gain*out_scale[b][n]))).  mode 0: (n, k) = (out, in) channel of w_oihw
 * [N][K][3][3]; mode 1 (data gradient): w_oihw is [K][N][3][3], taps flipped.  9*N*K elements per copy.
 * dge_conv_pp: y[b] = act(conv3x3(x[b], W'[b]) * m + noise*noise_w*gain + bias*bias_scale*gain), m = out_scale*gain, or 1
 * when out_scale is NULL (folded weights carry demodulation and gain: `gain` then only multiplies noise and bias).
 * bf16 only; shapes per dge_conv_pp_supported(). */
typedef struct dge_conv_pp_desc {
    const void* x;            /* [B,H,W,Cin] bf16 */
    const void* w_pp;         /* dge_pack_conv_pp */
    void* y;                  /* [B,H,W,Cout] bf16 */
    long long w_bstride;      /* bf16 ELEMENTS between the samples' weight copies (9*Cin*Cout; the image is always bf16), 0 = one shared copy */
    const float* out_scale;   /* optional [B,Cout] */
    const float* bias;        /* optional [Cout] */
    const float* noise;       /* optional [noise_batch,H,W] */
    const float* noise_w;     /* [Cout] or [1] */
    int B, H, W, Cin, Cout;
    int noise_batch;          /* 1 = shared plane, else B */
    int noise_w_per_channel;
    int act;                  /* DGE_ACT_* */
    float bias_scale, gain;
    /* Data-gradient form (dgrad = 1; the epilogue menu of dge_conv_desc's dot_src / addend / stats / prep / mask_relu, same math:
     * stylegan2_generator.py:908-921 adjoint, model/E/E.py:60-84 adjoint): on the raw accumulator a of W' (dge_pack_conv_pp mode 1 or
     * dge_pack_conv_pp_rows, the demodulation factor of the layer above folded into K) -
     *   stats[slot][b][c] += (sum a*dot_src, sum a) (with prep only the first: the synthesis chain has no use for the plain sum);
 *   v = a*out_scale[b][c] + add_scale*addend;  mask_relu: v *= [dot_src > 0] (no statistics);
     *   prep: v = g_z = v*prep_gain*lrelu'(dot_src), prep_stats[slot][b][c] += (sum g_z*(z - ns*noise), sum g_z), z = dot_src / (gain*slope).
     * No bias / noise / activation.  in_s2d: x is [B,2H,2W,Cin/4] read space-to-depth (adjoint of the up layer in folded form,
     * weights from dge_pack_conv_weight mode DGE_PACK_UPFOLD_DGRAD in f32 through dge_pack_conv_pp_rows). */
    int dgrad, in_s2d, stats_slots, prep, prep_noise_batch, mask_relu;
    int in_t2d;               /* x is dge_fir_t2d's [B,H+1,W+1,Cin]; taps (dy,dx) in {1,2}^2 only (weights: dge_pack_conv_pp_rows with t2d = 1); needs prep */
    float add_scale, prep_gain;
    const void* dot_src;      /* [B,H,W,Cout] bf16 */
    const void* addend;       /* [B,H,W,Cout] bf16 */
    float* stats;             /* [stats_slots][B,Cout,2], pre-zeroed */
    float* prep_stats;        /* [stats_slots][B,Cout,2], pre-zeroed */
    const float* prep_noise;  /* [prep_noise_batch,H,W] or NULL */
    const float* prep_ns;     /* device scalar or NULL */
} dge_conv_pp_desc;
int dge_conv_pp_supported(int B, int H, int W, int Cin, int Cout, int dtype);
int dge_pack_conv_pp(const float* w_oihw, void* out, int N, int K, float wscale, const float* in_scale, const float* out_scale,
                     float gain, int nb, int mode, dge_stream_t stream);
/* the same LDS image from rows of an f32 dge_pack_conv_weight copy [9][src_rows][K] (any of its modes: the folded up layer's data
 * gradient, ...); in_scale [nb][in_period] repeats along K with period in_period (space-to-depth: the four phases of a channel);
 * t2d = 1: the 4-tap image of the phase-form adjoint (DGE_PACK_UPT2D_DGRAD rows; 4*N*K elements per copy) */
int dge_pack_conv_pp_rows(const float* w_rows, int src_rows, void* out, int N, int K, const float* in_scale, int in_period,
                          const float* out_scale, float gain, int nb, int t2d, dge_stream_t stream);
int dge_conv_pp(const dge_conv_pp_desc* d, dge_stream_t stream);

/* ---- PGGAN (model/pggan/pggan_generator.py) ------------------------------------------------ */
/* pixel-wise feature normalisation over the channel axis of an NHWC tensor (PixelNormLayer :207-216) */
int dge_pixelnorm_nhwc(const void* x, void* y, long npix, int C, float eps, int dtype, dge_stream_t stream);
/* backward of dge_pixelnorm_nhwc: gx = r*(gy - yhat*mean_c(gy*yhat)), r = rsqrt(mean_c x^2 + eps), yhat = x*r */
int dge_pixelnorm_nhwc_bwd(const void* gy, const void* x, void* gx, long npix, int C, float eps, int dtype, dge_stream_t stream);

/* ---- BigGAN-deep (model/biggan_generator.py) ----------------------------------------------- */
/* BigGANBatchNorm :127-150 as a per-(b,c) affine: a = (1 + scale[b,c]) / sqrt(var[c]+eps), b = offset[b,c] - mean[c]*a
 * (scale/offset row stride 0 = batch-shared, for the non-conditional final BN pass weight-1 and bias) */
int dge_cbn_affine(const float* scale, const float* offset, int ld, const float* mean, const float* var, float eps, float* a,
                   float* b, int B, int C, dge_stream_t stream);
/* y[b,oy,ox,c] = x[b,oy>>up,ox>>up,c], c < Cout <= Cin   (skip path of GenBlock :197-201: drop channels + nearest x2) */
int dge_slice_up(const void* x, void* y, int B, int H, int W, int Cin, int Cout, int up, int dtype, dge_stream_t stream);
/* O[b,n,:] = sum_m softmax_m(Q[b,n,:].K[b,m,:]) V[b,m,:]   (SelfAttn :75-97; Q [B,N,D], K [B,M,D], V [B,M,DV], O [B,N,DV]) */
int dge_attention(const void* q, const void* k, const void* v, void* o, int B, int N, int M, int D, int DV, int dtype,
                  dge_stream_t stream);
/* img[b,c,p] = tanh(x[b,p,c]) for c < 3   (Generator.forward :251-255; x NHWC with >= 3 channels) */
int dge_rgb_tanh(const void* x, float* img, int B, int HW, int C, int dtype, dge_stream_t stream);
/* ---- BigGAN-deep backward (data gradient w.r.t. z for E_align --mtype 4, E_align_s2.py:140-162) ---- */
/* backward of the conv prologue u = relu(a*x + b) (BigGANBatchNorm :127-150 + relu :180-197):
 * gx = [a*x+b > 0]*a*gu; stats [B,C,2] (pre-zeroed) += (d/da, d/db) */
int dge_affine_relu_bwd(const void* gu, const void* x, const float* a, const float* b, void* gx, float* stats, int B, int HW, int C,
                        int dtype, dge_stream_t stream);
/* adjoint of dge_slice_up accumulated into gx [B,H,W,Cin]: channels < Cout receive the 2^up x 2^up block sums of gy */
int dge_slice_up_bwd(const void* gy, void* gx, int B, int H, int W, int Cin, int Cout, int up, int dtype, dge_stream_t stream);
/* SelfAttn :75-97 backward pieces: in-place row softmax of the recomputed scores [R,M]; gS = P*(gP - sum P*gP) over gP */
int dge_softmax_rows(void* S, long R, int M, int dtype, dge_stream_t stream);
int dge_softmax_rows_bwd(const void* P, void* gP, long R, int M, int dtype, dge_stream_t stream);
/* backward of dge_rgb_tanh (:251-254): gy [B,HW,C] NHWC, channels >= 3 zero */
int dge_rgb_tanh_bwd(const float* gimg, const float* img, void* gy, int B, int HW, int C, int dtype, dge_stream_t stream);

/* ---- Grad-CAM++ attention maps (metric/grad_cam.py; wired by E_mis_align_cropping_s1.py:99-106,159-170).  The VGG16
 * convolutions of the classifier run on dge_conv2d (forward, data gradient), its dense layers on dge_linear / dge_linear_t. */
/* gpre = (a > 0 && (!guided || g > 0)) ? g : 0 over n elements: nn.ReLU backward + GuidedBackPropagation.backward_hook
 * (grad_cam.py:208-217, clamp(grad_in, min=0)); a = the ReLU's output */
int dge_guided_relu_bwd(const void* g, const void* a, void* gpre, long n, int guided, int dtype, dge_stream_t stream);
/* MaxPool2d(2,2) backward followed by the backward of the ReLU that produced x (guided: grad_cam.py:208-217), one pass */
/* adjoint of max-pool 2x2 + addend, then the ReLU backward of the layer that produced x: gx = (route(gy) + addend) * [x > 0] */
int dge_maxpool2_bwd_relu(const void* gy, const void* x, const void* addend, void* gx, int B, int H, int W, int C, int dtype,
                          dge_stream_t stream);
int dge_maxpool2_relu_bwd(const void* gy, const void* x, void* gx, int B, int H, int W, int C, int guided, int dtype,
                          dge_stream_t stream);
/* vgg16.avgpool + torch.flatten: x NHWC [B,H,W,C] -> y [B, C*49] f32 in (c, i, j) order; and its adjoint */
int dge_adaptive_pool7(const void* x, float* y, int B, int H, int W, int C, int dtype, dge_stream_t stream);
int dge_adaptive_pool7_bwd(const float* gy, void* gx, int B, int H, int W, int C, int dtype, dge_stream_t stream);
/* grad_cam.py:166-170: index = per-row argmax of logits [B,K] (or index_in [B] when given), index_max = its most frequent
 * value (smallest on ties) -> index_out[0] (index_out[1..B] = the per-row indices); glogits [B,K] = d mean_n logits[n,index_max] */
int dge_class_target(const float* logits, const int* index_in, int* index_out, float* glogits, int B, int K, dge_stream_t stream);
/* y[b,:] = scale * w[index[0], :]  (w [O,I] f32, index on the device) */
int dge_gather_row(const float* w, const int* index, float* y, int B, int I, float scale, dge_stream_t stream);
/* per sample, mode 1 = Grad-CAM++ (grad_cam.py:180-191): weight[c] = sum relu(grad) * 1/sum relu(grad) (0 when the sum is 0),
 * cam[p] = sum_c feat[p,c]*weight[c]; mode 0 = Grad-CAM (:101-105): weight[c] = mean grad, cam = relu(sum).  wgt [B,C],
 * cam [B,HW], minmax [B,2] = (min, max) of cam.  grad / feat NHWC [B,HW,C] */
int dge_campp_map(const void* grad, const void* feat, float* wgt, float* cam, float* minmax, int B, int HW, int C, int mode,
                  int dtype, dge_stream_t stream);
/* mask [B,1,H,W] = cv2.resize((cam - min) / (max - min), (W, H)) with INTER_LINEAR geometry (grad_cam.py:190-193) */
int dge_cam_resize(const float* cam, const float* minmax, float* mask, int B, int h, int w, int H, int W, dge_stream_t stream);
/* mask2cam (grad_cam.py:234-251): heat [B,3,H,W] = lut[uint8(255*mask)] / 255 (lut: 256 x (r,g,b) ints), cam = heat + img,
 * then the reference's sequential normalisation (minimum over the whole array as it stands when sample i is reached).
 * part: scratch [B, dge_mask2cam_blocks(HW), 3] f32, coef: scratch [B,2] f32 */
int dge_mask2cam_blocks(int HW);
int dge_mask2cam(const float* mask, const float* img, const int* lut, float* heat, float* cam, float* part, float* coef, int B,
                 int HW, dge_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
