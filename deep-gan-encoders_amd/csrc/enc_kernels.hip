// Encoder (E.BE) support kernels: FromRGB, instance-norm statistics, pooling / residual blends.
// All are HBM-bound streaming kernels over NHWC tensors: 16-byte accesses per lane, per-(b,c)
// statistics accumulated in registers -> LDS -> one atomic per channel per workgroup.
// Reference: model/E/E.py:50-85 (BEBlock.forward), model/utils/net.py:231-240 (FromRGB).
#include "common.h"
#include "../../include/dge_hip.h"

// ------------------------------------------------------------------ FromRGB
// y[b,p,o] = lrelu(sum_c img[b,c,p] W[o,c] + bias[o], 0.2)   img NCHW f32 -> y NHWC T, + stats
template <typename T>
__global__ __launch_bounds__(256) void fromrgb_kernel(const float* __restrict__ img, const float* __restrict__ W,
                                                       const float* __restrict__ bias, T* __restrict__ y,
                                                       float* __restrict__ stats, int HW, int C, float4* __restrict__ img4) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 2 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    float w0[EP], w1[EP], w2[EP], bb[EP], sq[2][EP];
#pragma unroll
    for (int e = 0; e < EP; e++) {
        const int o = chunk * EP + e;
        w0[e] = W[o * 3 + 0]; w1[e] = W[o * 3 + 1]; w2[e] = W[o * 3 + 2]; bb[e] = bias[o];
        sq[0][e] = 0.f; sq[1][e] = 0.f;
    }
    const float* ib = img + (size_t)b * 3 * HW;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            const float r = ib[p], g = ib[HW + p], bl = ib[2 * HW + p];
            if (img4 && chunk == 0) img4[(size_t)b * HW + p] = make_float4(r, g, bl, 1.f);      // pixel-major copy for the backward (dge_conv_desc.fr_img4)
            float f[EP];
#pragma unroll
            for (int e = 0; e < EP; e++) {
                float v = r * w0[e] + g * w1[e] + bl * w2[e] + bb[e];
                v = v > 0.f ? v : 0.2f * v;
                f[e] = v; sq[0][e] += v; sq[1][e] += v * v;
            }
            *(uint4*)(y + ((size_t)b * HW + p) * C + chunk * EP) = pack16(f, (T*)nullptr);
        }
    }
    if (stats) block_chan_flush<EP, 2>(sq, cpt, ppi, stats + (size_t)b * C * 2, C, red);
}

// ------------------------------------------------------------------ stats -> mean/std + IN affine
// stats [B,C,2] (sum, sumsq) over npix pixels ->
//   musig [B,2C] = [mean | sqrt(biased var)]          (E.py:51-53, no eps)
//   sc [B,C] = rsqrt(var + eps), sh = -mean*sc         (InstanceNorm2d eps=1e-8, E.py:57)
// (nslot copies of the sums - dge_conv2d spreads its statistics atomics - are added here: no separate slot-sum launch)
__global__ void stats_finalize_kernel(const float* __restrict__ stats, float* __restrict__ musig, float* __restrict__ sc,
                                      float* __restrict__ sh, int B, int C, float inv_n, float eps, int nslot) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx % C;
    const float2 ss = sum_slot_pairs(stats + (size_t)idx * 2, (size_t)B * C * 2, nslot);
    const float m = ss.x * inv_n;
    float v = ss.y * inv_n - m * m;
    v = v > 0.f ? v : 0.f;
    musig[(size_t)b * 2 * C + c] = m;
    musig[(size_t)b * 2 * C + C + c] = sqrtf(v);
    const float r = rsqrtf(v + eps);
    sc[idx] = r; sh[idx] = -m * r;
}

// ------------------------------------------------------------------ pooling / blends
// mode 0: y = alpha * f(x)                 + beta * z     (same resolution)
// mode 1: y = alpha * avgpool2(f(x))       + beta * z     (y, z at half resolution)
// f(x) = x*sc[b,c] + sh[b,c] when sc != null (instance-norm apply), else identity.
// Optional per-(b,c) statistics of y.
template <typename T>
__global__ __launch_bounds__(256) void blend_kernel(const T* __restrict__ x, const T* __restrict__ z, T* __restrict__ y,
                                                     const float* __restrict__ sc, const float* __restrict__ sh,
                                                     float* __restrict__ stats, int OH, int OW, int C, int pool,
                                                     float alpha, float beta, unsigned* __restrict__ mask) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 2 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    const int OHW = OH * OW, IW = pool ? 2 * OW : OW, IHW = pool ? 4 * OHW : OHW;
    float sq[2][EP], a[EP], d[EP];
#pragma unroll
    for (int e = 0; e < EP; e++) {
        sq[0][e] = 0.f; sq[1][e] = 0.f;
        a[e] = sc ? sc[(size_t)b * C + chunk * EP + e] : 1.f;
        d[e] = sh ? sh[(size_t)b * C + chunk * EP + e] : 0.f;
    }
    const T* xb = x + (size_t)b * IHW * C;
    for (int p0 = blockIdx.x * ppi; p0 < OHW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < OHW) {
            float f[EP];
            if (pool) {
                const int oy = p / OW, ox = p % OW;
                const T* base = xb + ((size_t)(2 * oy) * IW + 2 * ox) * C + chunk * EP;
                float f0[EP], f1[EP], f2[EP], f3[EP];
                unpack16(*(const uint4*)base, f0, (T*)nullptr);
                unpack16(*(const uint4*)(base + C), f1, (T*)nullptr);
                unpack16(*(const uint4*)(base + (size_t)IW * C), f2, (T*)nullptr);
                unpack16(*(const uint4*)(base + (size_t)IW * C + C), f3, (T*)nullptr);
#pragma unroll
                for (int e = 0; e < EP; e++) f[e] = 0.25f * (f0[e] + f1[e] + f2[e] + f3[e]);
                if (mask) {            // signs of the 4 x EP pooled inputs: all the activation backward needs of them (dge_act_bwd_mask)
                    unsigned m = 0;
#pragma unroll
                    for (int e = 0; e < EP; e++)
                        m |= (f0[e] > 0.f ? 1u : 0u) << e | (f1[e] > 0.f ? 1u : 0u) << (EP + e) | (f2[e] > 0.f ? 1u : 0u) << (2 * EP + e) |
                             (f3[e] > 0.f ? 1u : 0u) << (3 * EP + e);
                    mask[((size_t)b * OHW + p) * cpt + chunk] = m;
                }
            } else {
                unpack16(*(const uint4*)(xb + (size_t)p * C + chunk * EP), f, (T*)nullptr);
            }
#pragma unroll
            for (int e = 0; e < EP; e++) f[e] = alpha * (f[e] * a[e] + d[e]);
            if (z) {
                float g[EP];
                unpack16(*(const uint4*)(z + ((size_t)b * OHW + p) * C + chunk * EP), g, (T*)nullptr);
#pragma unroll
                for (int e = 0; e < EP; e++) f[e] += beta * g[e];
            }
#pragma unroll
            for (int e = 0; e < EP; e++) { sq[0][e] += f[e]; sq[1][e] += f[e] * f[e]; }
            *(uint4*)(y + ((size_t)b * OHW + p) * C + chunk * EP) = pack16(f, (T*)nullptr);
        }
    }
    if (stats) block_chan_flush<EP, 2>(sq, cpt, ppi, stats + (size_t)b * C * 2, C, red);
}

// ------------------------------------------------------------------ StyleGAN1 blur + noise + bias + lrelu
// y = lrelu(blur(x) + nw[c]*noise[b,p] + bias[c], 0.2), blur = depthwise [1,2,1]x[1,2,1]/16 with zero padding
template <typename T>
__global__ __launch_bounds__(256) void blur_noise_act_kernel(const T* __restrict__ x, const float* __restrict__ noise,
                                                              const float* __restrict__ nw, const float* __restrict__ bias,
                                                              T* __restrict__ y, float* __restrict__ stats, int H, int W, int C,
                                                              int do_blur, int noise_bstride, int act) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 2 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    const int HW = H * W;
    float sq[2][EP], nwv[EP], bv[EP];
#pragma unroll
    for (int e = 0; e < EP; e++) {
        sq[0][e] = sq[1][e] = 0.f;
        nwv[e] = nw ? nw[chunk * EP + e] : 0.f; bv[e] = bias ? bias[chunk * EP + e] : 0.f;
    }
    const T* xb = x + (size_t)b * HW * C;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            float f[EP];
            if (do_blur) {
                const int py = p / W, px = p % W;
#pragma unroll
                for (int e = 0; e < EP; e++) f[e] = 0.f;
#pragma unroll
                for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                    for (int dx = -1; dx <= 1; dx++) {
                        const int yy = py + dy, xx = px + dx;
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                            float g[EP];
                            unpack16(*(const uint4*)(xb + ((size_t)yy * W + xx) * C + chunk * EP), g, (T*)nullptr);
                            const float wgt = (float)((2 - (dy < 0 ? -dy : dy)) * (2 - (dx < 0 ? -dx : dx))) * (1.f / 16.f);
#pragma unroll
                            for (int e = 0; e < EP; e++) f[e] += wgt * g[e];
                        }
                    }
            } else {
                unpack16(*(const uint4*)(xb + (size_t)p * C + chunk * EP), f, (T*)nullptr);
            }
            const float nz = noise ? noise[(size_t)b * noise_bstride + p] : 0.f;
#pragma unroll
            for (int e = 0; e < EP; e++) {
                float v = f[e] + nwv[e] * nz + bv[e];
                if (act) v = v > 0.f ? v : 0.2f * v;
                f[e] = v; sq[0][e] += v; sq[1][e] += v * v;
            }
            *(uint4*)(y + ((size_t)b * HW + p) * C + chunk * EP) = pack16(f, (T*)nullptr);
        }
    }
    if (stats) block_chan_flush<EP, 2>(sq, cpt, ppi, stats + (size_t)b * C * 2, C, red);
}

// a = sc*(s0+1), b = sh*(s0+1) + s1  (instance norm followed by style_mod as one affine)
__global__ void affine_compose_kernel(const float* __restrict__ sc, const float* __restrict__ sh, const float* __restrict__ style,
                                      float* __restrict__ a, float* __restrict__ bq, int B, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx % C;
    const float s0 = style[(size_t)b * 2 * C + c] + 1.f, s1 = style[(size_t)b * 2 * C + C + c];
    a[idx] = sc[idx] * s0; bq[idx] = sh[idx] * s0 + s1;
}

// ------------------------------------------------------------------ PGGAN pixel norm (NHWC)
// y[p,:] = x[p,:] / sqrt(mean_c x[p,c]^2 + eps)    (model/pggan/pggan_generator.py:207-216)
// LPP lanes cooperate on one pixel with 16-byte accesses.
template <typename T, int NJ>
__global__ __launch_bounds__(256) void pixelnorm_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, long npix, int C, int lpp, float eps) {
    constexpr int EP = Elem<T>::PER16;
    const int lane = threadIdx.x & 63;
    const int ppw = 64 / lpp, sub = lane / lpp, li = lane % lpp;
    const long p = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ppw + sub;
    const bool ok = p < npix;
    float f[NJ][EP];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        if (ok) unpack16(*(const uint4*)(x + (size_t)p * C + (j * lpp + li) * EP), f[j], (T*)nullptr);
        else {
#pragma unroll
            for (int e = 0; e < EP; e++) f[j][e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < EP; e++) s += f[j][e] * f[j][e];
    }
    for (int m = lpp >> 1; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
    const float r = rsqrtf(s / (float)C + eps);
    if (ok) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
#pragma unroll
            for (int e = 0; e < EP; e++) f[j][e] *= r;
            *(uint4*)(y + (size_t)p * C + (j * lpp + li) * EP) = pack16(f[j], (T*)nullptr);
        }
    }
}

// backward of the pixel norm: with r = rsqrt(mean_c x^2 + eps), yhat = x*r:  gx = r * (gy - yhat * mean_c(gy*yhat))
template <typename T, int NJ>
__global__ __launch_bounds__(256) void pixelnorm_nhwc_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ x, T* __restrict__ gx,
                                                                  long npix, int C, int lpp, float eps) {
    constexpr int EP = Elem<T>::PER16;
    const int lane = threadIdx.x & 63;
    const int ppw = 64 / lpp, sub = lane / lpp, li = lane % lpp;
    const long p = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * ppw + sub;
    const bool ok = p < npix;
    float f[NJ][EP], g[NJ][EP];
    float s = 0.f, d = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        if (ok) {
            unpack16(*(const uint4*)(x + (size_t)p * C + (j * lpp + li) * EP), f[j], (T*)nullptr);
            unpack16(*(const uint4*)(gy + (size_t)p * C + (j * lpp + li) * EP), g[j], (T*)nullptr);
        } else {
#pragma unroll
            for (int e = 0; e < EP; e++) { f[j][e] = 0.f; g[j][e] = 0.f; }
        }
#pragma unroll
        for (int e = 0; e < EP; e++) { s += f[j][e] * f[j][e]; d += f[j][e] * g[j][e]; }
    }
    for (int m = lpp >> 1; m > 0; m >>= 1) { s += __shfl_xor(s, m, 64); d += __shfl_xor(d, m, 64); }
    const float r = rsqrtf(s / (float)C + eps);
    const float k = d * r * r * r / (float)C;          // yhat * mean(gy*yhat) * r = x * (sum gy*x) * r^3 / C
    if (ok) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
#pragma unroll
            for (int e = 0; e < EP; e++) g[j][e] = g[j][e] * r - f[j][e] * k;
            *(uint4*)(gx + (size_t)p * C + (j * lpp + li) * EP) = pack16(g[j], (T*)nullptr);
        }
    }
}

// =================================================================== C ABI

extern "C" int dge_fromrgb2(const float* img, const float* w, const float* bias, void* y, float* stats, float* img4, int B, int HW,
                            int C, int dtype, hipStream_t s);
extern "C" int dge_fromrgb(const float* img, const float* w, const float* bias, void* y, float* stats, int B, int HW,
                           int C, int dtype, hipStream_t s) {
    return dge_fromrgb2(img, w, bias, y, stats, nullptr, B, HW, C, dtype, s);
}
extern "C" int dge_fromrgb2(const float* img, const float* w, const float* bias, void* y, float* stats, float* img4, int B, int HW,
                            int C, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0 && C / ep <= 256 && 256 % (C / ep) == 0, "fromrgb: unsupported channel count %d", C);
    const int ppi = 256 / (C / ep);
    dim3 grid(dge_stream_grid(HW, ppi, B), B);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(fromrgb_kernel<bf16_t>, grid, dim3(256), 0, s, img, w, bias, (bf16_t*)y, stats, HW, C, (float4*)img4);
    else hipLaunchKernelGGL(fromrgb_kernel<float>, grid, dim3(256), 0, s, img, w, bias, (float*)y, stats, HW, C, (float4*)img4);
    DGE_LAUNCH_CHECK("fromrgb");
    return 0;
}

extern "C" int dge_stats_finalize_slots(const float* stats, int nslot, float* musig, float* sc, float* sh, int B, int C, int npix,
                                        float eps, hipStream_t s) {
    DGE_CHECK(nslot >= 1, "stats_finalize: nslot %d", nslot);
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((B * C + 63) / 64), dim3(64), 0, s, stats, musig, sc, sh, B, C,
                       1.0f / (float)npix, eps, nslot);
    DGE_LAUNCH_CHECK("stats_finalize");
    return 0;
}
extern "C" int dge_stats_finalize(const float* stats, float* musig, float* sc, float* sh, int B, int C, int npix, float eps,
                                  hipStream_t s) {
    return dge_stats_finalize_slots(stats, 1, musig, sc, sh, B, C, npix, eps, s);
}

static int blend_launch(const void* x, const void* z, void* y, const float* sc, const float* sh, float* stats, int B,
                        int OH, int OW, int C, int pool, float alpha, float beta, unsigned* mask, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0 && C / ep <= 256 && 256 % (C / ep) == 0, "blend: unsupported channel count %d", C);
    const int ppi = 256 / (C / ep);
    dim3 grid(dge_stream_grid(OH * OW, ppi, B), B);
    if (dtype == DGE_BF16)
        hipLaunchKernelGGL(blend_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)z, (bf16_t*)y, sc, sh, stats, OH, OW, C, pool, alpha, beta, mask);
    else
        hipLaunchKernelGGL(blend_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)z, (float*)y, sc, sh, stats, OH, OW, C, pool, alpha, beta, mask);
    DGE_LAUNCH_CHECK("blend");
    return 0;
}
extern "C" int dge_blend(const void* x, const void* z, void* y, const float* sc, const float* sh, float* stats, int B,
                         int OH, int OW, int C, int pool, float alpha, float beta, int dtype, hipStream_t s) {
    return blend_launch(x, z, y, sc, sh, stats, B, OH, OW, C, pool, alpha, beta, nullptr, dtype, s);
}
// the pooling blend, also leaving the SIGNS of its full-resolution input: mask [B, OH*OW, C/ep] words, bit q*ep + e = (x at the q-th
// of the 2x2 children, channel e of the chunk) > 0.  The activation backward of that input reads 1 bit per element instead of the tensor.
extern "C" int dge_blend_pool_mask(const void* x, const void* z, void* y, float* stats, unsigned* mask, int B, int OH, int OW, int C,
                                   float alpha, float beta, int dtype, hipStream_t s) {
    DGE_CHECK(mask, "blend_pool_mask: null mask");
    return blend_launch(x, z, y, nullptr, nullptr, stats, B, OH, OW, C, 1, alpha, beta, mask, dtype, s);
}

extern "C" int dge_blur_noise_act(const void* x, const float* noise, const float* noise_w, const float* bias, void* y, float* stats,
                                  int B, int H, int W, int C, int do_blur, int noise_batch, int act, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0 && C / ep <= 256 && 256 % (C / ep) == 0, "blur_noise_act: unsupported channel count %d", C);
    const int ppi = 256 / (C / ep);
    dim3 grid(dge_stream_grid(H * W, ppi, B), B);
    const int nbs = noise_batch > 1 ? H * W : 0;
    if (dtype == DGE_BF16)
        hipLaunchKernelGGL(blur_noise_act_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, noise, noise_w, bias, (bf16_t*)y, stats, H, W, C, do_blur, nbs, act);
    else
        hipLaunchKernelGGL(blur_noise_act_kernel<float>, grid, dim3(256), 0, s, (const float*)x, noise, noise_w, bias, (float*)y, stats, H, W, C, do_blur, nbs, act);
    DGE_LAUNCH_CHECK("blur_noise_act");
    return 0;
}

extern "C" int dge_affine_compose(const float* sc, const float* sh, const float* style, float* a, float* b, int B, int C, hipStream_t s) {
    hipLaunchKernelGGL(affine_compose_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, sc, sh, style, a, b, B, C);
    DGE_LAUNCH_CHECK("affine_compose");
    return 0;
}

template <typename T>
static int pixelnorm_nhwc_launch(const void* x, void* y, long npix, int C, float eps, hipStream_t s) {
    constexpr int EP = Elem<T>::PER16;
    const int chunks = C / EP;
    const int lpp = chunks >= 64 ? 64 : chunks;
    const int nj = chunks / lpp;
    const long ppb = 4 * (64 / lpp);
    dim3 grid((unsigned)((npix + ppb - 1) / ppb));
#define PN(NJ) hipLaunchKernelGGL((pixelnorm_nhwc_kernel<T, NJ>), grid, dim3(256), 0, s, (const T*)x, (T*)y, npix, C, lpp, eps)
    if (nj == 1) PN(1); else if (nj == 2) PN(2); else { dge_set_error("pixelnorm_nhwc: unsupported C=%d", C); return -1; }
#undef PN
    return 0;
}
extern "C" int dge_pixelnorm_nhwc(const void* x, void* y, long npix, int C, float eps, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0 && ((C / ep) & (C / ep - 1)) == 0, "pixelnorm_nhwc: C=%d must be a power-of-two multiple of %d", C, ep);
    const int rc = dtype == DGE_BF16 ? pixelnorm_nhwc_launch<bf16_t>(x, y, npix, C, eps, s) : pixelnorm_nhwc_launch<float>(x, y, npix, C, eps, s);
    if (rc) return rc;
    DGE_LAUNCH_CHECK("pixelnorm_nhwc");
    return 0;
}

template <typename T>
static int pixelnorm_nhwc_bwd_launch(const void* gy, const void* x, void* gx, long npix, int C, float eps, hipStream_t s) {
    constexpr int EP = Elem<T>::PER16;
    const int chunks = C / EP;
    const int lpp = chunks >= 64 ? 64 : chunks;
    const int nj = chunks / lpp;
    const long ppb = 4 * (64 / lpp);
    dim3 grid((unsigned)((npix + ppb - 1) / ppb));
#define PNB(NJ) hipLaunchKernelGGL((pixelnorm_nhwc_bwd_kernel<T, NJ>), grid, dim3(256), 0, s, (const T*)gy, (const T*)x, (T*)gx, npix, C, lpp, eps)
    if (nj == 1) PNB(1); else if (nj == 2) PNB(2); else { dge_set_error("pixelnorm_nhwc_bwd: unsupported C=%d", C); return -1; }
#undef PNB
    return 0;
}
extern "C" int dge_pixelnorm_nhwc_bwd(const void* gy, const void* x, void* gx, long npix, int C, float eps, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0 && ((C / ep) & (C / ep - 1)) == 0, "pixelnorm_nhwc_bwd: C=%d must be a power-of-two multiple of %d", C, ep);
    const int rc = dtype == DGE_BF16 ? pixelnorm_nhwc_bwd_launch<bf16_t>(gy, x, gx, npix, C, eps, s)
                                      : pixelnorm_nhwc_bwd_launch<float>(gy, x, gx, npix, C, eps, s);
    if (rc) return rc;
    DGE_LAUNCH_CHECK("pixelnorm_nhwc_bwd");
    return 0;
}
