// Backward kernels of the encoder E.BE (reference model/E/E.py:50-85 differentiated):
// conv weight gradient (MFMA, split over pixel tiles, f32 atomics), instance-norm / statistics
// backward, activation backward with bias / noise-weight reductions, FromRGB and dense-layer
// weight gradients.  The conv data gradients reuse conv_igemm (DGE_PACK_DGRAD).
#include "common.h"
#include "../../include/dge_hip.h"

// ------------------------------------------------------------------ conv weight gradient
// dW[o][i][tap] += sum_{b,p} g[b,p,o] * Xn[b,p+tap,i],  Xn = X*sc[b,i] + sh[b,i] inside the image, 0 outside.
// GEMM view: M = o (32 per block), N = i (32 per block), K = pixels.  One MFMA K-step = one 16-pixel
// tile row; activations stay NHWC in LDS and K-contiguous fragments are gathered with 16-bit LDS
// reads (bf16) or single dword reads (f32).
template <typename T> struct WgMma;
template <> struct WgMma<bf16_t> {
    static constexpr int KSTEP = 16;                  // pixels per MFMA
    // lane (m = l&31, kg = l>>5) gathers 8 pixels kg*8..kg*8+7 of channel m
    __device__ static __forceinline__ uint4 gather(const bf16_t* base, int pix_stride_elems, int kg) {
        const bf16_t* p = base + (size_t)(kg * 8) * pix_stride_elems;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            w[j] = (uint32_t)p[(2 * j) * pix_stride_elems] | ((uint32_t)p[(2 * j + 1) * pix_stride_elems] << 16);
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ static __forceinline__ void mma(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&a, *(const bf16x8_t*)&b, c, 0, 0, 0);
    }
};
template <> struct WgMma<float> {
    static constexpr int KSTEP = 16;                  // processed as 8 x (32x32x2)
    __device__ static __forceinline__ void run16(const float* abase, const float* bbase, int astr, int bstr, int kg, f32x16_t& c) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(abase[(size_t)(2 * j + kg) * astr], bbase[(size_t)(2 * j + kg) * bstr], c, 0, 0, 0);
    }
};

template <typename T, int KS>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const T* __restrict__ g, const T* __restrict__ X,
                                                          const float* __restrict__ sc, const float* __restrict__ sh,
                                                          float* __restrict__ dW, int B, int H, int W, int Co, int Ci,
                                                          int tiles_x, int tiles_y, int ntile_groups) {
    constexpr int TH = 8, TW = 16, HALO = KS / 2, HH = TH + 2 * HALO, HW = TW + 2 * HALO;
    constexpr int EP = Elem<T>::PER16;
    constexpr int CSTR = 32 + EP;                       // elements per pixel row in LDS (padded)
    constexpr int NTAP = KS * KS;
    constexpr int TPW = (NTAP + 3) / 4;                 // taps per wave (KS=3: 3,2,2,2)
    __shared__ __attribute__((aligned(16))) T lg[TH * TW * CSTR];
    __shared__ __attribute__((aligned(16))) T lx[HH * HW * CSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_it = (Ci + 31) / 32;
    const int o0 = (blockIdx.x / n_it) * 32, i0 = (blockIdx.x % n_it) * 32;
    f32x16_t acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    const int ntiles = tiles_x * tiles_y * B;
    constexpr int CPR = 32 / EP;                        // 16-byte chunks per 32-channel row
    for (int tile = blockIdx.y; tile < ntiles; tile += ntile_groups) {
        const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx_i * TW, y0 = ty_i * TH;
        __syncthreads();
        for (int idx = tid; idx < TH * TW * CPR; idx += 256) {       // gradient tile
            const int c = idx % CPR, pix = idx / CPR, gy = y0 + pix / TW, gx = x0 + pix % TW;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (gy < H && gx < W && o0 + c * EP < Co) v = *(const uint4*)(g + ((size_t)(b * H + gy) * W + gx) * Co + o0 + c * EP);
            *(uint4*)(lg + pix * CSTR + c * EP) = v;
        }
        for (int idx = tid; idx < HH * HW * CPR; idx += 256) {       // normalised input halo tile
            const int c = idx % CPR, pix = idx / CPR, gy = y0 + pix / HW - HALO, gx = x0 + pix % HW - HALO;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W && i0 + c * EP < Ci) {
                v = *(const uint4*)(X + ((size_t)(b * H + gy) * W + gx) * Ci + i0 + c * EP);
                if (sc) {
                    float f[EP];
                    unpack16(v, f, (T*)nullptr);
#pragma unroll
                    for (int e = 0; e < EP; e++) {
                        const int ci = b * Ci + i0 + c * EP + e;
                        f[e] = f[e] * sc[ci] + sh[ci];
                    }
                    v = pack16(f, (T*)nullptr);
                }
            }
            *(uint4*)(lx + pix * CSTR + c * EP) = v;
        }
        __syncthreads();
        const int m = lane & 31, kg = lane >> 5;
        if (KS == 3) {
#pragma unroll
            for (int t = 0; t < TPW; t++) {
                const int tap = wave + 4 * t;
                if (tap < NTAP) {
                    const int dy = tap / 3, dx = tap % 3;
#pragma unroll
                    for (int ks = 0; ks < TH; ks++) {                  // one 16-pixel row per MFMA K-step
                        const T* ab = lg + (ks * TW) * CSTR + m;
                        const T* bb = lx + ((ks + dy) * HW + dx) * CSTR + m;
                        if constexpr (sizeof(T) == 2) {
                            const uint4 a = WgMma<bf16_t>::gather((const bf16_t*)ab, CSTR, kg);
                            const uint4 bq = WgMma<bf16_t>::gather((const bf16_t*)bb, CSTR, kg);
                            WgMma<bf16_t>::mma(a, bq, acc[t]);
                        } else {
                            WgMma<float>::run16((const float*)ab, (const float*)bb, CSTR, CSTR, kg, acc[t]);
                        }
                    }
                }
            }
        } else {                                                       // 1x1: waves split the pixel rows
#pragma unroll
            for (int q = 0; q < TH / 4; q++) {
                const int ks = wave + 4 * q;
                const T* ab = lg + (ks * TW) * CSTR + m;
                const T* bb = lx + (ks * HW) * CSTR + m;
                if constexpr (sizeof(T) == 2) {
                    const uint4 a = WgMma<bf16_t>::gather((const bf16_t*)ab, CSTR, kg);
                    const uint4 bq = WgMma<bf16_t>::gather((const bf16_t*)bb, CSTR, kg);
                    WgMma<bf16_t>::mma(a, bq, acc[0]);
                } else {
                    WgMma<float>::run16((const float*)ab, (const float*)bb, CSTR, CSTR, kg, acc[0]);
                }
            }
        }
    }
    // flush: D[m = o][n = i]
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const int tap = KS == 3 ? wave + 4 * t : 0;
        if (tap >= NTAP) continue;
        const int i = i0 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (o < Co && i < Ci) atomicAdd(dW + ((size_t)o * Ci + i) * NTAP + tap, acc[t][r]);
        }
    }
}

// ------------------------------------------------------------------ activation backward (+pool adjoint)
// a = lrelu(pre) saved.  g_pre[b,p,c] = scale * g_up[b,q(p),c] * lrelu'(a)   (q = p/2 per axis when pool)
// red[c,:] (pre-zeroed, summed over the whole batch) += { sum g_pre, sum g_pre*noise[b,p] }
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ gup, const T* __restrict__ a,
                                                       const float* __restrict__ noise, T* __restrict__ gpre,
                                                       float* __restrict__ red_out, int H, int W, int C, int pool, float scale, float slope) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 2 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    const int HW = H * W, UW = pool ? W / 2 : W, UHW = pool ? HW / 4 : HW;
    float s[2][EP];
#pragma unroll
    for (int e = 0; e < EP; e++) s[0][e] = s[1][e] = 0.f;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            const int q = pool ? (p / W / 2) * UW + (p % W) / 2 : p;
            float g[EP], av[EP];
            unpack16(*(const uint4*)(gup + ((size_t)b * UHW + q) * C + chunk * EP), g, (T*)nullptr);
            unpack16(*(const uint4*)(a + ((size_t)b * HW + p) * C + chunk * EP), av, (T*)nullptr);
            const float nz = noise ? noise[(size_t)b * HW + p] : 0.f;
#pragma unroll
            for (int e = 0; e < EP; e++) {
                g[e] = scale * g[e] * (av[e] > 0.f ? 1.f : slope);
                s[0][e] += g[e]; s[1][e] += g[e] * nz;
            }
            *(uint4*)(gpre + ((size_t)b * HW + p) * C + chunk * EP) = pack16(g, (T*)nullptr);
        }
    }
    if (red_out) block_chan_flush<EP, 2>(s, cpt, ppi, red_out, C, red);
}

// ------------------------------------------------------------------ instance-norm + statistics backward
// Coefficients of g_X = A*g_y + Bc*X + Cc for y = (X-mu)*r, with the extra gradients g_mu, g_sigma of
// the (mean, std) outputs (E.py:51-53): see DESIGN.md.  dots[b,c] = (sum g_y*X, sum g_y) from the dgrad conv
// epilogue (may be null when y has no consumer), gms [B,2C] = [g_mu | g_sigma].
__global__ void in_bwd_coef_kernel(const float* __restrict__ dots, const float* __restrict__ gms, const float* __restrict__ musig,
                                   const float* __restrict__ sc, const float* __restrict__ sh, float* __restrict__ coef,
                                   int B, int C, float inv_n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx % C;
    const float r = sc[idx], s = sh[idx];
    const float S2 = dots ? dots[(size_t)idx * 2] : 0.f, S1 = dots ? dots[(size_t)idx * 2 + 1] : 0.f;
    const float m1 = S1 * inv_n, m2 = (r * S2 + s * S1) * inv_n;
    const float mu = musig[(size_t)b * 2 * C + c], sg = musig[(size_t)b * 2 * C + C + c];
    const float gmu = gms ? gms[(size_t)b * 2 * C + c] : 0.f, gsg = gms ? gms[(size_t)b * 2 * C + C + c] : 0.f;
    const float k = sg > 0.f ? gsg * inv_n / sg : 0.f;
    coef[(size_t)idx * 3 + 0] = r;
    coef[(size_t)idx * 3 + 1] = -r * r * m2 + k;
    coef[(size_t)idx * 3 + 2] = -r * m1 - r * m2 * s + gmu * inv_n - k * mu;
}

// g_X = A*g_y + Bc*X + Cc + extra_scale * extra[q(p)]   then, when act != 0 (X = lrelu(pre)):
// g_pre = g_X * lrelu'(X) with the bias / noise-weight reductions as in act_bwd_kernel.
template <typename T>
__global__ __launch_bounds__(256) void in_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ X, const float* __restrict__ coef,
                                                      const T* __restrict__ extra, const float* __restrict__ noise,
                                                      T* __restrict__ gout, float* __restrict__ red_out, int H, int W, int C,
                                                      int extra_pool, float extra_scale, int act) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 2 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    const int HW = H * W, UW = extra_pool ? W / 2 : W, UHW = extra_pool ? HW / 4 : HW;
    float s[2][EP], A[EP], Bc[EP], Cc[EP];
#pragma unroll
    for (int e = 0; e < EP; e++) {
        s[0][e] = s[1][e] = 0.f;
        const size_t ci = ((size_t)b * C + chunk * EP + e) * 3;
        A[e] = coef[ci]; Bc[e] = coef[ci + 1]; Cc[e] = coef[ci + 2];
    }
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            const size_t o = ((size_t)b * HW + p) * C + chunk * EP;
            float g[EP], xv[EP];
            if (gy) unpack16(*(const uint4*)(gy + o), g, (T*)nullptr);
            else {
#pragma unroll
                for (int e = 0; e < EP; e++) g[e] = 0.f;
            }
            unpack16(*(const uint4*)(X + o), xv, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < EP; e++) g[e] = A[e] * g[e] + Bc[e] * xv[e] + Cc[e];
            if (extra) {
                const int q = extra_pool ? (p / W / 2) * UW + (p % W) / 2 : p;
                float ex[EP];
                unpack16(*(const uint4*)(extra + ((size_t)b * UHW + q) * C + chunk * EP), ex, (T*)nullptr);
#pragma unroll
                for (int e = 0; e < EP; e++) g[e] += extra_scale * ex[e];
            }
            if (act) {
                const float nz = noise ? noise[(size_t)b * HW + p] : 0.f;
#pragma unroll
                for (int e = 0; e < EP; e++) {
                    g[e] *= (xv[e] > 0.f ? 1.f : 0.2f);
                    s[0][e] += g[e]; s[1][e] += g[e] * nz;
                }
            }
            *(uint4*)(gout + o) = pack16(g, (T*)nullptr);
        }
    }
    if (red_out) block_chan_flush<EP, 2>(s, cpt, ppi, red_out, C, red);
}

// per-channel sum over batch and pixels of an NHWC tensor: out[c] += scale * sum x[b,p,c]
template <typename T>
__global__ __launch_bounds__(256) void chan_sum_kernel(const T* __restrict__ x, float* __restrict__ out, int HW, int C, float scale) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    float s[1][EP];
#pragma unroll
    for (int e = 0; e < EP; e++) s[0][e] = 0.f;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            float f[EP];
            unpack16(*(const uint4*)(x + ((size_t)b * HW + p) * C + chunk * EP), f, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < EP; e++) s[0][e] += scale * f[e];
        }
    }
    block_chan_flush<EP, 1>(s, cpt, ppi, out, C, red);
}

// FromRGB backward: x0 = lrelu(W img + b).  g_pre = g_x0*lrelu'(x0);  out[o][0..2] += sum g_pre*img[c], out[o][3] += sum g_pre
template <typename T>
__global__ __launch_bounds__(256) void fromrgb_bwd_kernel(const T* __restrict__ gx, const T* __restrict__ x0,
                                                           const float* __restrict__ img, float* __restrict__ out, int HW, int C) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 4 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    float s[4][EP];
#pragma unroll
    for (int e = 0; e < EP; e++) s[0][e] = s[1][e] = s[2][e] = s[3][e] = 0.f;
    const float* ib = img + (size_t)b * 3 * HW;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            const size_t o = ((size_t)b * HW + p) * C + chunk * EP;
            float g[EP], xv[EP];
            unpack16(*(const uint4*)(gx + o), g, (T*)nullptr);
            unpack16(*(const uint4*)(x0 + o), xv, (T*)nullptr);
            const float r = ib[p], gr = ib[HW + p], bl = ib[2 * HW + p];
#pragma unroll
            for (int e = 0; e < EP; e++) {
                const float gp = g[e] * (xv[e] > 0.f ? 1.f : 0.2f);
                s[0][e] += gp * r; s[1][e] += gp * gr; s[2][e] += gp * bl; s[3][e] += gp;
            }
        }
    }
    block_chan_flush<EP, 4>(s, cpt, ppi, out, C, red);
}

// dense layer parameter gradients: gW[o][i] (+)= sum_b gy[b][o]*x[b][i];  gb[o] (+)= sum_b gy[b][o]
__global__ void dense_wgrad_kernel(const float* __restrict__ gy, int ldgy, const float* __restrict__ x, int ldx,
                                   float* __restrict__ gW, float* __restrict__ gb, int B, int O, int I, int accumulate) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= (long)O * I) return;
    const int o = idx / I, i = idx % I;
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; b++) { const float g = gy[(size_t)b * ldgy + o]; s += g * x[(size_t)b * ldx + i]; sb += g; }
    gW[idx] = accumulate ? gW[idx] + s : s;
    if (gb && i == 0) gb[o] = accumulate ? gb[o] + sb : sb;
}

// =================================================================== C ABI
static int sgrid(int hw, int ppi) { int g = (hw + ppi - 1) / ppi; return g > 1024 ? 1024 : (g < 1 ? 1 : g); }
#define CHAN_OK(C, ep) ((C) % (ep) == 0 && (C) / (ep) <= 256 && 256 % ((C) / (ep)) == 0)

extern "C" int dge_conv_wgrad(const void* g, const void* x, const float* in_scale, const float* in_shift, float* dw, int B, int H,
                              int W, int cout, int cin, int ksize, int dtype, hipStream_t s) {
    DGE_CHECK(ksize == 1 || ksize == 3, "conv_wgrad: ksize %d unsupported", ksize);
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(cout % ep == 0 && cin % ep == 0, "conv_wgrad: channels must be multiples of %d", ep);
    DGE_CHECK((in_scale == nullptr) == (in_shift == nullptr), "conv_wgrad: in_scale and in_shift go together");
    const int tx = (W + 15) / 16, ty = (H + 7) / 8;
    const int ntiles = tx * ty * B;
    const int noi = ((cout + 31) / 32) * ((cin + 31) / 32);
    int groups = 2048 / noi; if (groups < 1) groups = 1; if (groups > ntiles) groups = ntiles;
    dim3 grid(noi, groups);
#define WG(T, KS) hipLaunchKernelGGL((conv_wgrad_kernel<T, KS>), grid, dim3(256), 0, s, (const T*)g, (const T*)x, in_scale, in_shift, dw, B, H, W, cout, cin, tx, ty, groups)
    if (dtype == DGE_BF16) { if (ksize == 3) WG(bf16_t, 3); else WG(bf16_t, 1); }
    else { if (ksize == 3) WG(float, 3); else WG(float, 1); }
#undef WG
    DGE_LAUNCH_CHECK("conv_wgrad");
    return 0;
}

extern "C" int dge_act_bwd(const void* gup, const void* a, const float* noise, void* gpre, float* red, int B, int H, int W, int C,
                           int pool, float scale, float slope, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(CHAN_OK(C, ep), "act_bwd: unsupported channel count %d", C);
    dim3 grid(sgrid(H * W, 256 / (C / ep)), B);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)gup, (const bf16_t*)a, noise, (bf16_t*)gpre, red, H, W, C, pool, scale, slope);
    else hipLaunchKernelGGL(act_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)gup, (const float*)a, noise, (float*)gpre, red, H, W, C, pool, scale, slope);
    DGE_LAUNCH_CHECK("act_bwd");
    return 0;
}

extern "C" int dge_in_bwd_coef(const float* dots, const float* gms, const float* musig, const float* sc, const float* sh,
                               float* coef, int B, int C, int npix, hipStream_t s) {
    hipLaunchKernelGGL(in_bwd_coef_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, dots, gms, musig, sc, sh, coef, B, C, 1.0f / (float)npix);
    DGE_LAUNCH_CHECK("in_bwd_coef");
    return 0;
}

extern "C" int dge_in_bwd(const void* gy, const void* x, const float* coef, const void* extra, const float* noise, void* gout,
                          float* red, int B, int H, int W, int C, int extra_pool, float extra_scale, int act, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(CHAN_OK(C, ep), "in_bwd: unsupported channel count %d", C);
    dim3 grid(sgrid(H * W, 256 / (C / ep)), B);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(in_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)gy, (const bf16_t*)x, coef, (const bf16_t*)extra, noise, (bf16_t*)gout, red, H, W, C, extra_pool, extra_scale, act);
    else hipLaunchKernelGGL(in_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)gy, (const float*)x, coef, (const float*)extra, noise, (float*)gout, red, H, W, C, extra_pool, extra_scale, act);
    DGE_LAUNCH_CHECK("in_bwd");
    return 0;
}

extern "C" int dge_chan_sum(const void* x, float* out, int B, int HW, int C, float scale, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(CHAN_OK(C, ep), "chan_sum: unsupported channel count %d", C);
    dim3 grid(sgrid(HW, 256 / (C / ep)), B);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(chan_sum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, out, HW, C, scale);
    else hipLaunchKernelGGL(chan_sum_kernel<float>, grid, dim3(256), 0, s, (const float*)x, out, HW, C, scale);
    DGE_LAUNCH_CHECK("chan_sum");
    return 0;
}

extern "C" int dge_fromrgb_bwd(const void* gx, const void* x0, const float* img, float* out4, int B, int HW, int C, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(CHAN_OK(C, ep), "fromrgb_bwd: unsupported channel count %d", C);
    dim3 grid(sgrid(HW, 256 / (C / ep)), B);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(fromrgb_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)gx, (const bf16_t*)x0, img, out4, HW, C);
    else hipLaunchKernelGGL(fromrgb_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)gx, (const float*)x0, img, out4, HW, C);
    DGE_LAUNCH_CHECK("fromrgb_bwd");
    return 0;
}

extern "C" int dge_dense_wgrad(const float* gy, int ldgy, const float* x, int ldx, float* gw, float* gb, int B, int O, int I,
                               int accumulate, hipStream_t s) {
    const long n = (long)O * I;
    hipLaunchKernelGGL(dense_wgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gy, ldgy, x, ldx, gw, gb, B, O, I, accumulate);
    DGE_LAUNCH_CHECK("dense_wgrad");
    return 0;
}
