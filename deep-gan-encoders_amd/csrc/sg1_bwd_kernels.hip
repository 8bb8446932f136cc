// StyleGAN1 synthesis data-gradient glue (reference model/stylegan1/net.py DecodeBlock.forward
// :141-169 differentiated w.r.t. the styles): coefficients of the instance-norm + style_mod
// backward, per-(b,c) dot statistics, and the adjoint of the nearest x2 upsample.  The heavy
// parts reuse conv_igemm (data gradients), in_bwd_kernel (instance-norm + leaky-relu backward)
// and blur_noise_act (the 3x3 blur is self-adjoint).
#include "common.h"
#include "../../include/dge_hip.h"

#define CHAN_OK(C, ep) ((C) % (ep) == 0 && (C) / (ep) <= 256 && 256 % ((C) / (ep)) == 0)

// u = style_mod(IN(y)) = (y*r + s)*(1+s0) + s1 with r = sc, s = sh = -mu*r, style = [s0 | s1].
// Given dots = (sum_p g_u*y, sum_p g_u):
//   g_s0 = sum g_u*yhat = r*S2 + s*S1,  g_s1 = S1,
//   g_y  = a*(g_u - mean(g_u) - yhat*mean(g_u*yhat)),  a = r*(1+s0)
//        = A*g_u + Bc*y + Cc  with A = a, Bc = -a*r*m2, Cc = -a*m1 - a*m2*s.
__global__ void sg1_in_bwd_coef_kernel(const float* __restrict__ dots, const float* __restrict__ sc, const float* __restrict__ sh,
                                       const float* __restrict__ style, float* __restrict__ coef, float* __restrict__ gstyle,
                                       int B, int C, float inv_n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx % C;
    const float r = sc[idx], s = sh[idx];
    const float S2 = dots[(size_t)idx * 2], S1 = dots[(size_t)idx * 2 + 1];
    const float D = r * S2 + s * S1;
    const float m1 = S1 * inv_n, m2 = D * inv_n;
    const float a = r * (1.f + style[(size_t)b * 2 * C + c]);
    coef[(size_t)idx * 3 + 0] = a;
    coef[(size_t)idx * 3 + 1] = -a * r * m2;
    coef[(size_t)idx * 3 + 2] = -a * m1 - a * m2 * s;
    gstyle[(size_t)b * 2 * C + c] = D;
    gstyle[(size_t)b * 2 * C + C + c] = S1;
}

// stats[b,c,:] (pre-zeroed) += (sum_p g*x, sum_p g)
template <typename T>
__global__ __launch_bounds__(256) void dot_stats_kernel(const T* __restrict__ g, const T* __restrict__ x, float* __restrict__ stats,
                                                         int HW, int C) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 2 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    float s[2][EP];
#pragma unroll
    for (int e = 0; e < EP; e++) s[0][e] = s[1][e] = 0.f;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            const size_t o = ((size_t)b * HW + p) * C + chunk * EP;
            float gv[EP], xv[EP];
            unpack16(*(const uint4*)(g + o), gv, (T*)nullptr);
            unpack16(*(const uint4*)(x + o), xv, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < EP; e++) { s[0][e] += gv[e] * xv[e]; s[1][e] += gv[e]; }
        }
    }
    block_chan_flush<EP, 2>(s, cpt, ppi, stats + (size_t)b * C * 2, C, red);
}

// adjoint of upscale2d (nearest x2, net.py:37-43): glow[b,y,x,c] = sum of the 2x2 block of ghi, with the
// dot statistics of the result against the low-resolution activation x (for the next instance-norm backward)
template <typename T>
__global__ __launch_bounds__(256) void nearest_up2_bwd_kernel(const T* __restrict__ ghi, const T* __restrict__ x, T* __restrict__ glow,
                                                               float* __restrict__ stats, int H, int W, int C) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 2 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    const int HW = H * W;
    float s[2][EP];
#pragma unroll
    for (int e = 0; e < EP; e++) s[0][e] = s[1][e] = 0.f;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            const int y = p / W, xq = p % W;
            float acc[EP];
#pragma unroll
            for (int e = 0; e < EP; e++) acc[e] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float gv[EP];
                unpack16(*(const uint4*)(ghi + (((size_t)b * 2 * H + 2 * y + (q >> 1)) * (2 * W) + 2 * xq + (q & 1)) * C + chunk * EP), gv, (T*)nullptr);
#pragma unroll
                for (int e = 0; e < EP; e++) acc[e] += gv[e];
            }
            const size_t o = ((size_t)b * HW + p) * C + chunk * EP;
            const uint4 packed = pack16(acc, (T*)nullptr);
            *(uint4*)(glow + o) = packed;
            if (stats) {
                float gq[EP], xv[EP];
                unpack16(packed, gq, (T*)nullptr);          // statistics of the value the consumer will read
                unpack16(*(const uint4*)(x + o), xv, (T*)nullptr);
#pragma unroll
                for (int e = 0; e < EP; e++) { s[0][e] += gq[e] * xv[e]; s[1][e] += gq[e]; }
            }
        }
    }
    if (stats) block_chan_flush<EP, 2>(s, cpt, ppi, stats + (size_t)b * C * 2, C, red);
}

extern "C" int dge_sg1_in_bwd_coef(const float* dots, const float* sc, const float* sh, const float* style, float* coef,
                                   float* gstyle, int B, int C, int npix, hipStream_t s) {
    hipLaunchKernelGGL(sg1_in_bwd_coef_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, dots, sc, sh, style, coef, gstyle, B, C,
                       1.0f / (float)npix);
    DGE_LAUNCH_CHECK("sg1_in_bwd_coef");
    return 0;
}

extern "C" int dge_dot_stats(const void* g, const void* x, float* stats, int B, int HW, int C, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(CHAN_OK(C, ep), "dot_stats: unsupported channel count %d", C);
    dim3 grid(dge_stream_grid(HW, 256 / (C / ep), B), B);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(dot_stats_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)x, stats, HW, C);
    else hipLaunchKernelGGL(dot_stats_kernel<float>, grid, dim3(256), 0, s, (const float*)g, (const float*)x, stats, HW, C);
    DGE_LAUNCH_CHECK("dot_stats");
    return 0;
}

extern "C" int dge_nearest_up2_bwd(const void* ghi, const void* x, void* glow, float* stats, int B, int H, int W, int C, int dtype,
                                   hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(CHAN_OK(C, ep), "nearest_up2_bwd: unsupported channel count %d", C);
    DGE_CHECK(!stats || x, "nearest_up2_bwd: statistics need the low-resolution activation");
    dim3 grid(dge_stream_grid(H * W, 256 / (C / ep), B), B);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(nearest_up2_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)ghi, (const bf16_t*)x, (bf16_t*)glow, stats, H, W, C);
    else hipLaunchKernelGGL(nearest_up2_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)ghi, (const float*)x, (float*)glow, stats, H, W, C);
    DGE_LAUNCH_CHECK("nearest_up2_bwd");
    return 0;
}
