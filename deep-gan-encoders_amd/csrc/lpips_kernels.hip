// LPIPS (net='vgg') glue kernels around conv_igemm: input scaling + NHWC packing, 2x2 max
// pooling (forward / backward), and the perceptual head (channel-unit-normalise, squared
// difference, learned 1x1 `lin`, spatial mean) with its analytic gradient.
// Third-party algorithm (not under the reference tree): richzhang/PerceptualSimilarity v0.1 as
// called at training_utils.py:93 / E_align_s2.py:98 - structure restated from the published
// method; see oracle/lpips_ref.py for the CPU restatement and DESIGN.md for the parity caveat.
#include "common.h"
#include "../../include/dge_hip.h"

// img [B,3,h,w] f32 -> x [B,h,w,CP] T (CP >= 3, extra channels zero): (img - shift[c]) / scale[c]
template <typename T>
__global__ void lpips_prep_kernel(const float* __restrict__ img, T* __restrict__ x, int B, int HW, int CP,
                                  float s0, float s1, float s2, float i0, float i1, float i2) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= (long)B * HW) return;
    const int b = idx / HW, p = idx % HW;
    const float* ib = img + (size_t)b * 3 * HW;
    T* xp = x + (size_t)idx * CP;
    Elem<T>::st(xp + 0, (ib[p] - s0) * i0);
    Elem<T>::st(xp + 1, (ib[HW + p] - s1) * i1);
    Elem<T>::st(xp + 2, (ib[2 * HW + p] - s2) * i2);
    for (int c = 3; c < CP; c++) Elem<T>::st(xp + c, 0.f);
}
// adjoint: gimg[b,c,p] (+)= factor * gx[b,p,c] * inv_scale[c]
template <typename T>
__global__ void lpips_prep_bwd_kernel(const T* __restrict__ gx, float* __restrict__ gimg, int B, int HW, int CP,
                                      float i0, float i1, float i2, float factor, int accumulate) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= (long)B * HW) return;
    const int b = idx / HW, p = idx % HW;
    const T* xp = gx + (size_t)idx * CP;
    float* gb = gimg + (size_t)b * 3 * HW;
    const float v0 = Elem<T>::ld(xp) * i0 * factor, v1 = Elem<T>::ld(xp + 1) * i1 * factor, v2 = Elem<T>::ld(xp + 2) * i2 * factor;
    if (accumulate) { gb[p] += v0; gb[HW + p] += v1; gb[2 * HW + p] += v2; }
    else { gb[p] = v0; gb[HW + p] = v1; gb[2 * HW + p] = v2; }
}

// 2x2 max pooling, stride 2, floor (nn.MaxPool2d(2,2)); NHWC
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    constexpr int EP = Elem<T>::PER16;
    const int OH = H / 2, OW = W / 2, cpt = C / EP;
    const long n = (long)B * OH * OW * cpt;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= n) return;
    const int ch = idx % cpt; long r = idx / cpt; const int ox = r % OW; r /= OW; const int oy = r % OH; const int b = r / OH;
    const T* base = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + ch * EP;
    float f0[EP], f1[EP], f2[EP], f3[EP];
    unpack16(*(const uint4*)base, f0, (T*)nullptr);
    unpack16(*(const uint4*)(base + C), f1, (T*)nullptr);
    unpack16(*(const uint4*)(base + (size_t)W * C), f2, (T*)nullptr);
    unpack16(*(const uint4*)(base + (size_t)W * C + C), f3, (T*)nullptr);
#pragma unroll
    for (int e = 0; e < EP; e++) f0[e] = fmaxf(fmaxf(f0[e], f1[e]), fmaxf(f2[e], f3[e]));
    *(uint4*)(y + (((size_t)b * OH + oy) * OW + ox) * C + ch * EP) = pack16(f0, (T*)nullptr);
}
// gx[b,p,c] = (p is the first arg-max of its window ? gy[window] : 0) + addend[b,p,c]
// relu_mode (x = output of a ReLU): 1 = followed by that ReLU's backward (x > 0), 2 = by the guided form (x > 0 and g > 0)
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ x, const T* __restrict__ addend,
                                   T* __restrict__ gx, int B, int H, int W, int C, int relu_mode) {
    constexpr int EP = Elem<T>::PER16;
    const int OH = H / 2, OW = W / 2, cpt = C / EP;
    const long n = (long)B * H * W * cpt;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= n) return;
    const int ch = idx % cpt; long r = idx / cpt; const int px = r % W; r /= W; const int py = r % H; const int b = r / H;
    const size_t o = (((size_t)b * H + py) * W + px) * C + ch * EP;
    float out[EP];
#pragma unroll
    for (int e = 0; e < EP; e++) out[e] = 0.f;
    const int oy = py / 2, ox = px / 2;
    if (oy < OH && ox < OW) {
        const T* base = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + ch * EP;
        float f[4][EP], g[EP];
        unpack16(*(const uint4*)base, f[0], (T*)nullptr);
        unpack16(*(const uint4*)(base + C), f[1], (T*)nullptr);
        unpack16(*(const uint4*)(base + (size_t)W * C), f[2], (T*)nullptr);
        unpack16(*(const uint4*)(base + (size_t)W * C + C), f[3], (T*)nullptr);
        unpack16(*(const uint4*)(gy + (((size_t)b * OH + oy) * OW + ox) * C + ch * EP), g, (T*)nullptr);
        const int me = (py & 1) * 2 + (px & 1);
#pragma unroll
        for (int e = 0; e < EP; e++) {
            int am = 0; float mv = f[0][e];
#pragma unroll
            for (int q = 1; q < 4; q++) if (f[q][e] > mv) { mv = f[q][e]; am = q; }
            if (am == me) out[e] = g[e];
            // (the gradient lands on the arg-max only, whose value is mv)
            if (relu_mode && relu_mode != 3 && !(mv > 0.f && (relu_mode == 1 || out[e] > 0.f))) out[e] = 0.f;
        }
    }
    if (addend) {
        float a[EP];
        unpack16(*(const uint4*)(addend + o), a, (T*)nullptr);
#pragma unroll
        for (int e = 0; e < EP; e++) out[e] += a[e];
    }
    if (relu_mode == 3) {           // ReLU backward of the layer that produced x, on the complete gradient (pool route + addend)
        float xs[EP];
        unpack16(*(const uint4*)(x + o), xs, (T*)nullptr);
#pragma unroll
        for (int e = 0; e < EP; e++) out[e] = xs[e] > 0.f ? out[e] : 0.f;
    }
    *(uint4*)(gx + o) = pack16(out, (T*)nullptr);
}

// LPIPS head for one tap.  f: [2B,h,w,C] T (samples [0,B) = image a, [B,2B) = image b), lin [C] f32.
// val[b] (pre-zeroed) += (1/(h*w)) sum_p sum_c lin_c (n0_c - n1_c)^2,  n = f / (||f||_2 + 1e-10)
// g1 (optional, [B,h,w,C] T) = gscale * d val[b] / d f[B+b]
// LPP = min(64, C/EP) lanes cooperate on one pixel with 16-byte loads; a wave walks PPW pixels per
// iteration and a block walks a contiguous pixel range of ONE sample, so val gets one atomic per wave.
template <typename T, int NJ>
__global__ __launch_bounds__(256) void lpips_head_kernel(const T* __restrict__ f, const float* __restrict__ lin,
                                                          float* __restrict__ val, T* __restrict__ g1, int B, int HW, int C,
                                                          float gscale, int lpp, int pix_per_block) {
    constexpr int EP = Elem<T>::PER16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int ppw = 64 / lpp, sub = lane / lpp, li = lane % lpp;
    const int p_begin = blockIdx.x * pix_per_block, p_end = min(HW, p_begin + pix_per_block);
    float linv[NJ][EP];
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int e = 0; e < EP; e++) linv[j][e] = lin[(j * lpp + li) * EP + e];
    const float inv_hw = 1.f / (float)HW;
    float dacc = 0.f;
    for (int p0 = p_begin + wave * ppw; p0 < p_end; p0 += 4 * ppw) {
        const int p = p0 + sub;
        const bool ok = p < p_end;
        const size_t o0 = ((size_t)b * HW + (ok ? p : p_begin)) * C, o1 = ((size_t)(B + b) * HW + (ok ? p : p_begin)) * C;
        float a[NJ][EP], bv[NJ][EP];
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            unpack16(*(const uint4*)(f + o0 + (j * lpp + li) * EP), a[j], (T*)nullptr);
            unpack16(*(const uint4*)(f + o1 + (j * lpp + li) * EP), bv[j], (T*)nullptr);
#pragma unroll
            for (int e = 0; e < EP; e++) { s0 += a[j][e] * a[j][e]; s1 += bv[j][e] * bv[j][e]; }
        }
        for (int m = lpp >> 1; m > 0; m >>= 1) { s0 += __shfl_xor(s0, m, 64); s1 += __shfl_xor(s1, m, 64); }
        const float n1 = sqrtf(s1), r0 = 1.f / (sqrtf(s0) + 1e-10f), r1 = 1.f / (n1 + 1e-10f);
        float d = 0.f, dot = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int e = 0; e < EP; e++) {
                const float u = a[j][e] * r0 - bv[j][e] * r1;
                d += linv[j][e] * u * u;
                dot += (-2.f * linv[j][e] * u) * bv[j][e];
            }
        for (int m = lpp >> 1; m > 0; m >>= 1) { d += __shfl_xor(d, m, 64); dot += __shfl_xor(dot, m, 64); }
        if (ok && li == 0) dacc += d;
        if (g1 && ok) {
            const float k = n1 > 0.f ? dot * r1 * r1 / n1 : 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                float g[EP];
#pragma unroll
                for (int e = 0; e < EP; e++) {
                    const float u = a[j][e] * r0 - bv[j][e] * r1;
                    g[e] = gscale * inv_hw * ((-2.f * linv[j][e] * u) * r1 - bv[j][e] * k);
                }
                *(uint4*)(g1 + ((size_t)b * HW + p) * C + (j * lpp + li) * EP) = pack16(g, (T*)nullptr);
            }
        }
    }
    dacc = wave_sum(dacc);
    __shared__ float wred[4];                             // one atomic per workgroup (same-address atomics serialise)
    if (lane == 0) wred[threadIdx.x >> 6] = dacc;
    __syncthreads();
    if (det_on()) {            // deterministic mode: domain = sample, one slot per workgroup
        const int nslots = gridDim.x;
        float* slot = det_slot(b, gridDim.y, blockIdx.x, nslots, 1);
        if (threadIdx.x == 0) slot[0] = ((wred[0] + wred[1]) + (wred[2] + wred[3])) * inv_hw;
        if (det_arrive_wg(b, nslots)) {
            __shared__ float dred[16];
            const float t = det_total_wg(b, nslots, 1, 0, dred);
            if (threadIdx.x == 0) val[b] += t;
        }
        return;
    }
    if (threadIdx.x == 0) {
        const float t = (wred[0] + wred[1]) + (wred[2] + wred[3]);
        if (t != 0.f) atomicAdd(val + b, t * inv_hw);
    }
}

// out[0] = mean_b val[b]
__global__ void mean_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { float s = 0.f; for (int i = 0; i < n; i++) s += v[i]; out[0] = s / (float)n; }
}

// =================================================================== C ABI
extern "C" int dge_lpips_prep(const float* img, void* x, int B, int HW, int cpad, const float* host_shift3,
                              const float* host_scale3, int dtype, hipStream_t s) {
    DGE_CHECK(cpad >= 3, "lpips_prep: cpad < 3");
    const long n = (long)B * HW;
    const float i0 = 1.f / host_scale3[0], i1 = 1.f / host_scale3[1], i2 = 1.f / host_scale3[2];
    if (dtype == DGE_BF16) hipLaunchKernelGGL(lpips_prep_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img, (bf16_t*)x, B, HW, cpad, host_shift3[0], host_shift3[1], host_shift3[2], i0, i1, i2);
    else hipLaunchKernelGGL(lpips_prep_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img, (float*)x, B, HW, cpad, host_shift3[0], host_shift3[1], host_shift3[2], i0, i1, i2);
    DGE_LAUNCH_CHECK("lpips_prep");
    return 0;
}

extern "C" int dge_lpips_prep_bwd(const void* gx, float* gimg, int B, int HW, int cpad, const float* host_scale3, float factor,
                                  int accumulate, int dtype, hipStream_t s) {
    const long n = (long)B * HW;
    const float i0 = 1.f / host_scale3[0], i1 = 1.f / host_scale3[1], i2 = 1.f / host_scale3[2];
    if (dtype == DGE_BF16) hipLaunchKernelGGL(lpips_prep_bwd_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)gx, gimg, B, HW, cpad, i0, i1, i2, factor, accumulate);
    else hipLaunchKernelGGL(lpips_prep_bwd_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)gx, gimg, B, HW, cpad, i0, i1, i2, factor, accumulate);
    DGE_LAUNCH_CHECK("lpips_prep_bwd");
    return 0;
}

extern "C" int dge_maxpool2(const void* x, void* y, int B, int H, int W, int C, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0, "maxpool2: C %% %d != 0", ep);
    const long n = (long)B * (H / 2) * (W / 2) * (C / ep);
    if (n == 0) return 0;
    if (dtype == DGE_BF16) hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, H, W, C);
    else hipLaunchKernelGGL(maxpool_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)x, (float*)y, B, H, W, C);
    DGE_LAUNCH_CHECK("maxpool2");
    return 0;
}

static int maxpool2_bwd_launch(const void* gy, const void* x, const void* addend, void* gx, int B, int H, int W, int C, int relu_mode,
                               int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0, "maxpool2_bwd: C %% %d != 0", ep);
    DGE_CHECK((H % 2 == 0 && W % 2 == 0) || relu_mode == 0 || relu_mode == 3, "maxpool2_relu_bwd: H and W must be even");
    const long n = (long)B * H * W * (C / ep);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)gy, (const bf16_t*)x, (const bf16_t*)addend, (bf16_t*)gx, B, H, W, C, relu_mode);
    else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)gy, (const float*)x, (const float*)addend, (float*)gx, B, H, W, C, relu_mode);
    DGE_LAUNCH_CHECK("maxpool2_bwd");
    return 0;
}

extern "C" int dge_maxpool2_bwd(const void* gy, const void* x, const void* addend, void* gx, int B, int H, int W, int C, int dtype,
                                hipStream_t s) {
    return maxpool2_bwd_launch(gy, x, addend, gx, B, H, W, C, 0, dtype, s);
}

extern "C" int dge_maxpool2_bwd_relu(const void* gy, const void* x, const void* addend, void* gx, int B, int H, int W, int C, int dtype,
                                     hipStream_t s) {
    return maxpool2_bwd_launch(gy, x, addend, gx, B, H, W, C, 3, dtype, s);
}

extern "C" int dge_maxpool2_relu_bwd(const void* gy, const void* x, void* gx, int B, int H, int W, int C, int guided, int dtype,
                                     hipStream_t s) {
    return maxpool2_bwd_launch(gy, x, nullptr, gx, B, H, W, C, guided ? 2 : 1, dtype, s);
}

template <typename T>
static int lpips_head_launch(const void* feat, const float* lin, float* val, void* g1, int B, int HW, int C, float gscale, hipStream_t s) {
    constexpr int EP = Elem<T>::PER16;
    const int chunks = C / EP;
    const int lpp = chunks >= 64 ? 64 : chunks;          // power of two for C in {64,128,256,512}
    const int nj = chunks / lpp;
    int ppb = 4 * (64 / lpp) * 8;                         // 8 iterations per wave
    dim3 grid((HW + ppb - 1) / ppb, B);
#define LH(NJ) hipLaunchKernelGGL((lpips_head_kernel<T, NJ>), grid, dim3(256), 0, s, (const T*)feat, lin, val, (T*)g1, B, HW, C, gscale, lpp, ppb)
    if (nj == 1) LH(1); else if (nj == 2) LH(2); else if (nj == 4) LH(4); else { dge_set_error("lpips_head: unsupported C=%d", C); return -1; }
#undef LH
    return 0;
}

extern "C" int dge_lpips_head(const void* feat, const float* lin, float* val, void* g1, int B, int HW, int C, float gscale,
                              int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0 && ((C / ep) & (C / ep - 1)) == 0, "lpips_head: C=%d must be a power-of-two multiple of %d", C, ep);
    const int rc = dtype == DGE_BF16 ? lpips_head_launch<bf16_t>(feat, lin, val, g1, B, HW, C, gscale, s)
                                      : lpips_head_launch<float>(feat, lin, val, g1, B, HW, C, gscale, s);
    if (rc) return rc;
    DGE_LAUNCH_CHECK("lpips_head");
    return 0;
}

extern "C" int dge_mean(const float* v, float* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, s, v, out, n);
    DGE_LAUNCH_CHECK("mean");
    return 0;
}
