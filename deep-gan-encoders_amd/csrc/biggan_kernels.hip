// BigGAN-deep support kernels (reference model/biggan_generator.py): conditional batch norm as a
// per-(b,c) affine, skip-path channel drop + nearest upsample, self-attention, final tanh.
// The convolutions / dense layers run on conv_igemm / linear_kernel.
#include "common.h"
#include "../../include/dge_hip.h"

// a = (1 + scale[b,c]) / sqrt(var[c] + eps) ; b = offset[b,c] - mean[c]*a          (:127-150)
__global__ void cbn_affine_kernel(const float* __restrict__ scale, const float* __restrict__ offset, int ld,
                                  const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                  float* __restrict__ a, float* __restrict__ bq, int B, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx % C;
    const float av = (1.f + scale[(size_t)b * ld + c]) / sqrtf(var[c] + eps);
    a[idx] = av;
    bq[idx] = offset[(size_t)b * ld + c] - mean[c] * av;
}

// y[b,oy,ox,c] = x[b,oy>>up,ox>>up,c] for c < Cout
template <typename T>
__global__ void slice_up_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int Cin, int Cout, int up) {
    constexpr int EP = Elem<T>::PER16;
    const int OH = H << up, OW = W << up, cpt = Cout / EP;
    const long n = (long)B * OH * OW * cpt;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= n) return;
    const int ch = idx % cpt; long r = idx / cpt; const int ox = r % OW; r /= OW; const int oy = r % OH; const int b = r / OH;
    *(uint4*)(y + (((size_t)b * OH + oy) * OW + ox) * Cout + ch * EP) =
        *(const uint4*)(x + (((size_t)b * H + (oy >> up)) * W + (ox >> up)) * Cin + ch * EP);
}

// One wavefront per query.  Scores: lanes own keys m = lane, lane+64, ... (query row held in LDS);
// softmax by wave reductions; output: lanes own DV/64-channel slices and walk all keys (probabilities in LDS).
template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ Q, const T* __restrict__ K, const T* __restrict__ V,
                                                         T* __restrict__ O, int B, int N, int M, int D, int DV) {
    extern __shared__ float sm[];                       // per wave: q[D] + p[M]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* qs = sm + wave * (D + M);
    float* ps = qs + D;
    const long qi = (long)blockIdx.x * 4 + wave;
    if (qi >= (long)B * N) return;
    const int b = qi / N;
    const T* qp = Q + (size_t)qi * D;
    for (int d = lane; d < D; d += 64) qs[d] = Elem<T>::ld(qp + d);
    __builtin_amdgcn_wave_barrier();
    const T* Kb = K + (size_t)b * M * D;
    float mx = -INFINITY;
    for (int m = lane; m < M; m += 64) {
        const T* kp = Kb + (size_t)m * D;
        float s = 0.f;
        for (int d = 0; d < D; d++) s += qs[d] * Elem<T>::ld(kp + d);
        ps[m] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int m = lane; m < M; m += 64) { const float e = __expf(ps[m] - mx); ps[m] = e; sum += e; }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.f / sum;
    const T* Vb = V + (size_t)b * M * DV;
    T* op = O + (size_t)qi * DV;
    for (int c0 = lane; c0 < DV; c0 += 64) {
        float acc = 0.f;
        for (int m = 0; m < M; m++) acc += ps[m] * Elem<T>::ld(Vb + (size_t)m * DV + c0);
        Elem<T>::st(op + c0, acc * inv);
    }
}

template <typename T>
__global__ void rgb_tanh_kernel(const T* __restrict__ x, float* __restrict__ img, int B, int HW, int C) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= (long)B * HW) return;
    const int b = idx / HW, p = idx % HW;
    const T* xp = x + (size_t)idx * C;
#pragma unroll
    for (int c = 0; c < 3; c++) img[((size_t)b * 3 + c) * HW + p] = tanhf(Elem<T>::ld(xp + c));
}

// ------------------------------------------------------------------- backward (data gradient w.r.t. z, E_align --mtype 4)
// Backward of the conv prologue u = relu(a[b,c]*x + bq[b,c]) (conditional batch norm + ReLU, GenBlock :175-203):
//   gx = [a*x + bq > 0] * a * gu ;  stats[b,c,:] (pre-zeroed) += (sum_p m*gu*x, sum_p m*gu) = (d/da, d/dbq).
// Channel groups of Cg channels (blockIdx.z) so that any C streams with 256 threads.
template <typename T>
__global__ __launch_bounds__(256) void affine_relu_bwd_kernel(const T* __restrict__ gu, const T* __restrict__ x,
                                                               const float* __restrict__ a, const float* __restrict__ bq,
                                                               T* __restrict__ gx, float* __restrict__ stats, int HW, int C, int Cg) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 2 * EP];
    const int b = blockIdx.y, c0 = blockIdx.z * Cg;
    const int cpt = Cg / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    float s[2][EP], av[EP], bv[EP];
#pragma unroll
    for (int e = 0; e < EP; e++) {
        s[0][e] = s[1][e] = 0.f;
        av[e] = a[(size_t)b * C + c0 + chunk * EP + e];
        bv[e] = bq[(size_t)b * C + c0 + chunk * EP + e];
    }
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            const size_t o = ((size_t)b * HW + p) * C + c0 + chunk * EP;
            float gv[EP], xv[EP], r[EP];
            unpack16(*(const uint4*)(gu + o), gv, (T*)nullptr);
            unpack16(*(const uint4*)(x + o), xv, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < EP; e++) {
                const float m = (av[e] * xv[e] + bv[e] > 0.f) ? gv[e] : 0.f;
                r[e] = m * av[e];
                s[0][e] += m * xv[e]; s[1][e] += m;
            }
            *(uint4*)(gx + o) = pack16(r, (T*)nullptr);
        }
    }
    block_chan_flush<EP, 2>(s, cpt, ppi, stats + ((size_t)b * C + c0) * 2, Cg, red);
}

// adjoint of slice_up, accumulated: gx[b,y,x,c] += sum over the 2^up x 2^up block of gy[b,.,.,c] for c < Cout
template <typename T>
__global__ void slice_up_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int B, int H, int W, int Cin, int Cout, int up) {
    constexpr int EP = Elem<T>::PER16;
    const int cpt = Cout / EP;
    const long n = (long)B * H * W * cpt;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= n) return;
    const int ch = idx % cpt; long r = idx / cpt; const int xq = r % W; r /= W; const int y = r % H; const int b = r / H;
    T* dst = gx + (((size_t)b * H + y) * W + xq) * Cin + ch * EP;
    float acc[EP], gv[EP];
    unpack16(*(const uint4*)dst, acc, (T*)nullptr);
    const int f = 1 << up, OW = W << up;
    for (int dy = 0; dy < f; dy++)
        for (int dx = 0; dx < f; dx++) {
            unpack16(*(const uint4*)(gy + (((size_t)b * (H << up) + (y << up) + dy) * OW + (xq << up) + dx) * Cout + ch * EP), gv, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < EP; e++) acc[e] += gv[e];
        }
    *(uint4*)dst = pack16(acc, (T*)nullptr);
}

// Row softmax over M keys, in place (scores S [R, M] of the self-attention backward recomputation); one wavefront per row.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* __restrict__ S, long R, int M) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    T* sp = S + (size_t)row * M;
    float mx = -INFINITY;
    for (int m = lane; m < M; m += 64) mx = fmaxf(mx, Elem<T>::ld(sp + m));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int m = lane; m < M; m += 64) sum += __expf(Elem<T>::ld(sp + m) - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int m = lane; m < M; m += 64) Elem<T>::st(sp + m, __expf(Elem<T>::ld(sp + m) - mx) * inv);
}

// gS = P * (gP - sum_m P*gP) per row, written over gP
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const T* __restrict__ P, T* __restrict__ gP, long R, int M) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const T* pp = P + (size_t)row * M;
    T* gp = gP + (size_t)row * M;
    float dot = 0.f;
    for (int m = lane; m < M; m += 64) dot += Elem<T>::ld(pp + m) * Elem<T>::ld(gp + m);
    dot = wave_sum(dot);
    for (int m = lane; m < M; m += 64) Elem<T>::st(gp + m, Elem<T>::ld(pp + m) * (Elem<T>::ld(gp + m) - dot));
}

// backward of rgb_tanh: gy[b,p,c] = gimg[b,c,p] * (1 - img[b,c,p]^2) for c < 3, 0 for the other (unused) channels
template <typename T>
__global__ void rgb_tanh_bwd_kernel(const float* __restrict__ gimg, const float* __restrict__ img, T* __restrict__ gy, int B, int HW, int C) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= (long)B * HW) return;
    const int b = idx / HW, p = idx % HW;
    T* gp = gy + (size_t)idx * C;
    for (int c = 0; c < C; c++) {
        float v = 0.f;
        if (c < 3) { const size_t o = ((size_t)b * 3 + c) * HW + p; const float t = img[o]; v = gimg[o] * (1.f - t * t); }
        Elem<T>::st(gp + c, v);
    }
}

// =================================================================== C ABI
extern "C" int dge_cbn_affine(const float* scale, const float* offset, int ld, const float* mean, const float* var, float eps,
                              float* a, float* b, int B, int C, hipStream_t s) {
    hipLaunchKernelGGL(cbn_affine_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, scale, offset, ld, mean, var, eps, a, b, B, C);
    DGE_LAUNCH_CHECK("cbn_affine");
    return 0;
}

extern "C" int dge_slice_up(const void* x, void* y, int B, int H, int W, int Cin, int Cout, int up, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(Cout % ep == 0 && Cin % ep == 0 && Cout <= Cin && (up == 0 || up == 1), "slice_up: bad arguments");
    const long n = (long)B * (H << up) * (W << up) * (Cout / ep);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(slice_up_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, H, W, Cin, Cout, up);
    else hipLaunchKernelGGL(slice_up_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)x, (float*)y, B, H, W, Cin, Cout, up);
    DGE_LAUNCH_CHECK("slice_up");
    return 0;
}

extern "C" int dge_attention(const void* q, const void* k, const void* v, void* o, int B, int N, int M, int D, int DV, int dtype,
                             hipStream_t s) {
    const size_t shm = (size_t)4 * (D + M) * sizeof(float);
    DGE_CHECK(shm <= 64 * 1024, "attention: %d keys x %d dims do not fit the LDS score buffer", M, D);
    const long nq = (long)B * N;
    dim3 grid((unsigned)((nq + 3) / 4));
    if (dtype == DGE_BF16) hipLaunchKernelGGL(attention_kernel<bf16_t>, grid, dim3(256), shm, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, B, N, M, D, DV);
    else hipLaunchKernelGGL(attention_kernel<float>, grid, dim3(256), shm, s, (const float*)q, (const float*)k, (const float*)v, (float*)o, B, N, M, D, DV);
    DGE_LAUNCH_CHECK("attention");
    return 0;
}

extern "C" int dge_rgb_tanh(const void* x, float* img, int B, int HW, int C, int dtype, hipStream_t s) {
    DGE_CHECK(C >= 3, "rgb_tanh: need at least 3 channels");
    const long n = (long)B * HW;
    if (dtype == DGE_BF16) hipLaunchKernelGGL(rgb_tanh_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x, img, B, HW, C);
    else hipLaunchKernelGGL(rgb_tanh_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)x, img, B, HW, C);
    DGE_LAUNCH_CHECK("rgb_tanh");
    return 0;
}

extern "C" int dge_affine_relu_bwd(const void* gu, const void* x, const float* a, const float* b, void* gx, float* stats, int B,
                                   int HW, int C, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0 && ((C / ep) & (C / ep - 1)) == 0, "affine_relu_bwd: C=%d must be a power-of-two multiple of %d", C, ep);
    const int Cg = C / ep > 256 ? 256 * ep : C;
    dim3 grid(dge_stream_grid(HW, 256 / (Cg / ep), B), B, C / Cg);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(affine_relu_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)gu, (const bf16_t*)x, a, b, (bf16_t*)gx, stats, HW, C, Cg);
    else hipLaunchKernelGGL(affine_relu_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)gu, (const float*)x, a, b, (float*)gx, stats, HW, C, Cg);
    DGE_LAUNCH_CHECK("affine_relu_bwd");
    return 0;
}

extern "C" int dge_slice_up_bwd(const void* gy, void* gx, int B, int H, int W, int Cin, int Cout, int up, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(Cout % ep == 0 && Cin % ep == 0 && Cout <= Cin && (up == 0 || up == 1), "slice_up_bwd: bad arguments");
    const long n = (long)B * H * W * (Cout / ep);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(slice_up_bwd_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)gy, (bf16_t*)gx, B, H, W, Cin, Cout, up);
    else hipLaunchKernelGGL(slice_up_bwd_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)gy, (float*)gx, B, H, W, Cin, Cout, up);
    DGE_LAUNCH_CHECK("slice_up_bwd");
    return 0;
}

extern "C" int dge_softmax_rows(void* S, long R, int M, int dtype, hipStream_t s) {
    dim3 grid((unsigned)((R + 3) / 4));
    if (dtype == DGE_BF16) hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)S, R, M);
    else hipLaunchKernelGGL(softmax_rows_kernel<float>, grid, dim3(256), 0, s, (float*)S, R, M);
    DGE_LAUNCH_CHECK("softmax_rows");
    return 0;
}

extern "C" int dge_softmax_rows_bwd(const void* P, void* gP, long R, int M, int dtype, hipStream_t s) {
    dim3 grid((unsigned)((R + 3) / 4));
    if (dtype == DGE_BF16) hipLaunchKernelGGL(softmax_rows_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)P, (bf16_t*)gP, R, M);
    else hipLaunchKernelGGL(softmax_rows_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)P, (float*)gP, R, M);
    DGE_LAUNCH_CHECK("softmax_rows_bwd");
    return 0;
}

extern "C" int dge_rgb_tanh_bwd(const float* gimg, const float* img, void* gy, int B, int HW, int C, int dtype, hipStream_t s) {
    DGE_CHECK(C >= 3, "rgb_tanh_bwd: need at least 3 channels");
    const long n = (long)B * HW;
    if (dtype == DGE_BF16) hipLaunchKernelGGL(rgb_tanh_bwd_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gimg, img, (bf16_t*)gy, B, HW, C);
    else hipLaunchKernelGGL(rgb_tanh_bwd_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gimg, img, (float*)gy, B, HW, C);
    DGE_LAUNCH_CHECK("rgb_tanh_bwd");
    return 0;
}

// =================================================================== grouped spectral normalisation
// torch.nn.utils.spectral_norm (one power iteration per forward in train mode) for ALL spectrally-normalised weights
// of a module in five launches instead of ~8 small library calls per weight (BigGAN-deep-256: 162 weights per generator forward;
// the step was bound by the host issuing ~9000 launches).  Entry e: W [O,K] row-major f32, u [O], v [K] (the module's
// buffers, updated in place), scratch t [K] / s [O], outputs W_eff = W / sigma and sigma = u . (W v).
struct SnEntry { const float* W; float* u; float* v; float* t; float* s; float* weff; float* usnap; float* vsnap; int O, K; };

// t[k] += sum over a slab of rows of W[o,k] * u[o]       grid (ceil(K/256), row slabs of 64, entries)
__global__ __launch_bounds__(256) void sn_wtu_kernel(const SnEntry* __restrict__ E) {
    const SnEntry e = E[blockIdx.z];
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int o0 = blockIdx.y * 64;
    if (det_on()) {
        // deterministic mode: domain = (entry, k block), slot = row slab; every workgroup of the grid arrives (slabs beyond the
        // entry's rows contribute zeros), the last one adds the ordered sum
        const int dom = blockIdx.x + gridDim.x * blockIdx.z, nslots = gridDim.y;
        float a = 0.f;
        if (k < e.K && o0 < e.O) {
            const int o1 = min(o0 + 64, e.O);
            for (int o = o0; o < o1; o++) a += e.W[(size_t)o * e.K + k] * e.u[o];
        }
        float* slot = det_slot(dom, gridDim.x * gridDim.z, blockIdx.y, nslots, 256);
        slot[threadIdx.x] = a;
        if (det_arrive_wg(dom, nslots) && k < e.K) e.t[k] += det_sum(dom, nslots, 256, threadIdx.x);
        return;
    }
    if (k >= e.K || o0 >= e.O) return;
    const int o1 = min(o0 + 64, e.O);
    float a = 0.f;
    for (int o = o0; o < o1; o++) a += e.W[(size_t)o * e.K + k] * e.u[o];
    atomicAdd(e.t + k, a);
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

// v = t / max(|t|, eps)                                    grid (entries)
__global__ __launch_bounds__(256) void sn_norm_v_kernel(const SnEntry* __restrict__ E, float eps) {
    __shared__ float red[4];
    const SnEntry e = E[blockIdx.x];
    float a = 0.f;
    for (int k = threadIdx.x; k < e.K; k += 256) { const float x = e.t[k]; a += x * x; }
    const float inv = 1.f / fmaxf(sqrtf(block_sum_256(a, red)), eps);
    for (int k = threadIdx.x; k < e.K; k += 256) e.v[k] = e.t[k] * inv;
}

// s[o] = sum_k W[o,k] * v[k]; one wavefront per row        grid (ceil(maxO/4), entries)
__global__ __launch_bounds__(256) void sn_wv_kernel(const SnEntry* __restrict__ E) {
    const SnEntry e = E[blockIdx.y];
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= e.O) return;
    const float* wr = e.W + (size_t)o * e.K;
    float a = 0.f;
    for (int k = lane; k < e.K; k += 64) a += wr[k] * e.v[k];
    a = wave_sum(a);
    if (lane == 0) e.s[o] = a;
}

// train: u = s / max(|s|, eps); sigma = u . s  (snapshots of u, v for the backward); eval: sigma = u . s with the stored u
__global__ __launch_bounds__(256) void sn_norm_u_kernel(const SnEntry* __restrict__ E, float* __restrict__ sigma, float eps, int training) {
    __shared__ float red[4];
    const SnEntry e = E[blockIdx.x];
    float sg;
    if (training) {
        float a = 0.f;
        for (int o = threadIdx.x; o < e.O; o += 256) { const float x = e.s[o]; a += x * x; }
        const float n2 = block_sum_256(a, red);
        const float inv = 1.f / fmaxf(sqrtf(n2), eps);
        for (int o = threadIdx.x; o < e.O; o += 256) { const float uv = e.s[o] * inv; e.u[o] = uv; e.usnap[o] = uv; }
        sg = n2 * inv;
    } else {
        float a = 0.f;
        for (int o = threadIdx.x; o < e.O; o += 256) { const float uv = e.u[o]; a += uv * e.s[o]; e.usnap[o] = uv; }
        sg = block_sum_256(a, red);
    }
    for (int k = threadIdx.x; k < e.K; k += 256) e.vsnap[k] = e.v[k];
    if (threadIdx.x == 0) sigma[blockIdx.x] = sg;
}

// W_eff = W / sigma                                         grid (blocks, entries)
__global__ __launch_bounds__(256) void sn_scale_kernel(const SnEntry* __restrict__ E, const float* __restrict__ sigma) {
    const SnEntry e = E[blockIdx.y];
    const float inv = 1.f / sigma[blockIdx.y];
    const long n = (long)e.O * e.K;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) e.weff[i] = e.W[i] * inv;
}

// entries: device array of SnEntry (host-built pointer table); t scratch must be pre-zeroed by the caller in train mode.
extern "C" int dge_sn_group(const void* entries, int n, int maxO, int maxK, float* sigma, float eps, int training, hipStream_t s) {
    DGE_CHECK(n > 0 && maxO > 0 && maxK > 0, "sn_group: empty group");
    const SnEntry* E = (const SnEntry*)entries;
    if (training) {
        hipLaunchKernelGGL(sn_wtu_kernel, dim3((maxK + 255) / 256, (maxO + 63) / 64, n), dim3(256), 0, s, E);
        hipLaunchKernelGGL(sn_norm_v_kernel, dim3(n), dim3(256), 0, s, E, eps);
    }
    hipLaunchKernelGGL(sn_wv_kernel, dim3((maxO + 3) / 4, n), dim3(256), 0, s, E);
    hipLaunchKernelGGL(sn_norm_u_kernel, dim3(n), dim3(256), 0, s, E, sigma, eps, training);
    hipLaunchKernelGGL(sn_scale_kernel, dim3(64, n), dim3(256), 0, s, E, sigma);
    DGE_LAUNCH_CHECK("sn_group");
    return 0;
}
extern "C" int dge_sn_entry_size(void) { return (int)sizeof(SnEntry); }

// ------------------------------------------------------------------ parameter gradients of ALL conditional batch norms of a backward
// A BigGANBatchNorm has two spectral-norm linears of the condition vector (scale / offset, biggan BigGANBatchNorm :141-144).  With the
// per-(b,c) sums (dL/da, dL/db) of its affine from the data-gradient epilogue, the gradient w.r.t. weight_orig is
//   gy[b,o] = (g_a - g_b * mean[o]) * rstd[o]   (scale)   |   g_b   (offset)
//   gw[o,k] = sum_b gy[b,o] * cond[b,k]                                  (dense weight gradient)
//   out     = (gw - <gw, W> / sigma * u v^T) / sigma                    (spectral norm backward, u / v constant: sn_weight_grad)
// - 20 launches per batch norm as torch glue (E_BIG: 21 norms, two backward passes per step: ~900 launches of a host-bound step).
// Here: one table entry per linear, two launches per backward.  No atomics: pass 1 leaves gw and one partial <gw, W> per row, pass 2
// sums an entry's partials in a fixed order.
struct CbnGradEntry { const float* dots; const float* mean; const float* rstd; const float* W; const float* u; const float* v; const float* sigma;
                      float* out; int C; int kind; long long row0; };
__global__ __launch_bounds__(256) void cbn_sn_wgrad_rows_kernel(const CbnGradEntry* __restrict__ E, int n, long long rows, const float* __restrict__ cond,
                                                                int B, int K, float* __restrict__ rowdot) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (E[mid].row0 <= row) lo = mid; else hi = mid - 1; }
    const CbnGradEntry e = E[lo];
    const int o = (int)(row - e.row0);
    float part = 0.f;
    for (int k = lane; k < K; k += 64) {
        float gw = 0.f;
        for (int b = 0; b < B; b++) {
            const float ga = e.dots[((size_t)b * e.C + o) * 2], gb = e.dots[((size_t)b * e.C + o) * 2 + 1];
            const float gy = e.kind == 0 ? (ga - gb * e.mean[o]) * e.rstd[o] : gb;
            gw = fmaf(gy, cond[(size_t)b * K + k], gw);
        }
        e.out[(size_t)o * K + k] = gw;
        part = fmaf(gw, e.W[(size_t)o * K + k], part);
    }
    part = wave_sum(part);
    if (lane == 0) rowdot[row] = part;
}
__global__ __launch_bounds__(256) void cbn_sn_wgrad_apply_kernel(const CbnGradEntry* __restrict__ E, int K, const float* __restrict__ rowdot) {
    __shared__ float red[256];
    const CbnGradEntry e = E[blockIdx.y];
    float s = 0.f;
    for (int o = threadIdx.x; o < e.C; o += 256) s += rowdot[e.row0 + o];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
    const float sigma = e.sigma[0];
    const float dot = red[0] / sigma;
    const long long total = (long long)e.C * K;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int o = (int)(i / K), k = (int)(i - (long long)o * K);
        e.out[i] = (e.out[i] - dot * (e.u[o] * e.v[k])) / sigma;
    }
}
extern "C" int dge_cbn_sn_wgrad_entry_size(void) { return (int)sizeof(CbnGradEntry); }
// entries: device array of n CbnGradEntry (row0 = prefix sum of C); rowdot: scratch of `rows` floats; cond [B][K] f32
extern "C" int dge_cbn_sn_wgrad_group(const void* entries, int n, long long rows, int maxC, const float* cond, int B, int K, float* rowdot,
                                      hipStream_t s) {
    DGE_CHECK(entries && cond && rowdot && n >= 1 && rows >= 1 && B >= 1 && K >= 1, "cbn_sn_wgrad_group: bad arguments");
    hipLaunchKernelGGL(cbn_sn_wgrad_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const CbnGradEntry*)entries, n, rows, cond, B, K, rowdot);
    DGE_LAUNCH_CHECK("cbn_sn_wgrad_rows");
    const long long per = (long long)maxC * K;
    int gx = (int)((per + 256 * 8 - 1) / (256 * 8)); gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    hipLaunchKernelGGL(cbn_sn_wgrad_apply_kernel, dim3(gx, n), dim3(256), 0, s, (const CbnGradEntry*)entries, K, rowdot);
    DGE_LAUNCH_CHECK("cbn_sn_wgrad_apply");
    return 0;
}
