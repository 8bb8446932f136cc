// Grad-CAM++ attention maps on the device (reference metric/grad_cam.py, driven by
// E_mis_align_cropping_s1.py:99-106,159-170).  The VGG16 convolutions run on conv_igemm (forward and data gradient);
// this file holds what sits between them and the masks: guided ReLU backward, adaptive average pool + flatten and its
// adjoint, the class-target selection, the Grad-CAM++ channel weights / weighted channel sum, min-max normalisation +
// bilinear resize with OpenCV's INTER_LINEAR geometry, and mask2cam (JET look-up, overlay, the reference's sequential
// normalisation).  All streaming / HBM-bound; NHWC activations of type T, everything else f32.
#include "common.h"
#include "../../include/dge_hip.h"

namespace {

// gpre = (a > 0 && g > 0) ? g : 0  -- nn.ReLU backward followed by GuidedBackPropagation.backward_hook
// (grad_cam.py:208-217: clamp(grad_in, min=0)); guided == 0 gives the plain ReLU backward.
template <typename T>
__global__ void guided_relu_bwd_kernel(const T* __restrict__ g, const T* __restrict__ a, T* __restrict__ gpre, long nchunks,
                                       int guided) {
    constexpr int EP = Elem<T>::PER16;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nchunks; i += (long)gridDim.x * 256) {
        float fg[EP], fa[EP];
        unpack16(((const uint4*)g)[i], fg, (T*)nullptr);
        unpack16(((const uint4*)a)[i], fa, (T*)nullptr);
#pragma unroll
        for (int e = 0; e < EP; e++) fg[e] = (fa[e] > 0.f && (!guided || fg[e] > 0.f)) ? fg[e] : 0.f;
        ((uint4*)gpre)[i] = pack16(fg, (T*)nullptr);
    }
}

__device__ __forceinline__ int win_lo(int i, int n) { return (i * n) / 7; }
__device__ __forceinline__ int win_hi(int i, int n) { return ((i + 1) * n + 6) / 7; }

// nn.AdaptiveAvgPool2d((7,7)) + torch.flatten(x, 1) of an NHWC tensor: y[b, c*49 + i*7 + j]
template <typename T>
__global__ void adaptive_pool7_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
    const long n = (long)B * 49 * C;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= n) return;
    const int c = idx % C; long r = idx / C; const int j = r % 7; r /= 7; const int i = r % 7; const int b = r / 7;
    const int y0 = win_lo(i, H), y1 = win_hi(i, H), x0 = win_lo(j, W), x1 = win_hi(j, W);
    float s = 0.f;
    for (int yy = y0; yy < y1; yy++)
        for (int xx = x0; xx < x1; xx++) s += Elem<T>::ld(x + (((size_t)b * H + yy) * W + xx) * C + c);
    y[(size_t)b * C * 49 + (size_t)c * 49 + i * 7 + j] = s / (float)((y1 - y0) * (x1 - x0));
}
// adjoint: gx[b,yy,xx,c] = sum over the windows (i,j) that contain (yy,xx) of gy[b, c*49+i*7+j] / |window|
template <typename T>
__global__ void adaptive_pool7_bwd_kernel(const float* __restrict__ gy, T* __restrict__ gx, int B, int H, int W, int C) {
    const long n = (long)B * H * W * C;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= n) return;
    const int c = idx % C; long r = idx / C; const int xx = r % W; r /= W; const int yy = r % H; const int b = r / H;
    float s = 0.f;
    for (int i = 0; i < 7; i++) {
        const int y0 = win_lo(i, H), y1 = win_hi(i, H);
        if (yy < y0 || yy >= y1) continue;
        for (int j = 0; j < 7; j++) {
            const int x0 = win_lo(j, W), x1 = win_hi(j, W);
            if (xx < x0 || xx >= x1) continue;
            s += gy[(size_t)b * C * 49 + (size_t)c * 49 + i * 7 + j] / (float)((y1 - y0) * (x1 - x0));
        }
    }
    Elem<T>::st(gx + idx, s);
}

// grad_cam.py:166-170 on the device (no host round trip): per-row arg-max of the logits (first maximum, like np.argmax),
// the most frequent of those indices (smallest index among ties: np.argmax(np.bincount(index))), and the gradient of
// target = mean_n logits[n, index_max]: glogits[n, k] = (k == index_max) / B.  One workgroup.
__global__ void class_target_kernel(const float* __restrict__ logits, const int* __restrict__ index_in, int* __restrict__ index_out,
                                    float* __restrict__ glogits, int B, int K) {
    __shared__ int row_idx[1024];
    __shared__ int best;
    const int tid = threadIdx.x;
    for (int b = tid; b < B; b += 256) {
        if (index_in) { row_idx[b] = index_in[b]; continue; }
        int am = 0; float mv = logits[(size_t)b * K];
        for (int k = 1; k < K; k++) { const float v = logits[(size_t)b * K + k]; if (v > mv) { mv = v; am = k; } }
        row_idx[b] = am;
    }
    __syncthreads();
    if (tid == 0) {
        int bi = 0, bc = -1;
        for (int b = 0; b < B; b++) {
            int cnt = 0;
            for (int q = 0; q < B; q++) cnt += (row_idx[q] == row_idx[b]);
            if (cnt > bc || (cnt == bc && row_idx[b] < bi)) { bc = cnt; bi = row_idx[b]; }
        }
        best = bi;
        index_out[0] = bi;
        for (int b = 0; b < B; b++) index_out[1 + b] = row_idx[b];
    }
    __syncthreads();
    const float inv = 1.f / (float)B;
    for (long i = tid; i < (long)B * K; i += 256) glogits[i] = ((int)(i % K) == best) ? inv : 0.f;
}

// one row of a dense weight, selected by a device-side index: y[b, :] = scale * w[index[0], :]  (the first backward
// step of the classifier: d target / d hidden2 = W6[index_max, :] / B for every sample)
__global__ void gather_row_kernel(const float* __restrict__ w, const int* __restrict__ index, float* __restrict__ y, int B, int I,
                                  float scale) {
    const float* row = w + (size_t)index[0] * I;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < (long)B * I; i += (long)gridDim.x * 256) y[i] = scale * row[i % I];
}

// Channel weights.  Grad-CAM++ (mode 1, grad_cam.py:180-186): s = sum_hw relu(g); weight = s > 0 ? s * (1/s) : 0.
// Grad-CAM (mode 0, :101-102): weight = mean_hw g.  grid (C/64, B); 16 waves split the pixels, lane = channel (the grid is
// small - 64 workgroups for 8 x 512 channels - so each workgroup brings many waves to hide the load latency).
template <typename T>
__global__ __launch_bounds__(1024) void campp_weight_kernel(const T* __restrict__ g, float* __restrict__ wgt, int HW, int C, int mode) {
    __shared__ float red[16][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), wv = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int p = wv; p < HW; p += 16) {
            const float v = Elem<T>::ld(g + ((size_t)b * HW + p) * C + c);
            s += mode ? fmaxf(v, 0.f) : v;
        }
    red[wv][threadIdx.x & 63] = s;
    __syncthreads();
    if (wv == 0 && c < C) {
        s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) s += red[k][threadIdx.x];
        wgt[(size_t)b * C + c] = mode ? (s > 0.f ? s * (1.0f / s) : 0.f) : s / (float)HW;
    }
}

// cam[b,p] = sum_c feature[b,p,c] * weight[b,c]  (grad_cam.py:187-189; Grad-CAM adds a ReLU, :105); one wave per pixel,
// 16-byte channel chunks.
template <typename T>
__global__ void campp_sum_kernel(const T* __restrict__ f, const float* __restrict__ wgt, float* __restrict__ cam, int HW, int C,
                                 int relu) {
    constexpr int EP = Elem<T>::PER16;
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= HW) return;
    float s = 0.f;
    for (int ch = lane; ch < C / EP; ch += 64) {
        float v[EP];
        unpack16(*(const uint4*)(f + ((size_t)b * HW + p) * C + ch * EP), v, (T*)nullptr);
#pragma unroll
        for (int e = 0; e < EP; e++) s += v[e] * wgt[(size_t)b * C + ch * EP + e];
    }
    s = wave_sum(s);
    if (lane == 0) cam[(size_t)b * HW + p] = relu ? fmaxf(s, 0.f) : s;
}

// per-row minimum and maximum: mm[b] = (min, max); one workgroup per row
__global__ void row_minmax_kernel(const float* __restrict__ x, float* __restrict__ mm, int n) {
    __shared__ float lo[256], hi[256];
    const float* row = x + (size_t)blockIdx.x * n;
    float a = INFINITY, z = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) { a = fminf(a, row[i]); z = fmaxf(z, row[i]); }
    lo[threadIdx.x] = a; hi[threadIdx.x] = z;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { lo[threadIdx.x] = fminf(lo[threadIdx.x], lo[threadIdx.x + o]); hi[threadIdx.x] = fmaxf(hi[threadIdx.x], hi[threadIdx.x + o]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { mm[2 * blockIdx.x] = lo[0]; mm[2 * blockIdx.x + 1] = hi[0]; }
}

// (cam - min) / (max - min) (grad_cam.py:190-191), then cv2.resize(cam, (W, H)) with INTER_LINEAR (:193): source
// coordinate (d + 0.5) * (src / dst) - 0.5, taps clamped at both edges, float coefficients.
__device__ __forceinline__ void lin_tap(int d, int dn, int sn, int& i0, int& i1, float& a) {
    const double f = ((double)d + 0.5) * ((double)sn / (double)dn) - 0.5;
    int s0 = (int)floor(f);
    a = (float)(f - (double)s0);
    if (s0 < 0) { s0 = 0; a = 0.f; }
    if (s0 >= sn - 1) { s0 = sn - 1; a = 0.f; }
    i0 = s0; i1 = s0 + 1 < sn ? s0 + 1 : sn - 1;
}
__global__ void cam_resize_kernel(const float* __restrict__ cam, const float* __restrict__ mm, float* __restrict__ mask, int B, int h,
                                  int w, int H, int W) {
    const long n = (long)B * H * W;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= n) return;
    const int X = idx % W; long r = idx / W; const int Y = r % H; const int b = r / H;
    int y0, y1, x0, x1; float ay, ax;
    lin_tap(Y, H, h, y0, y1, ay);
    lin_tap(X, W, w, x0, x1, ax);
    const float lo = mm[2 * b], span = mm[2 * b + 1] - lo;
    const float* c = cam + (size_t)b * h * w;
    const float v00 = (c[y0 * w + x0] - lo) / span, v01 = (c[y0 * w + x1] - lo) / span;
    const float v10 = (c[y1 * w + x0] - lo) / span, v11 = (c[y1 * w + x1] - lo) / span;
    const float top = v00 * (1.f - ax) + v01 * ax, bot = v10 * (1.f - ax) + v11 * ax;
    mask[idx] = top * (1.f - ay) + bot * ay;
}

// mask2cam part 1 (grad_cam.py:241-248): heat = JET[uint8(255 * mask)] / 255 in RGB order; cam = heat + img; per-block
// partial (min img, min cam, max cam) of sample b into part[b][blk][3].  lut: 256 x (r, g, b) bytes stored as ints.
__global__ void jet_overlay_kernel(const float* __restrict__ mask, const float* __restrict__ img, const int* __restrict__ lut,
                                   float* __restrict__ heat, float* __restrict__ cam, float* __restrict__ part, int HW) {
    __shared__ float s0[256], s1[256], s2[256];
    const int b = blockIdx.y;
    float mi = INFINITY, mc = INFINITY, xc = -INFINITY;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const float m = mask[(size_t)b * HW + p];
        int u = (int)(255.f * m);                     // np.uint8(255 * j): truncation
        u = u < 0 ? 0 : (u > 255 ? 255 : u);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const size_t o = ((size_t)b * 3 + ch) * HW + p;
            const float hv = (float)lut[u * 3 + ch] / 255.f, iv = img[o], cv = hv + iv;
            heat[o] = hv; cam[o] = cv;
            mi = fminf(mi, iv); mc = fminf(mc, cv); xc = fmaxf(xc, cv);
        }
    }
    s0[threadIdx.x] = mi; s1[threadIdx.x] = mc; s2[threadIdx.x] = xc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            s0[threadIdx.x] = fminf(s0[threadIdx.x], s0[threadIdx.x + o]);
            s1[threadIdx.x] = fminf(s1[threadIdx.x], s1[threadIdx.x + o]);
            s2[threadIdx.x] = fmaxf(s2[threadIdx.x], s2[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* o = part + ((size_t)b * gridDim.x + blockIdx.x) * 3;
        o[0] = s0[0]; o[1] = s1[0]; o[2] = s2[0];
    }
}
// mask2cam part 2 (grad_cam.py:249-250): `cam[i] -= min(cam)` takes the minimum over the WHOLE array as it stands when
// sample i is processed (samples < i normalised, sample i = heat + img, samples > i still the raw images), then
// `cam[i] /= max(cam[i])`.  The recurrence over i needs only three extrema per sample.  One workgroup.
__global__ void cam_norm_coef_kernel(const float* __restrict__ part, float* __restrict__ coef, int B, int nblk) {
    __shared__ float ext[1024 * 3];
    for (int b = threadIdx.x; b < B; b += 256) {
        float mi = INFINITY, mc = INFINITY, xc = -INFINITY;
        for (int k = 0; k < nblk; k++) {
            const float* o = part + ((size_t)b * nblk + k) * 3;
            mi = fminf(mi, o[0]); mc = fminf(mc, o[1]); xc = fmaxf(xc, o[2]);
        }
        ext[b * 3] = mi; ext[b * 3 + 1] = mc; ext[b * 3 + 2] = xc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float done_min = INFINITY;                       // minimum over the already normalised samples
        for (int i = 0; i < B; i++) {
            float m = fminf(done_min, ext[i * 3 + 1]);
            for (int j = i + 1; j < B; j++) m = fminf(m, ext[j * 3]);
            const float scl = ext[i * 3 + 2] - m;
            coef[2 * i] = m; coef[2 * i + 1] = scl;
            done_min = fminf(done_min, (ext[i * 3 + 1] - m) / scl);
        }
    }
}
__global__ void cam_norm_apply_kernel(float* __restrict__ cam, const float* __restrict__ coef, long per_sample, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int b = (int)(i / per_sample);
        cam[i] = (cam[i] - coef[2 * b]) / coef[2 * b + 1];
    }
}

static inline unsigned blocks_for(long n, long cap = 65535L * 16) {
    long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int dge_guided_relu_bwd(const void* g, const void* a, void* gpre, long n, int guided, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(n % ep == 0, "guided_relu_bwd: n %% %d != 0", ep);
    if (n == 0) return 0;
    const long chunks = n / ep;
    const unsigned grid = blocks_for(chunks, 8192);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(guided_relu_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)a, (bf16_t*)gpre, chunks, guided);
    else hipLaunchKernelGGL(guided_relu_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)g, (const float*)a, (float*)gpre, chunks, guided);
    DGE_LAUNCH_CHECK("guided_relu_bwd");
    return 0;
}

extern "C" int dge_adaptive_pool7(const void* x, float* y, int B, int H, int W, int C, int dtype, hipStream_t s) {
    DGE_CHECK(B > 0 && H > 0 && W > 0 && C > 0, "adaptive_pool7: empty input");
    const long n = (long)B * 49 * C;
    if (dtype == DGE_BF16) hipLaunchKernelGGL(adaptive_pool7_kernel<bf16_t>, dim3(blocks_for(n)), dim3(256), 0, s, (const bf16_t*)x, y, B, H, W, C);
    else hipLaunchKernelGGL(adaptive_pool7_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, s, (const float*)x, y, B, H, W, C);
    DGE_LAUNCH_CHECK("adaptive_pool7");
    return 0;
}

extern "C" int dge_adaptive_pool7_bwd(const float* gy, void* gx, int B, int H, int W, int C, int dtype, hipStream_t s) {
    DGE_CHECK(B > 0 && H > 0 && W > 0 && C > 0, "adaptive_pool7_bwd: empty input");
    const long n = (long)B * H * W * C;
    if (dtype == DGE_BF16) hipLaunchKernelGGL(adaptive_pool7_bwd_kernel<bf16_t>, dim3(blocks_for(n)), dim3(256), 0, s, gy, (bf16_t*)gx, B, H, W, C);
    else hipLaunchKernelGGL(adaptive_pool7_bwd_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, s, gy, (float*)gx, B, H, W, C);
    DGE_LAUNCH_CHECK("adaptive_pool7_bwd");
    return 0;
}

extern "C" int dge_class_target(const float* logits, const int* index_in, int* index_out, float* glogits, int B, int K,
                                hipStream_t s) {
    DGE_CHECK(B >= 1 && B <= 1024 && K >= 1, "class_target: B=%d (1..1024), K=%d", B, K);
    hipLaunchKernelGGL(class_target_kernel, dim3(1), dim3(256), 0, s, logits, index_in, index_out, glogits, B, K);
    DGE_LAUNCH_CHECK("class_target");
    return 0;
}

extern "C" int dge_gather_row(const float* w, const int* index, float* y, int B, int I, float scale, hipStream_t s) {
    DGE_CHECK(B >= 1 && I >= 1, "gather_row: B=%d I=%d", B, I);
    hipLaunchKernelGGL(gather_row_kernel, dim3(blocks_for((long)B * I, 1024)), dim3(256), 0, s, w, index, y, B, I, scale);
    DGE_LAUNCH_CHECK("gather_row");
    return 0;
}

extern "C" int dge_campp_map(const void* grad, const void* feat, float* wgt, float* cam, float* minmax, int B, int HW, int C,
                             int mode, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(B >= 1 && HW >= 1 && C % ep == 0, "campp_map: B=%d HW=%d C=%d (C %% %d)", B, HW, C, ep);
    dim3 gw((C + 63) / 64, B), gs((HW + 3) / 4, B);
    if (dtype == DGE_BF16) {
        hipLaunchKernelGGL(campp_weight_kernel<bf16_t>, gw, dim3(1024), 0, s, (const bf16_t*)grad, wgt, HW, C, mode);
        hipLaunchKernelGGL(campp_sum_kernel<bf16_t>, gs, dim3(256), 0, s, (const bf16_t*)feat, wgt, cam, HW, C, mode == 0);
    } else {
        hipLaunchKernelGGL(campp_weight_kernel<float>, gw, dim3(1024), 0, s, (const float*)grad, wgt, HW, C, mode);
        hipLaunchKernelGGL(campp_sum_kernel<float>, gs, dim3(256), 0, s, (const float*)feat, wgt, cam, HW, C, mode == 0);
    }
    hipLaunchKernelGGL(row_minmax_kernel, dim3(B), dim3(256), 0, s, cam, minmax, HW);
    DGE_LAUNCH_CHECK("campp_map");
    return 0;
}

extern "C" int dge_cam_resize(const float* cam, const float* minmax, float* mask, int B, int h, int w, int H, int W, hipStream_t s) {
    DGE_CHECK(B >= 1 && h >= 1 && w >= 1 && H >= 1 && W >= 1, "cam_resize: bad sizes");
    hipLaunchKernelGGL(cam_resize_kernel, dim3(blocks_for((long)B * H * W)), dim3(256), 0, s, cam, minmax, mask, B, h, w, H, W);
    DGE_LAUNCH_CHECK("cam_resize");
    return 0;
}

extern "C" int dge_mask2cam_blocks(int HW) { int g = (HW + 255) / 256; return g > 256 ? 256 : (g < 1 ? 1 : g); }

extern "C" int dge_mask2cam(const float* mask, const float* img, const int* lut, float* heat, float* cam, float* part, float* coef,
                            int B, int HW, hipStream_t s) {
    DGE_CHECK(B >= 1 && B <= 1024 && HW >= 1, "mask2cam: B=%d (1..1024) HW=%d", B, HW);
    const int nblk = dge_mask2cam_blocks(HW);
    hipLaunchKernelGGL(jet_overlay_kernel, dim3(nblk, B), dim3(256), 0, s, mask, img, lut, heat, cam, part, HW);
    hipLaunchKernelGGL(cam_norm_coef_kernel, dim3(1), dim3(256), 0, s, part, coef, B, nblk);
    const long n = (long)B * 3 * HW;
    hipLaunchKernelGGL(cam_norm_apply_kernel, dim3(blocks_for(n, 4096)), dim3(256), 0, s, cam, coef, (long)3 * HW, n);
    DGE_LAUNCH_CHECK("mask2cam");
    return 0;
}
