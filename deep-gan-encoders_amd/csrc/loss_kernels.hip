// space_loss (reference training_utils.py:54-99) and SSIM (metric/pytorch_ssim.py:18-38) kernels:
// single-pass fused reductions (MSE / cosine / mean / std / KL), crop + average pooling,
// 11x11 Gaussian SSIM forward + analytic backward, and the fused image-space gradient.
// Images are NCHW f32 like the reference's tensors; all kernels are HBM/L2-bound streaming.
#include "common.h"
#include "../../include/dge_hip.h"

struct Crop { int H, W, y0, x0, h, w; };     // source plane H x W, window [y0,y0+h) x [x0,x0+w)

__device__ __forceinline__ void block_atomic_sums(float* vals, int n, float* out, float* red) {
    // vals: per-thread partials (n <= 8); red: [n][4] floats of LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = 0; i < n; i++) {
        const float s = wave_sum(vals[i]);
        if (lane == 0) red[i * 4 + wave] = s;
    }
    __syncthreads();
    if (det_on()) {            // deterministic mode: one slot of 8 sums per workgroup, ordered total into copy 0 by the last arriver
        const int nslots = gridDim.x;
        float* slot = det_slot(0, 1, blockIdx.x, nslots, 8);
        if (threadIdx.x < n) slot[threadIdx.x] = red[threadIdx.x * 4] + red[threadIdx.x * 4 + 1] + red[threadIdx.x * 4 + 2] + red[threadIdx.x * 4 + 3];
        if (det_arrive_wg(0, nslots)) {
            __shared__ float dred[16];
            for (int i = 0; i < n; i++) {
                const float t = det_total_wg(0, nslots, 8, i, dred);
                if (threadIdx.x == 0) out[i] += t;
            }
        }
        return;
    }
    // 16 slot copies of the 8 sums: thousands of workgroups end here and same-address f32 atomics serialise (~40 ns each)
    if (threadIdx.x < n) atomicAdd(out + (blockIdx.x & 15) * 8 + threadIdx.x, red[threadIdx.x * 4] + red[threadIdx.x * 4 + 1] + red[threadIdx.x * 4 + 2] + red[threadIdx.x * 4 + 3]);
}

// sums[0]=sum (a-b)^2, [1]=a.b, [2]=a.a, [3]=b.b, [4]=sum a, [5]=sum b, [6]=sum softmax_c(a)*(log softmax_c(a) - log softmax_c(b))
// a,b: [B,C,H,W] planes with a crop window; softmax is over the C axis (training_utils.py:67: implicit dim = 1
// for 4-D inputs; the 3-D latent case is passed as B'=1, C'=B so that the implicit dim 0 is reproduced).
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ sums, int B, int C, Crop cr) {
    __shared__ float red[7 * 4];
    float v[7] = {0, 0, 0, 0, 0, 0, 0};
    const long npix = (long)B * cr.h * cr.w;
    const size_t plane = (size_t)cr.H * cr.W;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < npix; idx += gridDim.x * 256L) {
        const int x = idx % cr.w; const long r = idx / cr.w; const int y = r % cr.h; const int bb = r / cr.h;
        const size_t off = (size_t)bb * C * plane + (size_t)(cr.y0 + y) * cr.W + cr.x0 + x;
        float ma = -INFINITY, mb = -INFINITY;
        for (int c = 0; c < C; c++) {
            const float av = a[off + c * plane], bv = b[off + c * plane];
            const float d = av - bv;
            v[0] += d * d; v[1] += av * bv; v[2] += av * av; v[3] += bv * bv; v[4] += av; v[5] += bv;
            ma = fmaxf(ma, av); mb = fmaxf(mb, bv);
        }
        float sa = 0.f, sb = 0.f;
        for (int c = 0; c < C; c++) { sa += __expf(a[off + c * plane] - ma); sb += __expf(b[off + c * plane] - mb); }
        const float lsa = __logf(sa) + ma, lsb = __logf(sb) + mb;
        for (int c = 0; c < C; c++) {
            const float la = a[off + c * plane] - lsa, lb = b[off + c * plane] - lsb;
            v[6] += __expf(la) * (la - lb);
        }
    }
    block_atomic_sums(v, 7, sums, red);
}

// out[b,c,y,x] = mean over k x k of src[b,c,y0+ky.., x0+kx..]   (crop + repeated avg_pool2d(2), :81-84)
__global__ void crop_pool_kernel(const float* __restrict__ src, float* __restrict__ dst, int BC, Crop cr, int k) {
    const int oh = cr.h / k, ow = cr.w / k;
    const long n = (long)BC * oh * ow;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= n) return;
    const int x = idx % ow; const long r = idx / ow; const int y = r % oh; const int bc = r / oh;
    const float* p = src + (size_t)bc * cr.H * cr.W + (size_t)(cr.y0 + y * k) * cr.W + cr.x0 + x * k;
    float s = 0.f;
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) s += p[(size_t)i * cr.W + j];
    dst[idx] = s / (float)(k * k);
}

struct Gauss11 { float g[11]; };

// SSIM forward on [BC, h, w] planes: ssim_sum += sum of the SSIM map; optional derivative maps
// dmap[0] = dS/dmu2, dmap[1] = dS/dE[b^2], dmap[2] = dS/dE[ab]  (each [BC,h,w]) for the backward.
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ ssim_sum, float* __restrict__ dmap,
                                                        int BC, int h, int w, Gauss11 G) {
    __shared__ float ta[26][27], tb[26][27];
    __shared__ float red[4];
    const int bc = blockIdx.z, x0 = blockIdx.x * 16, y0 = blockIdx.y * 16;
    const float* pa = a + (size_t)bc * h * w; const float* pb = b + (size_t)bc * h * w;
    for (int i = threadIdx.x; i < 26 * 26; i += 256) {
        const int ty = i / 26, tx = i % 26, gy = y0 + ty - 5, gx = x0 + tx - 5;
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
        ta[ty][tx] = in ? pa[(size_t)gy * w + gx] : 0.f;
        tb[ty][tx] = in ? pb[(size_t)gy * w + gx] : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x % 16, ly = threadIdx.x / 16, gx = x0 + lx, gy = y0 + ly;
    float S = 0.f;
    if (gx < w && gy < h) {
        float mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
        for (int i = 0; i < 11; i++)
            for (int j = 0; j < 11; j++) {
                const float wgt = G.g[i] * G.g[j];
                const float av = ta[ly + i][lx + j], bv = tb[ly + i][lx + j];
                mu1 += wgt * av; mu2 += wgt * bv; e11 += wgt * av * av; e22 += wgt * bv * bv; e12 += wgt * av * bv;
            }
        const float C1 = 1e-4f, C2 = 9e-4f;
        const float s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
        const float A1 = 2 * mu1 * mu2 + C1, A2 = 2 * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
        const float inv = 1.f / (B1 * B2);
        S = A1 * A2 * inv;
        if (dmap) {
            const size_t n = (size_t)BC * h * w, o = (size_t)bc * h * w + (size_t)gy * w + gx;
            dmap[o] = (2 * mu1 * A2 - 2 * mu1 * A1) * inv - S * (2 * mu2 / B1 - 2 * mu2 / B2);
            dmap[n + o] = -S / B2;
            dmap[2 * n + o] = 2 * A1 * inv;
        }
    }
    S = wave_sum(S);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = S;
    __syncthreads();
    if (det_on()) {            // deterministic mode: one slot per workgroup, ordered total into copy 0
        const int nslots = gridDim.x * gridDim.y * gridDim.z;
        const int sl = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        float* slot = det_slot(0, 1, sl, nslots, 1);
        if (threadIdx.x == 0) slot[0] = red[0] + red[1] + red[2] + red[3];
        if (det_arrive_wg(0, nslots)) {
            __shared__ float dred[16];
            const float t = det_total_wg(0, nslots, 1, 0, dred);
            if (threadIdx.x == 0) ssim_sum[0] += t;
        }
        return;
    }
    // 32 slot copies (same-address f32 atomics serialise; thousands of workgroups end here), summed by the finaliser
    if (threadIdx.x == 0) atomicAdd(ssim_sum + ((blockIdx.x + blockIdx.y * 7 + blockIdx.z * 13) & 31), red[0] + red[1] + red[2] + red[3]);
}

// g[q] = scale * ( G*(dmu2) + 2 b[q] G*(dE22) + a[q] G*(dE12) )[q]      (loss = 1 - mean(S): scale = -1/N)
__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ dmap, float* __restrict__ g,
                                                        int BC, int h, int w, Gauss11 G, float scale, int accumulate) {
    __shared__ float t0[26][27], t1[26][27], t2[26][27];
    const int bc = blockIdx.z, x0 = blockIdx.x * 16, y0 = blockIdx.y * 16;
    const size_t n = (size_t)BC * h * w, po = (size_t)bc * h * w;
    for (int i = threadIdx.x; i < 26 * 26; i += 256) {
        const int ty = i / 26, tx = i % 26, gy = y0 + ty - 5, gx = x0 + tx - 5;
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;
        const size_t o = po + (size_t)gy * w + gx;
        t0[ty][tx] = in ? dmap[o] : 0.f; t1[ty][tx] = in ? dmap[n + o] : 0.f; t2[ty][tx] = in ? dmap[2 * n + o] : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x % 16, ly = threadIdx.x / 16, gx = x0 + lx, gy = y0 + ly;
    if (gx >= w || gy >= h) return;
    float f0 = 0, f1 = 0, f2 = 0;
    for (int i = 0; i < 11; i++)
        for (int j = 0; j < 11; j++) {
            const float wgt = G.g[i] * G.g[j];
            f0 += wgt * t0[ly + i][lx + j]; f1 += wgt * t1[ly + i][lx + j]; f2 += wgt * t2[ly + i][lx + j];
        }
    const size_t o = po + (size_t)gy * w + gx;
    const float v = scale * (f0 + 2.f * b[o] * f1 + a[o] * f2);
    g[o] = accumulate ? g[o] + v : v;
}

// loss/info on device.  sums (loss_reduce), ssim_sum, lpips (mean over batch, may be null).
// out[0] = 5*mse + 3*cos + ssim_l + 2*lpips (:97); out[1..7] = mse, mse_mean, mse_std, kl, cos, ssim_l, lpips
__global__ void space_loss_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ ssim_sum,
                                           const float* __restrict__ lpips, float* __restrict__ out, float n, float n_pooled,
                                           int image_space) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float mse = sums[0] / n;
    const float cosv = 1.f - sums[1] / (sqrtf(sums[2]) * sqrtf(sums[3]));
    const float ma = sums[4] / n, mb = sums[5] / n;
    const float va = fmaxf((sums[2] - n * ma * ma) / (n - 1.f), 0.f), vb = fmaxf((sums[3] - n * mb * mb) / (n - 1.f), 0.f);
    const float dm = ma - mb, ds = sqrtf(va) - sqrtf(vb);
    float kl = sums[6] / n;
    if (isnan(kl)) kl = 0.f;
    if (isinf(kl)) kl = 1.f;
    float ssim_tot = 0.f;
    if (image_space) for (int k = 0; k < 32; k++) ssim_tot += ssim_sum[k];
    const float ssim_l = image_space ? 1.f - ssim_tot / n_pooled : 0.f;
    const float lp = (image_space && lpips) ? lpips[0] : 0.f;
    out[0] = 5.f * mse + 3.f * cosv + ssim_l + 2.f * lp;
    out[1] = mse; out[2] = dm * dm; out[3] = ds * ds; out[4] = kl; out[5] = cosv; out[6] = ssim_l; out[7] = lp;
}

// d(5*mse + 3*cos)/db at full resolution + un-pooled gradient of the pooled terms:
// g[b,c,y0+y,x0+x] (+)= wgt * ( 10 (b-a)/n + 3 (-a/(|a||b|) + (a.b) b/(|a||b|^3)) + gp[b,c,y/k,x/k]/k^2 )
__global__ void space_loss_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                      const float* __restrict__ sums, const float* __restrict__ gp, float* __restrict__ g,
                                      int BC, Crop cr, int k, float n, float wgt, int accumulate) {
    const long tot = (long)BC * cr.h * cr.w;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= tot) return;
    const int x = idx % cr.w; const long r = idx / cr.w; const int y = r % cr.h; const int bc = r / cr.h;
    const size_t off = (size_t)bc * cr.H * cr.W + (size_t)(cr.y0 + y) * cr.W + cr.x0 + x;
    const float na = sqrtf(sums[2]), nb = sqrtf(sums[3]);
    const float av = a[off], bv = b[off];
    float v = 10.f * (bv - av) / n + 3.f * (-av / (na * nb) + sums[1] * bv / (na * nb * nb * nb));
    if (gp) {
        const int oh = cr.h / k, ow = cr.w / k;
        if (y / k < oh && x / k < ow) v += gp[((size_t)bc * oh + y / k) * ow + x / k] / (float)(k * k);
    }
    v *= wgt;
    g[off] = accumulate ? g[off] + v : v;
}

// y = (accumulate ? y : 0) + x * scalar[0]      (chain-rule scaling by a device scalar)
__global__ void axpy_scalar_kernel(const float* __restrict__ x, const float* __restrict__ scalar, float* __restrict__ y,
                                   long n, float extra, int accumulate) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= n) return;
    const float v = x[idx] * (scalar ? scalar[0] : 1.f) * extra;
    y[idx] = accumulate ? y[idx] + v : v;
}

// ------------------------------------------------------------------ the three attention windows of E_align in one pass each
// E_align_s2.py:185-203 evaluates space_loss on the full image and on two nested crops (AT1, AT2) of the same pair.  Window by
// window the two 100 MB images were read three times by the reduction, three times by the crop + pool and three times by the
// gradient (which also re-read and re-wrote the 100 MB gradient as an accumulator).  These forms take up to 3 windows, all inside
// window 0, and touch every pixel once.
struct Win3 { int n; int y0[3], x0[3], h[3], w[3]; };
__global__ __launch_bounds__(256) void loss_reduce3_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ sums, int B, int C, int H, int W, Win3 wn) {
    __shared__ float red[7 * 4];
    float v[3][7];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 7; i++) v[k][i] = 0.f;
    const int h0 = wn.h[0], w0 = wn.w[0];
    const long npix = (long)B * h0 * w0;
    const size_t plane = (size_t)H * W;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < npix; idx += gridDim.x * 256L) {
        const int x = idx % w0; const long r = idx / w0; const int y = r % h0; const int bb = r / h0;
        const int gy = wn.y0[0] + y, gx = wn.x0[0] + x;
        const size_t off = (size_t)bb * C * plane + (size_t)gy * W + gx;
        float t[7] = {0, 0, 0, 0, 0, 0, 0};
        float ma = -INFINITY, mb = -INFINITY;
        for (int c = 0; c < C; c++) {
            const float av = a[off + c * plane], bv = b[off + c * plane];
            const float d = av - bv;
            t[0] += d * d; t[1] += av * bv; t[2] += av * av; t[3] += bv * bv; t[4] += av; t[5] += bv;
            ma = fmaxf(ma, av); mb = fmaxf(mb, bv);
        }
        float sa = 0.f, sb = 0.f;
        for (int c = 0; c < C; c++) { sa += __expf(a[off + c * plane] - ma); sb += __expf(b[off + c * plane] - mb); }
        const float lsa = __logf(sa) + ma, lsb = __logf(sb) + mb;
        for (int c = 0; c < C; c++) {
            const float la = a[off + c * plane] - lsa, lb = b[off + c * plane] - lsb;
            t[6] += __expf(la) * (la - lb);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const bool in = k < wn.n && (unsigned)(gy - wn.y0[k]) < (unsigned)wn.h[k] && (unsigned)(gx - wn.x0[k]) < (unsigned)wn.w[k];
#pragma unroll
            for (int i = 0; i < 7; i++) v[k][i] += in ? t[i] : 0.f;
        }
    }
    for (int k = 0; k < wn.n; k++) {                      // (not offered in deterministic mode: one slot domain per launch)
        block_atomic_sums(v[k], 7, sums + (size_t)k * 16 * 8, red);
        __syncthreads();
    }
}

// loss_reduce3 for RGB images, four consecutive pixels of a row per thread: every plane is read once with 16-byte loads (the general
// form walks the channels three times with scalar loads and decodes the pixel index with 64-bit divisions)
__global__ __launch_bounds__(256) void loss_reduce3_c3v4_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 float* __restrict__ sums, int B, int H, int W, Win3 wn) {
    __shared__ float red[7 * 4];
    float v[3][7];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int i = 0; i < 7; i++) v[k][i] = 0.f;
    const unsigned h0 = wn.h[0], w4 = wn.w[0] / 4;
    const unsigned n4 = (unsigned)B * h0 * w4;
    const size_t plane = (size_t)H * W;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n4; idx += gridDim.x * 256u) {
        const unsigned x4 = idx % w4, r = idx / w4, y = r % h0, bb = r / h0;
        const int gy = wn.y0[0] + (int)y, gx = wn.x0[0] + 4 * (int)x4;
        const size_t off = (size_t)bb * 3 * plane + (size_t)gy * W + gx;
        float4 A[3], Bq[3];
#pragma unroll
        for (int c = 0; c < 3; c++) { A[c] = *(const float4*)(a + off + c * plane); Bq[c] = *(const float4*)(b + off + c * plane); }
        bool rowin[3];
#pragma unroll
        for (int k = 0; k < 3; k++) rowin[k] = k < wn.n && (unsigned)(gy - wn.y0[k]) < (unsigned)wn.h[k];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float av[3], bv[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { av[c] = ((const float*)&A[c])[e]; bv[c] = ((const float*)&Bq[c])[e]; }
            float t[7] = {0, 0, 0, 0, 0, 0, 0};
            float ma = -INFINITY, mb = -INFINITY;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float d = av[c] - bv[c];
                t[0] += d * d; t[1] += av[c] * bv[c]; t[2] += av[c] * av[c]; t[3] += bv[c] * bv[c]; t[4] += av[c]; t[5] += bv[c];
                ma = fmaxf(ma, av[c]); mb = fmaxf(mb, bv[c]);
            }
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int c = 0; c < 3; c++) { sa += __expf(av[c] - ma); sb += __expf(bv[c] - mb); }
            const float lsa = __logf(sa) + ma, lsb = __logf(sb) + mb;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float la = av[c] - lsa, lb = bv[c] - lsb;
                t[6] += __expf(la) * (la - lb);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const bool in = rowin[k] && (unsigned)(gx + e - wn.x0[k]) < (unsigned)wn.w[k];
#pragma unroll
                for (int i = 0; i < 7; i++) v[k][i] += in ? t[i] : 0.f;
            }
        }
    }
    for (int k = 0; k < wn.n; k++) {
        block_atomic_sums(v[k], 7, sums + (size_t)k * 16 * 8, red);
        __syncthreads();
    }
}

struct Pool6 { const float* src[6]; float* dst[6]; int y0[6], x0[6], h[6], w[6], k[6]; int n; };
__global__ void crop_pool6_kernel(Pool6 t, int BC, int H, int W) {
    const int e = blockIdx.y;
    const int k = t.k[e], oh = t.h[e] / k, ow = t.w[e] / k;
    const long n = (long)BC * oh * ow;
    if (k == 4 && (W & 3) == 0 && (t.x0[e] & 3) == 0) {        // the pooled cell is four aligned 16-byte rows
        for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += gridDim.x * 256L) {
            const int x = idx % ow; const long r = idx / ow; const int y = r % oh; const int bc = r / oh;
            const float* p = t.src[e] + (size_t)bc * H * W + (size_t)(t.y0[e] + y * 4) * W + t.x0[e] + x * 4;
            const float4 r0 = *(const float4*)p, r1 = *(const float4*)(p + W), r2 = *(const float4*)(p + 2 * (size_t)W), r3 = *(const float4*)(p + 3 * (size_t)W);
            // (the scalar form's summation order: row by row, left to right)
            float s = 0.f;
            s += r0.x; s += r0.y; s += r0.z; s += r0.w; s += r1.x; s += r1.y; s += r1.z; s += r1.w;
            s += r2.x; s += r2.y; s += r2.z; s += r2.w; s += r3.x; s += r3.y; s += r3.z; s += r3.w;
            t.dst[e][idx] = s / 16.f;
        }
        return;
    }
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += gridDim.x * 256L) {
        const int x = idx % ow; const long r = idx / ow; const int y = r % oh; const int bc = r / oh;
        const float* p = t.src[e] + (size_t)bc * H * W + (size_t)(t.y0[e] + y * k) * W + t.x0[e] + x * k;
        float s = 0.f;
        for (int i = 0; i < k; i++)
            for (int j = 0; j < k; j++) s += p[(size_t)i * W + j];
        t.dst[e][idx] = s / (float)(k * k);
    }
}

struct Bwd3 { const float* sums[3]; const float* gp[3]; int k[3]; float n[3], wgt[3]; };
__global__ void space_loss_bwd3_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ g,
                                       int BC, int H, int W, Win3 wn, Bwd3 q) {
    const long tot = (long)BC * wn.h[0] * wn.w[0];
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= tot) return;
    const int x = idx % wn.w[0]; const long r = idx / wn.w[0]; const int y = r % wn.h[0]; const int bc = r / wn.h[0];
    const int gy = wn.y0[0] + y, gx = wn.x0[0] + x;
    const size_t off = (size_t)bc * H * W + (size_t)gy * W + gx;
    const float av = a[off], bv = b[off];
    float tot_v = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k >= wn.n || q.wgt[k] == 0.f) continue;
        const int ly = gy - wn.y0[k], lx = gx - wn.x0[k];
        if ((unsigned)ly >= (unsigned)wn.h[k] || (unsigned)lx >= (unsigned)wn.w[k]) continue;
        const float* sm = q.sums[k];
        const float na = sqrtf(sm[2]), nb = sqrtf(sm[3]);
        float v = 10.f * (bv - av) / q.n[k] + 3.f * (-av / (na * nb) + sm[1] * bv / (na * nb * nb * nb));
        if (q.gp[k]) {
            const int kk = q.k[k], oh = wn.h[k] / kk, ow = wn.w[k] / kk;
            if (ly / kk < oh && lx / kk < ow) v += q.gp[k][((size_t)bc * oh + ly / kk) * ow + lx / kk] / (float)(kk * kk);
        }
        tot_v += v * q.wgt[k];
    }
    g[off] = tot_v;
}

// exact a / d for 0 <= a < 2^22, 1 <= d (rd = 1/d): float estimate, corrected by one
__device__ __forceinline__ int div_small(int a, int d, float rd) {
    int q = (int)((float)a * rd);
    q -= (q * d > a) ? 1 : 0;
    q += ((q + 1) * d <= a) ? 1 : 0;
    return q;
}
// The same with four consecutive pixels of a row per thread (16-byte accesses; W, window 0's x0 and width multiples of 4) and
// (row, plane) from the grid: the scalar form spends ~10 integer divisions per pixel (400 instructions for 12 bytes of traffic:
// 243 us on the three windows of a batch of eight 1024^2 images, 4x its byte floor).
__global__ __launch_bounds__(256) void space_loss_bwd3_v4_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ g,
                                                                  int H, int W, Win3 wn, Bwd3 q) {
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x >= wn.w[0]) return;
    const int bc = blockIdx.z;
    const int gy = wn.y0[0] + blockIdx.y, gx = wn.x0[0] + x;
    const size_t off = ((size_t)bc * H + gy) * W + gx;
    const float4 a4 = *(const float4*)(a + off), b4 = *(const float4*)(b + off);
    const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
    float tot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k >= wn.n || q.wgt[k] == 0.f) continue;
        const int ly = gy - wn.y0[k];
        if ((unsigned)ly >= (unsigned)wn.h[k]) continue;
        const float* sm = q.sums[k];
        const float na = sqrtf(sm[2]), nb = sqrtf(sm[3]);
        const float c0 = 10.f / q.n[k], inv = 1.f / (na * nb), c2 = sm[1] * inv / (nb * nb);
        const int kk = q.k[k];
        const float rk = 1.f / (float)kk, rkk = 1.f / (float)(kk * kk);
        const int oh = div_small(wn.h[k], kk, rk), ow = div_small(wn.w[k], kk, rk);
        const int qy = div_small(ly, kk, rk);
        const float* gpr = (q.gp[k] && qy < oh) ? q.gp[k] + ((size_t)bc * oh + qy) * ow : nullptr;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int lx = gx + e - wn.x0[k];
            if ((unsigned)lx >= (unsigned)wn.w[k]) continue;
            float v = c0 * (bv[e] - av[e]) + 3.f * (c2 * bv[e] - av[e] * inv);
            if (gpr) { const int qx = div_small(lx, kk, rk); if (qx < ow) v += gpr[qx] * rkk; }
            tot[e] += v * q.wgt[k];
        }
    }
    *(float4*)(g + off) = make_float4(tot[0], tot[1], tot[2], tot[3]);
}

// =================================================================== C ABI
static Gauss11 gauss11() {
    Gauss11 G; float s = 0.f;
    for (int i = 0; i < 11; i++) { G.g[i] = expf(-(float)((i - 5) * (i - 5)) / (2.f * 1.5f * 1.5f)); s += G.g[i]; }
    for (int i = 0; i < 11; i++) G.g[i] /= s;
    return G;
}
static Crop mk(int H, int W, int y0, int x0, int h, int w) { Crop c = {H, W, y0, x0, h, w}; return c; }

extern "C" int dge_loss_reduce(const float* a, const float* b, float* sums7, int B, int C, int H, int W, int y0, int x0,
                               int h, int w, hipStream_t s) {
    DGE_CHECK(y0 >= 0 && x0 >= 0 && y0 + h <= H && x0 + w <= W && h > 0 && w > 0, "loss_reduce: bad crop");
    long npix = (long)B * h * w;
    int grid = (int)((npix + 255) / 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(grid), dim3(256), 0, s, a, b, sums7, B, C, mk(H, W, y0, x0, h, w));
    DGE_LAUNCH_CHECK("loss_reduce");
    return 0;
}

static int win3_from(const int* wins, int nwin, int H, int W, Win3& wn) {
    DGE_CHECK(wins && nwin >= 1 && nwin <= 3, "multi-window loss: 1..3 windows");
    wn.n = nwin;
    for (int k = 0; k < 3; k++) {
        const int j = k < nwin ? k : 0;
        wn.y0[k] = wins[4 * j]; wn.x0[k] = wins[4 * j + 1]; wn.h[k] = wins[4 * j + 2]; wn.w[k] = wins[4 * j + 3];
        DGE_CHECK(wn.y0[k] >= wn.y0[0] && wn.x0[k] >= wn.x0[0] && wn.y0[k] + wn.h[k] <= wn.y0[0] + wn.h[0] && wn.x0[k] + wn.w[k] <= wn.x0[0] + wn.w[0]
                  && wn.h[k] > 0 && wn.w[k] > 0 && wn.y0[0] >= 0 && wn.x0[0] >= 0 && wn.y0[0] + wn.h[0] <= H && wn.x0[0] + wn.w[0] <= W,
                  "multi-window loss: window %d must lie inside window 0, window 0 inside the image", k);
    }
    return 0;
}
// dge_loss_reduce for up to 3 windows (y0, x0, h, w each; all inside window 0) in one pass: sums [nwin][16][8] slot copies, pre-zeroed
extern "C" int dge_loss_reduce3(const float* a, const float* b, float* sums, int B, int C, int H, int W, const int* wins, int nwin,
                                hipStream_t s) {
    Win3 wn;
    if (win3_from(wins, nwin, H, W, wn)) return -1;
    DGE_CHECK(!dge_get_deterministic(), "loss_reduce3 is not offered in deterministic mode (run dge_loss_reduce per window)");
    const long npix = (long)B * wn.h[0] * wn.w[0];
    int grid = (int)((npix + 255) / 256); if (grid > 2048) grid = 2048;
    if (C == 3 && W % 4 == 0 && wn.x0[0] % 4 == 0 && wn.w[0] % 4 == 0 && npix / 4 < (1L << 31)) {
        int g4 = (int)((npix / 4 + 255) / 256); if (g4 > 2048) g4 = 2048;
        hipLaunchKernelGGL(loss_reduce3_c3v4_kernel, dim3(g4), dim3(256), 0, s, a, b, sums, B, H, W, wn);
    } else
    hipLaunchKernelGGL(loss_reduce3_kernel, dim3(grid), dim3(256), 0, s, a, b, sums, B, C, H, W, wn);
    DGE_LAUNCH_CHECK("loss_reduce3");
    return 0;
}
// dge_crop_pool of up to 6 (source, window, pooling factor) entries in one launch; entry e: src[e] [BC,H,W] -> dst[e] [BC,h/k,w/k]
extern "C" int dge_crop_pool_multi(const float* const* src, float* const* dst, const int* wins, const int* ks, int n, int BC, int H, int W,
                                   hipStream_t s) {
    DGE_CHECK(n >= 1 && n <= 6 && src && dst && wins && ks, "crop_pool_multi: 1..6 entries");
    Pool6 t; t.n = n;
    long mx = 1;
    for (int e = 0; e < n; e++) {
        t.src[e] = src[e]; t.dst[e] = dst[e];
        t.y0[e] = wins[4 * e]; t.x0[e] = wins[4 * e + 1]; t.h[e] = wins[4 * e + 2]; t.w[e] = wins[4 * e + 3]; t.k[e] = ks[e];
        DGE_CHECK(t.k[e] >= 1 && t.h[e] % t.k[e] == 0 && t.w[e] % t.k[e] == 0 && t.y0[e] >= 0 && t.x0[e] >= 0 && t.y0[e] + t.h[e] <= H && t.x0[e] + t.w[e] <= W,
                  "crop_pool_multi: bad entry %d", e);
        const long ne = (long)BC * (t.h[e] / t.k[e]) * (t.w[e] / t.k[e]);
        if (ne > mx) mx = ne;
    }
    long blocks = (mx + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(crop_pool6_kernel, dim3((unsigned)blocks, n), dim3(256), 0, s, t, BC, H, W);
    DGE_LAUNCH_CHECK("crop_pool_multi");
    return 0;
}
// dge_space_loss_bwd for up to 3 windows in one pass: g (window 0's pixels) = sum_k weight[k] * (window k's gradient); g is WRITTEN
// (no accumulation); pixels of the image outside window 0 are not touched.  weight[k] = 0 leaves window k out.
extern "C" int dge_space_loss_bwd3(const float* a, const float* b, const float* const* sums7, const float* const* g_pooled, float* g,
                                   int BC, int H, int W, const int* wins, const int* ks, const float* n, const float* weight, int nwin,
                                   hipStream_t s) {
    Win3 wn;
    if (win3_from(wins, nwin, H, W, wn)) return -1;
    Bwd3 q;
    for (int k = 0; k < 3; k++) {
        const int j = k < nwin ? k : 0;
        q.sums[k] = sums7[j]; q.gp[k] = g_pooled ? g_pooled[j] : nullptr; q.k[k] = ks[j]; q.n[k] = n[j]; q.wgt[k] = k < nwin ? weight[j] : 0.f;
        DGE_CHECK(q.sums[k] && q.k[k] >= 1 && q.n[k] > 0.f, "space_loss_bwd3: bad window %d", k);
    }
    const long tot = (long)BC * wn.h[0] * wn.w[0];
    if (W % 4 == 0 && wn.x0[0] % 4 == 0 && wn.w[0] % 4 == 0 && wn.h[0] <= 65535 && BC <= 65535 && wn.h[0] < (1 << 22) && wn.w[0] < (1 << 22))
        hipLaunchKernelGGL(space_loss_bwd3_v4_kernel, dim3((unsigned)((wn.w[0] / 4 + 255) / 256), wn.h[0], BC), dim3(256), 0, s, a, b, g, H, W, wn, q);
    else
        hipLaunchKernelGGL(space_loss_bwd3_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, a, b, g, BC, H, W, wn, q);
    DGE_LAUNCH_CHECK("space_loss_bwd3");
    return 0;
}

extern "C" int dge_crop_pool(const float* src, float* dst, int BC, int H, int W, int y0, int x0, int h, int w, int k,
                             hipStream_t s) {
    DGE_CHECK(k >= 1 && h % k == 0 && w % k == 0, "crop_pool: %dx%d not divisible by %d", h, w, k);
    const long n = (long)BC * (h / k) * (w / k);
    hipLaunchKernelGGL(crop_pool_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, BC, mk(H, W, y0, x0, h, w), k);
    DGE_LAUNCH_CHECK("crop_pool");
    return 0;
}

// skimage.measure.compare_ssim(X, Y, data_range=R, multichannel=True) as comparing-baseline.py:25 calls it (defaults: 7x7 uniform
// window, K1 = 0.01, K2 = 0.03, sample covariance i.e. cov_norm = 49/48, mean of the SSIM map with a 3-pixel border cropped, mean
// over the channels): sums[bc] (pre-zeroed) += sum of the map over the interior of plane bc.  `scale`/`shift` map the stored pixel
// values to the metric's scale (x*scale + shift), R is the data range there.
__global__ __launch_bounds__(256) void ssim_box7_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ sums,
                                                         int h, int w, float scale, float shift, float R) {
    __shared__ float ta[22][23], tb[22][23];
    __shared__ float red[4];
    const int bc = blockIdx.z, x0 = blockIdx.x * 16, y0 = blockIdx.y * 16;       // output (interior) coordinates: pixel (y+3, x+3)
    const float* pa = a + (size_t)bc * h * w; const float* pb = b + (size_t)bc * h * w;
    for (int i = threadIdx.x; i < 22 * 22; i += 256) {
        const int ty = i / 22, tx = i % 22, gy = y0 + ty, gx = x0 + tx;
        const bool in = gy < h && gx < w;
        ta[ty][tx] = in ? pa[(size_t)gy * w + gx] * scale + shift : 0.f;
        tb[ty][tx] = in ? pb[(size_t)gy * w + gx] * scale + shift : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x % 16, ly = threadIdx.x / 16;
    float S = 0.f;
    if (x0 + lx < w - 6 && y0 + ly < h - 6) {
        float sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (int i = 0; i < 7; i++)
#pragma unroll
            for (int j = 0; j < 7; j++) {
                const float av = ta[ly + i][lx + j], bv = tb[ly + i][lx + j];
                sx += av; sy += bv; sxx += av * av; syy += bv * bv; sxy += av * bv;
            }
        const float inv = 1.f / 49.f, cn = 49.f / 48.f;
        const float ux = sx * inv, uy = sy * inv;
        const float vx = cn * (sxx * inv - ux * ux), vy = cn * (syy * inv - uy * uy), vxy = cn * (sxy * inv - ux * uy);
        const float C1 = (0.01f * R) * (0.01f * R), C2 = (0.03f * R) * (0.03f * R);
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
    S = wave_sum(S);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = S;
    __syncthreads();
    if (det_on()) {            // deterministic mode: domain = plane bc, one slot per tile
        const int nslots = gridDim.x * gridDim.y, sl = blockIdx.x + gridDim.x * blockIdx.y;
        float* slot = det_slot(bc, gridDim.z, sl, nslots, 1);
        if (threadIdx.x == 0) slot[0] = red[0] + red[1] + red[2] + red[3];
        if (det_arrive_wg(bc, nslots)) {
            __shared__ float dred[16];
            const float t = det_total_wg(bc, nslots, 1, 0, dred);
            if (threadIdx.x == 0) sums[bc] += t;
        }
        return;
    }
    if (threadIdx.x == 0) atomicAdd(sums + bc, red[0] + red[1] + red[2] + red[3]);
}

extern "C" int dge_ssim_box7(const float* a, const float* b, float* sums, int BC, int h, int w, float scale, float shift, float data_range,
                             hipStream_t s) {
    DGE_CHECK(BC >= 1 && h >= 7 && w >= 7 && data_range > 0.f, "ssim_box7: planes must be at least 7x7 (got %dx%d)", h, w);
    hipLaunchKernelGGL(ssim_box7_kernel, dim3((w - 6 + 15) / 16, (h - 6 + 15) / 16, BC), dim3(256), 0, s, a, b, sums, h, w, scale, shift,
                       data_range);
    DGE_LAUNCH_CHECK("ssim_box7");
    return 0;
}

extern "C" int dge_ssim_fwd(const float* a, const float* b, float* ssim_sum, float* dmap, int BC, int h, int w, hipStream_t s) {
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3((w + 15) / 16, (h + 15) / 16, BC), dim3(256), 0, s, a, b, ssim_sum, dmap, BC, h, w, gauss11());
    DGE_LAUNCH_CHECK("ssim_fwd");
    return 0;
}

extern "C" int dge_ssim_bwd(const float* a, const float* b, const float* dmap, float* g, int BC, int h, int w, float scale,
                            int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3((w + 15) / 16, (h + 15) / 16, BC), dim3(256), 0, s, a, b, dmap, g, BC, h, w, gauss11(), scale, accumulate);
    DGE_LAUNCH_CHECK("ssim_bwd");
    return 0;
}

extern "C" int dge_space_loss_finalize(const float* sums7, const float* ssim_sum, const float* lpips, float* out8, float n,
                                       float n_pooled, int image_space, hipStream_t s) {
    hipLaunchKernelGGL(space_loss_finalize_kernel, dim3(1), dim3(64), 0, s, sums7, ssim_sum, lpips, out8, n, n_pooled, image_space);
    DGE_LAUNCH_CHECK("space_loss_finalize");
    return 0;
}

extern "C" int dge_space_loss_bwd(const float* a, const float* b, const float* sums7, const float* g_pooled, float* g,
                                  int BC, int H, int W, int y0, int x0, int h, int w, int k, float n, float weight,
                                  int accumulate, hipStream_t s) {
    const long tot = (long)BC * h * w;
    hipLaunchKernelGGL(space_loss_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, a, b, sums7, g_pooled, g, BC,
                       mk(H, W, y0, x0, h, w), k, n, weight, accumulate);
    DGE_LAUNCH_CHECK("space_loss_bwd");
    return 0;
}

extern "C" int dge_axpy_scalar(const float* x, const float* scalar, float* y, long n, float extra, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(axpy_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, scalar, y, n, extra, accumulate);
    DGE_LAUNCH_CHECK("axpy_scalar");
    return 0;
}
