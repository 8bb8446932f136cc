// Data-gradient path of the StyleGAN2 synthesis network w.r.t. the latent codes wp
// (the only gradient the encoder needs: E_align_s2.py:160,204 - G's weights receive no update).
// The heavy part (conv data gradients) reuses conv_igemm with DGE_PACK_DGRAD /
// DGE_PACK_UPFOLD_DGRAD weights; the kernels here are the HBM-bound glue around it.
#include "common.h"
#include "../../include/dge_hip.h"

// Forward (stylegan2_generator.py:905-921): z = yraw*d + noise*ns + bias ; x = lrelu(z)*gain.
// Given g_x: g_z = g_x*gain*lrelu'(x) ; g_y = g_z*d[b,c] (gradient w.r.t. yraw, fed to the dgrad conv)
// R[b,c,:] += { sum g_z*z, sum g_z*noise, sum g_z }   (for the demodulation gradient)
template <typename T>
__global__ __launch_bounds__(256) void modconv_bwd_prep_kernel(const T* __restrict__ gx, const T* __restrict__ x,
                                                                const float* __restrict__ d, const float* __restrict__ noise,
                                                                T* __restrict__ gy, float* __restrict__ R, int HW, int C,
                                                                int noise_bstride, float gain) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 3 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    float s[3][EP], dd[EP];
#pragma unroll
    for (int e = 0; e < EP; e++) { s[0][e] = s[1][e] = s[2][e] = 0.f; dd[e] = d ? d[(size_t)b * C + chunk * EP + e] : 1.f; }
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            const size_t o = ((size_t)b * HW + p) * C + chunk * EP;
            float g[EP], xv[EP];
            unpack16(*(const uint4*)(gx + o), g, (T*)nullptr);
            unpack16(*(const uint4*)(x + o), xv, (T*)nullptr);
            const float nz = noise ? noise[(size_t)b * noise_bstride + p] : 0.f;
#pragma unroll
            for (int e = 0; e < EP; e++) {
                const bool pos = xv[e] > 0.f;
                const float gz = g[e] * gain * (pos ? 1.f : 0.2f);
                const float z = pos ? xv[e] / gain : xv[e] / (0.2f * gain);
                s[0][e] += gz * z; s[1][e] += gz * nz; s[2][e] += gz;
                g[e] = gz * dd[e];
            }
            *(uint4*)(gy + o) = pack16(g, (T*)nullptr);
        }
    }
    if (R) block_chan_flush<EP, 3>(s, cpt, ppi, R + (size_t)b * C * 3, C, red);
}

// t[b,o] = -(R0 - ns*R1 - bias[o]*bscale*R2) * d[b,o]^2   (= g_d * dd/du up to the 2 s_i wsq factor)
__global__ void demod_bwd_kernel(const float* __restrict__ R, const float* __restrict__ d, const float* __restrict__ bias,
                                 const float* __restrict__ ns, float* __restrict__ t, int B, int C, float bscale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int o = idx % C;
    const float dv = d[idx];
    t[idx] = -(R[(size_t)idx * 3] - (ns ? ns[0] : 0.f) * R[(size_t)idx * 3 + 1] - bias[o] * bscale * R[(size_t)idx * 3 + 2]) * dv * dv;
}

// The same from the sums of the FUSED tail backward (ConvParams::prep, dge_torgb_bwd_prep): P[slot][b,o,:] = (sum g_z*(z - ns*noise),
// sum g_z), i.e. R0 - ns*R1 and R2 of the line above;  t[b,o] = -(P0 - bias[o]*bscale*P1) * d[b,o]^2.  Slot copies are added here.
__global__ void demod_bwd_prep_kernel(const float* __restrict__ P, int nslot, const float* __restrict__ d, const float* __restrict__ bias,
                                      float* __restrict__ t, int B, int C, float bscale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const float2 pp = sum_slot_pairs(P + (size_t)idx * 2, (size_t)B * C * 2, nslot);
    const float dv = d[idx];
    t[idx] = -(pp.x - bias[idx % C] * bscale * pp.y) * dv * dv;
}

// Top of the synthesis backward in one pass: the last layer's output x feeds only the last toRGB (image = rgb + up(prev),
// stylegan2_generator.py:515-522), so its gradient is the toRGB adjoint, t_i = wscale * sum_c g[b,c,p] Wrgb[c,i], g = t_i * s[b,i],
// gs[b,i] += sum_p t_i x[b,p,i] - immediately followed by the layer's own tail backward (:908-921): g_z = g * gain * lrelu'(x),
// P[b,i,:] += (sum g_z*(z - ns*noise), sum g_z).  (dge_torgb_bwd + dge_modconv_bwd_prep without the gradient round trip.)
template <typename T>
__global__ __launch_bounds__(256) void torgb_bwd_prep_kernel(const float* __restrict__ gimg, const T* __restrict__ x,
                                                              const float* __restrict__ wrgb, const float* __restrict__ sty,
                                                              const float* __restrict__ noise, const float* __restrict__ nsp,
                                                              T* __restrict__ gz_out, float* __restrict__ gs, float* __restrict__ P,
                                                              int HW, int C, float wscale, float gain, int noise_bstride) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 2 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    float w0[EP], w1[EP], w2[EP], sv[EP], sg[1][EP], sp[2][EP];
#pragma unroll
    for (int e = 0; e < EP; e++) {
        const int i = chunk * EP + e;
        w0[e] = wrgb[i] * wscale; w1[e] = wrgb[C + i] * wscale; w2[e] = wrgb[2 * C + i] * wscale;
        sv[e] = sty[(size_t)b * C + i]; sg[0][e] = 0.f; sp[0][e] = 0.f; sp[1][e] = 0.f;
    }
    const float ns = (noise && nsp) ? nsp[0] : 0.f;
    const float gpos = gain, gneg = 0.2f * gain, zpos = 1.f / gain, zneg = 1.f / (0.2f * gain);
    const float* gb = gimg + (size_t)b * 3 * HW;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            const float g0 = gb[p], g1 = gb[HW + p], g2 = gb[2 * HW + p];
            const float nzs = noise ? ns * noise[(size_t)b * noise_bstride + p] : 0.f;
            const size_t o = ((size_t)b * HW + p) * C + chunk * EP;
            float xv[EP], out[EP];
            unpack16(*(const uint4*)(x + o), xv, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < EP; e++) {
                const float t = g0 * w0[e] + g1 * w1[e] + g2 * w2[e];
                sg[0][e] += t * xv[e];
                const bool pos = xv[e] > 0.f;
                const float gz = t * sv[e] * (pos ? gpos : gneg);
                const float zt = xv[e] * (pos ? zpos : zneg) - nzs;
                sp[0][e] = fmaf(gz, zt, sp[0][e]); sp[1][e] += gz;
                out[e] = gz;
            }
            *(uint4*)(gz_out + o) = pack16(out, (T*)nullptr);
        }
    }
    block_chan_flush<EP, 1>(sg, cpt, ppi, gs + (size_t)b * C, C, red);
    block_chan_flush<EP, 2>(sp, cpt, ppi, P + (size_t)b * C * 2, C, red);
}

// y[b*ldy + k*incy] (+)= scale * mul[b,k] * sum_o x[b*ldx + o*incx] * W[o,k]
// block = 16 waves: lanes own 64 consecutive k (coalesced W rows), waves split the o range, LDS combine.
__global__ __launch_bounds__(1024) void linear_t_kernel(const float* __restrict__ x, int ldx, int incx, const float* __restrict__ W,
                                const float* __restrict__ mul, float* __restrict__ y, int ldy, int incy, int B, int O, int K,
                                float scale, int accumulate) {
    // The grid is small ((K/64) x B workgroups) and every wave walks its share of O serially: the loop is bound by load
    // latency, so 16 waves split O and each keeps 8 independent loads in flight.
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane, b = blockIdx.y;
    const int per = (O + 15) / 16, o0 = wave * per, o1 = min(O, o0 + per);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    if (k < K) {
        const float* xr = x + (size_t)b * ldx;
        int o = o0;
        for (; o + 7 < o1; o += 8) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] += xr[(size_t)(o + j) * incx] * W[(size_t)(o + j) * K + k];
        }
        for (; o < o1; o++) acc[0] += xr[(size_t)o * incx] * W[(size_t)o * K + k];
    }
    part[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (wave == 0 && k < K) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) s += part[j][lane];
        s *= scale;
        if (mul) s *= mul[(size_t)b * K + k];
        float* yp = y + (size_t)b * ldy + (size_t)k * incy;
        *yp = accumulate ? *yp + s : s;
    }
}

// toRGB backward: t_i = wscale * sum_c g[b,c,p] Wrgb[c,i];  gx[b,p,i] = t_i * s[b,i];  gs[b,i] += sum_p t_i x[b,p,i]
template <typename T>
__global__ __launch_bounds__(256) void torgb_bwd_kernel(const float* __restrict__ gimg, const T* __restrict__ x,
                                                         const float* __restrict__ wrgb, const float* __restrict__ sty,
                                                         T* __restrict__ gx, float* __restrict__ gs, int HW, int C, float wscale) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    float w0[EP], w1[EP], w2[EP], sv[EP], s[1][EP];
#pragma unroll
    for (int e = 0; e < EP; e++) {
        const int i = chunk * EP + e;
        w0[e] = wrgb[i] * wscale; w1[e] = wrgb[C + i] * wscale; w2[e] = wrgb[2 * C + i] * wscale;
        sv[e] = sty[(size_t)b * C + i]; s[0][e] = 0.f;
    }
    const float* gb = gimg + (size_t)b * 3 * HW;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            const float g0 = gb[p], g1 = gb[HW + p], g2 = gb[2 * HW + p];
            const size_t o = ((size_t)b * HW + p) * C + chunk * EP;
            float xv[EP], out[EP];
            unpack16(*(const uint4*)(x + o), xv, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < EP; e++) {
                const float t = g0 * w0[e] + g1 * w1[e] + g2 * w2[e];
                s[0][e] += t * xv[e];
                out[e] = t * sv[e];
            }
            *(uint4*)(gx + o) = pack16(out, (T*)nullptr);
        }
    }
    block_chan_flush<EP, 1>(s, cpt, ppi, gs + (size_t)b * C, C, red);
}

// adjoint of the skip-branch 2x FIR upsample (UpsamplingLayer scale 2, :603-615):
// gprev[m,n] = sum_{a,b in {-1,0,1,2}} w(a) w(b) g[2m+a, 2n+b],  w = {.25,.75,.75,.25}
__global__ void up2_bwd_kernel(const float* __restrict__ g, float* __restrict__ gprev, int BC, int h, int w) {
    const long n = (long)BC * h * w;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= n) return;
    const int x = idx % w; const long r = idx / w; const int y = r % h; const int bc = r / h;
    const float wt[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    const float* gp = g + (size_t)bc * 4 * h * w;
    const int H2 = 2 * h, W2 = 2 * w;
    float s = 0.f;
    for (int a = 0; a < 4; a++) {
        const int yy = 2 * y + a - 1;
        if (yy < 0 || yy >= H2) continue;
        for (int bq = 0; bq < 4; bq++) {
            const int xx = 2 * x + bq - 1;
            if (xx < 0 || xx >= W2) continue;
            s += wt[a] * wt[bq] * gp[(size_t)yy * W2 + xx];
        }
    }
    gprev[idx] = s;
}

// the same for w % 4 == 0: one thread = 4 consecutive outputs of a row; the 4 x 10 window comes in as four 16-byte loads per row
// (the scalar form issued 16 four-byte loads per output: 79 us at 1024^2 -> 512^2, batch 8, for 125 MB)
__global__ __launch_bounds__(256) void up2_bwd_v4_kernel(const float* __restrict__ g, float* __restrict__ gprev, int BC, int h, int w) {
    const int wq = w >> 2;
    const long n = (long)BC * h * wq;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= n) return;
    const int xq = idx % wq; const long r = idx / wq; const int y = r % h; const int bc = r / h;
    const int H2 = 2 * h, W2 = 2 * w, X = 8 * xq;          // first input column of B
    const float* gp = g + (size_t)bc * 4 * h * w;
    const float wt[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    float c[10];
#pragma unroll
    for (int j = 0; j < 10; j++) c[j] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int yy = 2 * y + a - 1;
        if (yy < 0 || yy >= H2) continue;
        const float* row = gp + (size_t)yy * W2 + X;
        const float4 B = *(const float4*)row, C4 = *(const float4*)(row + 4);
        const float Aw = X > 0 ? row[-1] : 0.f;
        const float Dx = X + 8 < W2 ? row[8] : 0.f;
        c[0] = fmaf(wt[a], Aw, c[0]);
        c[1] = fmaf(wt[a], B.x, c[1]); c[2] = fmaf(wt[a], B.y, c[2]); c[3] = fmaf(wt[a], B.z, c[3]); c[4] = fmaf(wt[a], B.w, c[4]);
        c[5] = fmaf(wt[a], C4.x, c[5]); c[6] = fmaf(wt[a], C4.y, c[6]); c[7] = fmaf(wt[a], C4.z, c[7]); c[8] = fmaf(wt[a], C4.w, c[8]);
        c[9] = fmaf(wt[a], Dx, c[9]);
    }
    float4 o;
    o.x = 0.25f * c[0] + 0.75f * c[1] + 0.75f * c[2] + 0.25f * c[3];
    o.y = 0.25f * c[2] + 0.75f * c[3] + 0.75f * c[4] + 0.25f * c[5];
    o.z = 0.25f * c[4] + 0.75f * c[5] + 0.75f * c[6] + 0.25f * c[7];
    o.w = 0.25f * c[6] + 0.75f * c[7] + 0.75f * c[8] + 0.25f * c[9];
    *(float4*)(gprev + (size_t)idx * 4) = o;
}

// =================================================================== C ABI

extern "C" int dge_modconv_bwd_prep(const void* gx, const void* x, const float* d, const float* noise, void* gy, float* R,
                                    int B, int HW, int C, int noise_batch, float gain, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0 && C / ep <= 256 && 256 % (C / ep) == 0, "modconv_bwd_prep: unsupported channel count %d", C);
    dim3 grid(dge_stream_grid(HW, 256 / (C / ep), B), B);
    const int nbs = noise_batch > 1 ? HW : 0;
    if (dtype == DGE_BF16)
        hipLaunchKernelGGL(modconv_bwd_prep_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)gx, (const bf16_t*)x, d, noise, (bf16_t*)gy, R, HW, C, nbs, gain);
    else
        hipLaunchKernelGGL(modconv_bwd_prep_kernel<float>, grid, dim3(256), 0, s, (const float*)gx, (const float*)x, d, noise, (float*)gy, R, HW, C, nbs, gain);
    DGE_LAUNCH_CHECK("modconv_bwd_prep");
    return 0;
}

extern "C" int dge_demod_bwd(const float* R, const float* d, const float* bias, const float* noise_strength, float* t, int B,
                             int C, float bscale, hipStream_t s) {
    hipLaunchKernelGGL(demod_bwd_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, R, d, bias, noise_strength, t, B, C, bscale);
    DGE_LAUNCH_CHECK("demod_bwd");
    return 0;
}

// ------------------------------------------------------------------ FIR^T of the up layer, stored t-grid-to-depth
// Up layer (stylegan2_generator.py:879-896): t = conv_transpose2d(x, stride 2) on the (2H+1)^2 grid, y[v] = sum_j k1[j] t[v + j - 1]
// per axis (k1 = [1,3,3,1]/4, pad 1, :603-615).  Adjoint of the filter: g_t[u] = sum_j k1[j] g_y[u - j + 1] (g_y zero outside
// [0, 2H)).  The data-gradient conv wants the four phases of g_t around coarse pixel m side by side (dge_conv_desc.in_t2d):
//   Z[b, m, (py,px)*C + c] = scale[b,c] * g_t[b, 2m + p, c],   m in [0,H] x [0,W],   zero where 2m + p > 2H (2W).
// One thread = one 16-byte channel chunk of one m: the 5 x 5 window of g_y it needs is filtered separably (rows first).
template <typename T>
__global__ __launch_bounds__(256) void fir_t2d_kernel(const T* __restrict__ g, const float* __restrict__ scale, T* __restrict__ z,
                                                       int H, int W, int C, long total) {
    constexpr int EP = Elem<T>::PER16;
    const int cpt = C / EP;
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= total) return;
    const int ch = idx % cpt; long r = idx / cpt;
    const int mx = r % (W + 1); r /= (W + 1);
    const int my = r % (H + 1); const int b = r / (H + 1);
    const int FH = 2 * H, FW = 2 * W;
    const T* gb = g + (size_t)b * FH * FW * C + ch * EP;
    const float k1[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    // horizontal pass: for each of the 5 rows v = 2my-2 .. 2my+2, h[px][row] = sum_j k1[j] g[v][2mx + px + 1 - j]
    float h0[5][EP], h1[5][EP];
#pragma unroll
    for (int rr = 0; rr < 5; rr++) {
        const int v = 2 * my - 2 + rr;
        float f[5][EP];
#pragma unroll
        for (int cc = 0; cc < 5; cc++) {
            const int u = 2 * mx - 2 + cc;
            uint4 val = make_uint4(0, 0, 0, 0);
            if ((unsigned)v < (unsigned)FH && (unsigned)u < (unsigned)FW) val = *(const uint4*)(gb + ((size_t)v * FW + u) * C);
            unpack16(val, f[cc], (T*)nullptr);
        }
        // px = 0: columns 2mx+1, 2mx, 2mx-1, 2mx-2 = f[3], f[2], f[1], f[0];   px = 1: 2mx+2 .. 2mx-1 = f[4], f[3], f[2], f[1]
#pragma unroll
        for (int e = 0; e < EP; e++) {
            h0[rr][e] = k1[0] * f[3][e] + k1[1] * f[2][e] + k1[2] * f[1][e] + k1[3] * f[0][e];
            h1[rr][e] = k1[0] * f[4][e] + k1[1] * f[3][e] + k1[2] * f[2][e] + k1[3] * f[1][e];
        }
    }
    float sc[EP];
#pragma unroll
    for (int e = 0; e < EP; e++) sc[e] = scale ? scale[(size_t)b * C + ch * EP + e] : 1.f;
    T* zb = z + (((size_t)b * (H + 1) + my) * (W + 1) + mx) * 4 * C + ch * EP;
    const bool vy1 = 2 * my + 1 <= 2 * H, vx1 = 2 * mx + 1 <= 2 * W;     // (phase 0 always lies on the grid)
#pragma unroll
    for (int py = 0; py < 2; py++) {
        // rows 2my+py+1-j, j = 0..3  = window rows (py + 3 - j)
        float o0[EP], o1[EP];
#pragma unroll
        for (int e = 0; e < EP; e++) {
            o0[e] = sc[e] * (k1[0] * h0[py + 3][e] + k1[1] * h0[py + 2][e] + k1[2] * h0[py + 1][e] + k1[3] * h0[py][e]);
            o1[e] = sc[e] * (k1[0] * h1[py + 3][e] + k1[1] * h1[py + 2][e] + k1[2] * h1[py + 1][e] + k1[3] * h1[py][e]);
        }
        const bool vy = py == 0 || vy1;
        if (!vy) {
#pragma unroll
            for (int e = 0; e < EP; e++) { o0[e] = 0.f; o1[e] = 0.f; }
        }
        if (!vx1) {
#pragma unroll
            for (int e = 0; e < EP; e++) o1[e] = 0.f;
        }
        *(uint4*)(zb + (size_t)(py * 2 + 0) * C) = pack16(o0, (T*)nullptr);
        *(uint4*)(zb + (size_t)(py * 2 + 1) * C) = pack16(o1, (T*)nullptr);
    }
}

extern "C" int dge_fir_t2d(const void* g, const float* scale, void* z, int B, int H, int W, int C, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(g && z && B > 0 && H > 0 && W > 0 && C > 0 && C % ep == 0, "fir_t2d: bad arguments");
    const long total = (long)B * (H + 1) * (W + 1) * (C / ep);
    DGE_CHECK((total + 255) / 256 < (1L << 31), "fir_t2d: grid too large");
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == DGE_BF16) hipLaunchKernelGGL(fir_t2d_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)g, scale, (bf16_t*)z, H, W, C, total);
    else hipLaunchKernelGGL(fir_t2d_kernel<float>, grid, dim3(256), 0, s, (const float*)g, scale, (float*)z, H, W, C, total);
    DGE_LAUNCH_CHECK("fir_t2d");
    return 0;
}

// ------------------------------------------------------------------ every style gradient of a synthesis backward in one launch
// Per modulated block the backward ends in three small steps (stylegan2_generator.py:858-864,908-909 differentiated):
//   t[b,o]   = -(P0 - bias[o]*bscale*P1) * d[b,o]^2                      (demodulation gradient from the fused tail sums)
//   g_s[b,c] = st[b,c,0] + s[b,c] * sum_o t[b,o] * wsq[o,c]              (direct part from the data-gradient statistics)
//   g_wp[b,row,:] += wscale * sum_c g_s[b,c] * Wstyle[c,:]               (transposed style DenseBlock)
// (toRGB blocks: g_s comes from dge_torgb_bwd).  None of it feeds the data-gradient chain, so the per-layer launches (4 x 17 + 9
// of 5 us each) are replaced by ONE launch at the end of the backward over a table of the blocks, passed by value.
// Two blocks may share a row (layer 2k+1 and toRGB k): their contributions meet in g_wp (pre-zeroed) through atomics; a sum of
// two terms is order independent.
struct S2GradTable { dge_s2_grad_entry e[32]; };
__global__ __launch_bounds__(512) void s2_style_grads_kernel(S2GradTable tab, float* __restrict__ g_wp, int B, int nrows, int K, float wscale) {
    __shared__ float t[512], gs[512];
    const dge_s2_grad_entry e = tab.e[blockIdx.x];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (e.P) {
        for (int o = tid; o < e.out_c; o += 512) {
            const int idx = b * e.out_c + o;
            const float2 pp = sum_slot_pairs(e.P + (size_t)idx * 2, (size_t)B * e.out_c * 2, e.nslot_p);
            const float dv = e.d[idx];
            t[o] = -(pp.x - e.bias[o] * e.bscale * pp.y) * dv * dv;
        }
        __syncthreads();
        for (int c = tid; c < e.in_c; c += 512) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            int o = 0;
            for (; o + 3 < e.out_c; o += 4) {
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = fmaf(t[o + j], e.wsq[(size_t)(o + j) * e.in_c + c], acc[j]);
            }
            for (; o < e.out_c; o++) acc[0] = fmaf(t[o], e.wsq[(size_t)o * e.in_c + c], acc[0]);
            const int idx = b * e.in_c + c;
            const float2 ss = sum_slot_pairs(e.st + (size_t)idx * 2, (size_t)B * e.in_c * 2, e.nslot_s);
            gs[c] = ss.x + e.s[idx] * ((acc[0] + acc[1]) + (acc[2] + acc[3]));
        }
    } else {
        for (int c = tid; c < e.in_c; c += 512) gs[c] = e.gs[b * e.in_c + c];
    }
    __syncthreads();
    for (int k = tid; k < K; k += 512) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int c = 0;
        for (; c + 3 < e.in_c; c += 4) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = fmaf(gs[c + j], e.wstyle[(size_t)(c + j) * K + k], acc[j]);
        }
        for (; c < e.in_c; c++) acc[0] = fmaf(gs[c], e.wstyle[(size_t)c * K + k], acc[0]);
        atomicAdd(g_wp + ((size_t)b * nrows + e.row) * K + k, wscale * ((acc[0] + acc[1]) + (acc[2] + acc[3])));
    }
}

extern "C" int dge_s2_style_grads(const dge_s2_grad_entry* entries, int n, float* g_wp, int B, int nrows, int K, float wscale,
                                  hipStream_t s) {
    DGE_CHECK(entries && n >= 1 && n <= 32 && B >= 1 && B <= 65535, "s2_style_grads: 1..32 blocks");
    S2GradTable tab;
    for (int i = 0; i < n; i++) {
        const dge_s2_grad_entry& e = entries[i];
        DGE_CHECK(e.in_c >= 1 && e.in_c <= 512 && e.row >= 0 && e.row < nrows && e.wstyle, "s2_style_grads: bad block %d", i);
        DGE_CHECK(e.P ? (e.st && e.d && e.s && e.bias && e.wsq && e.out_c >= 1 && e.out_c <= 512 && e.nslot_p >= 1 && e.nslot_s >= 1) : e.gs != nullptr,
                  "s2_style_grads: block %d is neither a conv block (P, st, d, s, bias, wsq) nor a toRGB block (gs)", i);
        tab.e[i] = e;
    }
    hipLaunchKernelGGL(s2_style_grads_kernel, dim3(n, B), dim3(512), 0, s, tab, g_wp, B, nrows, K, wscale);
    DGE_LAUNCH_CHECK("s2_style_grads");
    return 0;
}

extern "C" int dge_demod_bwd_prep(const float* P, int nslot, const float* d, const float* bias, float* t, int B, int C, float bscale,
                                  hipStream_t s) {
    DGE_CHECK(nslot >= 1, "demod_bwd_prep: bad slot count");
    hipLaunchKernelGGL(demod_bwd_prep_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, P, nslot, d, bias, t, B, C, bscale);
    DGE_LAUNCH_CHECK("demod_bwd_prep");
    return 0;
}

extern "C" int dge_torgb_bwd_prep(const float* gimg, const void* x, const float* wrgb, const float* style, const float* noise,
                                  const float* noise_strength, int noise_batch, void* gz, float* gs, float* P, int B, int HW, int C,
                                  float wscale, float gain, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0 && C / ep <= 256 && 256 % (C / ep) == 0, "torgb_bwd_prep: unsupported channel count %d", C);
    DGE_CHECK(gain > 0.f, "torgb_bwd_prep: gain must be positive");
    DGE_CHECK(!dge_get_deterministic(), "torgb_bwd_prep: two flushes per workgroup - not offered in deterministic mode; run dge_torgb_bwd + dge_modconv_bwd_prep");
    dim3 grid(dge_stream_grid(HW, 256 / (C / ep), B), B);
    const int nbs = noise_batch > 1 ? HW : 0;
    if (dtype == DGE_BF16)
        hipLaunchKernelGGL(torgb_bwd_prep_kernel<bf16_t>, grid, dim3(256), 0, s, gimg, (const bf16_t*)x, wrgb, style, noise, noise_strength,
                           (bf16_t*)gz, gs, P, HW, C, wscale, gain, nbs);
    else
        hipLaunchKernelGGL(torgb_bwd_prep_kernel<float>, grid, dim3(256), 0, s, gimg, (const float*)x, wrgb, style, noise, noise_strength,
                           (float*)gz, gs, P, HW, C, wscale, gain, nbs);
    DGE_LAUNCH_CHECK("torgb_bwd_prep");
    return 0;
}

extern "C" int dge_linear_t(const float* x, int ldx, int incx, const float* w, const float* mul, float* y, int ldy, int incy,
                            int B, int O, int K, float scale, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(linear_t_kernel, dim3((K + 63) / 64, B), dim3(1024), 0, s, x, ldx, incx, w, mul, y, ldy, incy, B, O, K, scale, accumulate);
    DGE_LAUNCH_CHECK("linear_t");
    return 0;
}

extern "C" int dge_torgb_bwd(const float* gimg, const void* x, const float* wrgb, const float* style, void* gx, float* gs,
                             int B, int HW, int C, float wscale, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0 && C / ep <= 256 && 256 % (C / ep) == 0, "torgb_bwd: unsupported channel count %d", C);
    dim3 grid(dge_stream_grid(HW, 256 / (C / ep), B), B);
    if (dtype == DGE_BF16)
        hipLaunchKernelGGL(torgb_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, gimg, (const bf16_t*)x, wrgb, style, (bf16_t*)gx, gs, HW, C, wscale);
    else
        hipLaunchKernelGGL(torgb_bwd_kernel<float>, grid, dim3(256), 0, s, gimg, (const float*)x, wrgb, style, (float*)gx, gs, HW, C, wscale);
    DGE_LAUNCH_CHECK("torgb_bwd");
    return 0;
}

extern "C" int dge_up2_bwd(const float* g, float* gprev, int BC, int h, int w, hipStream_t s) {
    const long n = (long)BC * h * w;
    if ((w & 3) == 0 && w >= 8) {
        const long nq = n / 4;
        hipLaunchKernelGGL(up2_bwd_v4_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, g, gprev, BC, h, w);
    } else
        hipLaunchKernelGGL(up2_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, gprev, BC, h, w);
    DGE_LAUNCH_CHECK("up2_bwd");
    return 0;
}
