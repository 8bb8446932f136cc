// StyleGAN2 support kernels around the implicit-GEMM conv: weight packing (incl. the folded
// transposed-conv + FIR up layer), sum-of-squares for demodulation, small dense layers
// (mapping MLP / style affine), toRGB + skip-branch upsample, layout conversion.
// Reference math: model/stylegan2_generator.py (lines cited per kernel).
#include <stdlib.h>
#include "common.h"
#include <vector>
#include <algorithm>
#include <string.h>
#include "conv_params.h"

// ------------------------------------------------------------------ weight packing
// in : OIHW f32 [Cout][Cin][KS][KS]       out: [tap][Ntot][Cin] T (rows >= valid N are zero)
// mode 0: forward conv          out[tap][o][i]      = scale * W[o][i][tap]
// mode 1: folded up layer       out[tap][ph*Cout+o][i] = scale * sum_{ty,tx} K[ty][tx] W[o][i][3-py-ty+2dy][3-px-tx+2dx]
//         (conv_transpose2d stride 2 with the flipped kernel followed by the 4x4 FIR,
//          stylegan2_generator.py:879-896 + :802-807, folded per output phase)
// mode 2: data-gradient conv    out[tap][i][o]      = scale * W[o][i][KS*KS-1-tap]   (N = Cin, K = Cout)
// mode 3: data-gradient of 1    out[tap][i][ph*Cout+o] = mode-1 weight of (ph, 8-tap, o, i)  (N = Cin, K = 4*Cout)
// mode 4: StyleGAN1 fused up     w is [Cin][Cout][3][3] (ConvTranspose2d(3, stride 2, pad 1) + transform_kernel:
//         W4[ky][kx] = W3[ky][kx] + W3[ky-1][kx] + W3[ky][kx-1] + W3[ky-1][kx-1], lreq.py:129-131);
//         out[2m+py] takes x[m+dy] with ky = py + 1 - 2 dy  ->  out[tap][ph*Cout+o][c]
__device__ __forceinline__ float sg1_up_weight(const float* __restrict__ w, int c, int o, int Cout, int ph, int tap) {
    const int py = ph >> 1, px = ph & 1, dy = tap / 3 - 1, dx = tap % 3 - 1;
    const int ky = py + 1 - 2 * dy, kx = px + 1 - 2 * dx;
    if (ky < 0 || ky > 3 || kx < 0 || kx > 3) return 0.f;
    const float* w3 = w + ((size_t)c * Cout + o) * 9;
    float v = 0.f;
    for (int a = 0; a < 2; a++)
        for (int bq = 0; bq < 2; bq++) {
            const int y = ky - a, x = kx - bq;
            if (y >= 0 && y < 3 && x >= 0 && x < 3) v += w3[y * 3 + x];
        }
    return v;
}

// folded (transposed conv stride 2, flipped 3x3) * (4x4 FIR) weight of output phase `ph`, input tap (dy,dx)
__device__ __forceinline__ float upfold_weight(const float* __restrict__ w, int o, int i, int Cin, int ph, int tap) {
    const int py = ph >> 1, px = ph & 1;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const float k1[4] = {0.25f, 0.75f, 0.75f, 0.25f};      // outer(k1,k1) = FIR/64*4
    float v = 0.f;
    for (int ty = 0; ty < 4; ty++) {
        const int wy = 3 - py - ty + 2 * dy;
        if (wy < 0 || wy > 2) continue;
        for (int tx = 0; tx < 4; tx++) {
            const int wx = 3 - px - tx + 2 * dx;
            if (wx < 0 || wx > 2) continue;
            v += k1[ty] * k1[tx] * w[((size_t)o * Cin + i) * 9 + wy * 3 + wx];
        }
    }
    return v;
}

// DGE_PACK_UPT2D_DGRAD (mode 6): adjoint of the up layer in PHASE form.  The data gradient runs on Z = FIR^T(g) stored
// t-grid-to-depth (dge_fir_t2d: Z[m][(py,px), o] = g_t[2m + p][o]): g_x[m][i] = sum_{a in {0,1}^2} sum_p sum_o
// Z[m + a][p, o] * w[o][i][2 - (2a + p)] for 2a + p <= 2 per axis (transposed conv t[2m + k] += x[m] w[2 - k], :879-895).
// In the 3x3 tap frame of dge_conv2d tap (dy,dx) reads input pixel m + (dy-1, dx-1): a = (dy-1, dx-1); taps with dy = 0 or
// dx = 0 are zero (and skipped by the kernel, dge_conv_desc.in_t2d).
__device__ __forceinline__ float upt2d_weight(const float* __restrict__ w, int o, int i, int Cin, int ph, int tap) {
    const int dy = tap / 3, dx = tap % 3;
    if (dy == 0 || dx == 0) return 0.f;
    const int ky = 2 * (dy - 1) + (ph >> 1), kx = 2 * (dx - 1) + (ph & 1);
    if (ky > 2 || kx > 2) return 0.f;
    return w[((size_t)o * Cin + i) * 9 + (2 - ky) * 3 + (2 - kx)];
}

// DGE_PACK_FRAG (mode bit 0x100): the same [tap][n][k] values stored in MFMA-fragment order for csrc/conv_small.hip, which loads
// its weight operand straight from global memory: the 32 rows x 16 k block (tap, n/32, k/16) is one contiguous 1 KiB run in
// lane order - lane = (n % 32) + 32 * ((k % 16) / 8) holds the 8 consecutive k of its v_mfma_f32_32x32x16_bf16 B operand - so a
// wave's operand load is `base + lane * 16`, fully coalesced.
__device__ __forceinline__ size_t pack_out_index(int frag, int tap, int n, int k, int Ntot, int Kdim) {
    if (!frag) return ((size_t)tap * Ntot + n) * Kdim + k;
    const size_t blk = ((size_t)tap * (Ntot >> 5) + (n >> 5)) * (Kdim >> 4) + (k >> 4);
    return blk * 512 + (size_t)(((n & 31) + 32 * ((k & 15) >> 3)) * 8 + (k & 7));
}

template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int KS,
                                   int Ntot, int mode, float scale) {
    const int frag = mode & 0x100;
    mode &= 0xff;
    const int ntap = KS * KS;
    const int Kdim = (mode == 2) ? Cout : ((mode == 3 || mode == 5 || mode == 6) ? 4 * Cout : Cin);
    const int total = ntap * Ntot * Kdim;            // < 2^31 (checked by the launcher): 32-bit index math
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int k = idx % Kdim;
        const int n = (idx / Kdim) % Ntot;
        const int tap = idx / (Kdim * Ntot);
        float v = 0.f;
        if (mode == 0) {
            if (n < Cout) v = w[((size_t)n * Cin + k) * ntap + tap];
        } else if (mode == 2) {
            if (n < Cin) v = w[((size_t)k * Cin + n) * ntap + (ntap - 1 - tap)];
        } else if (mode == 1) {
            if (n < 4 * Cout) v = upfold_weight(w, n % Cout, k, Cin, n / Cout, tap);
        } else if (mode == 4) {
            if (n < 4 * Cout) v = sg1_up_weight(w, k, n % Cout, Cout, n / Cout, tap);
        } else if (mode == 5) {
            if (n < Cin) v = sg1_up_weight(w, n, k % Cout, Cout, k / Cout, 8 - tap);
        } else if (mode == 6) {
            if (n < Cin) v = upt2d_weight(w, k % Cout, n, Cin, k / Cout, tap);
        } else {
            if (n < Cin) v = upfold_weight(w, k % Cout, n, Cin, k / Cout, 8 - tap);
        }
        Elem<T>::st(out + pack_out_index(frag, tap, n, k, Ntot, Kdim), v * scale);
    }
}

// Every packed weight copy of a module in ONE launch (the encoder re-packs ~35 weights after each optimizer step: 71 launches
// of 3-9 us per training step).  One thread owns a (packed row n, k) pair and loops over the taps: for the forward layout the
// k*k source values are one contiguous 36-byte run, and the writes of a wave are contiguous per tap.
struct DgePackDesc {
    const float* w; void* out;
    int cout, cin, ks, ntot, mode, dtype, kdim;
    float scale;
    long long pair_start;                 // first (n, k) pair of this tensor in the launch's pair space
    long long tile_start;                 // first 32 x 32 (n, k) tile of this tensor in the launch's tile space
    // paired entry (tiled kernel only): the data-gradient copy (mode 2) of the SAME weight rides on the forward copy's tiles - one
    // staged 32 x 32 x taps source block feeds both layouts (the encoder re-packs both after every optimizer step: the source is
    // read once instead of twice).  out2 == nullptr: not paired.  An entry that was absorbed by its partner has tiles_n = 0.
    void* out2;
    int ntot2, kdim2, mode2, tiles_i;     // tiles_i: tiles along the source's input-channel axis (paired entries)
};
__device__ __forceinline__ float pack_gather(const float* __restrict__ w, int mode, int Cout, int Cin, int ntap, int n, int k, int tap) {
    if (mode == 0) return n < Cout ? w[((size_t)n * Cin + k) * ntap + tap] : 0.f;
    if (mode == 2) return n < Cin ? w[((size_t)k * Cin + n) * ntap + (ntap - 1 - tap)] : 0.f;
    if (mode == 1) return n < 4 * Cout ? upfold_weight(w, n % Cout, k, Cin, n / Cout, tap) : 0.f;
    if (mode == 4) return n < 4 * Cout ? sg1_up_weight(w, k, n % Cout, Cout, n / Cout, tap) : 0.f;
    if (mode == 5) return n < Cin ? sg1_up_weight(w, n, k % Cout, Cout, k / Cout, 8 - tap) : 0.f;
    if (mode == 6) return n < Cin ? upt2d_weight(w, k % Cout, n, Cin, k / Cout, tap) : 0.f;
    return n < Cin ? upfold_weight(w, k % Cout, n, Cin, k / Cout, 8 - tap) : 0.f;
}
__global__ void pack_multi_kernel(const DgePackDesc* __restrict__ descs, int nd, long long total_pairs) {
    for (long long pidx = (long long)blockIdx.x * blockDim.x + threadIdx.x; pidx < total_pairs; pidx += (long long)gridDim.x * blockDim.x) {
        int lo = 0, hi = nd - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (descs[mid].pair_start <= pidx) lo = mid; else hi = mid - 1; }
        const DgePackDesc e = descs[lo];
        const long long local = pidx - e.pair_start;
        const int k = (int)(local % e.kdim), n = (int)(local / e.kdim);
        const int ntap = e.ks * e.ks;
        for (int tap = 0; tap < ntap; tap++) {
            const float v = pack_gather(e.w, e.mode & 0xff, e.cout, e.cin, ntap, n, k, tap) * e.scale;
            const size_t o = pack_out_index(e.mode & 0x100, tap, n, k, e.ntot, e.kdim);
            if (e.dtype == DGE_BF16) ((bf16_t*)e.out)[o] = f2bf(v);
            else ((float*)e.out)[o] = v;
        }
    }
}

// The same, tiled through LDS.  pack_multi_kernel reads the transposed (data-gradient) layouts with a 36-byte run per thread
// from rows Cin*36 bytes apart - every run drags a whole line in - and spent ~200 us on the encoder's 24 M parameters after
// each optimizer step.  Here a workgroup owns a 32 x 32 (n, k) tile of one packed tensor for all taps: the source block is
// 32 rows (output channels of w) of 32*taps CONTIGUOUS floats in both layouts, read coalesced into LDS, and written out k-fastest
// (64-byte runs of bf16, or the 1 KiB fragment runs).  Other modes gather from global memory as before.
// Round 5: (i) the tile -> tensor search runs on a copy of the tile_start column in LDS (it was ~7 dependent global loads per tile),
// (ii) the source block is fetched as 16-byte loads, all of a thread's loads in flight at once (it was 36 dependent 4-byte loads per
// thread), (iii) a forward copy and the data-gradient copy of the same weight share one staged block (DgePackDesc::out2).
constexpr int PK_LDW = 32 * 9 + 1;                        // row pitch in floats (odd: the transposed reads spread over the banks)
template <bool TR>
__device__ __forceinline__ void pack_emit_bf16(const float* __restrict__ tile, int ntap, int n0, int k0, bf16_t* __restrict__ out, int ntot, int kdim,
                                               int frag, float scale) {
    // 8 consecutive k per thread: one 16-byte store (row-major: a 64-byte run per n; fragment order: 8 k of a lane)
    for (int idx = threadIdx.x; idx < 128 * ntap; idx += 256) {
        const int kc = idx & 3, nn = (idx >> 2) & 31, tap = idx >> 7;
        const int n = n0 + nn, k = k0 + kc * 8;
        if (k >= kdim || n >= ntot) continue;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++)
            v[j] = scale * (!TR ? tile[nn * PK_LDW + (kc * 8 + j) * ntap + tap] : tile[(kc * 8 + j) * PK_LDW + nn * ntap + (ntap - 1 - tap)]);
        *(uint4*)(out + pack_out_index(frag, tap, n, k, ntot, kdim)) =
            make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
}
__global__ __launch_bounds__(256) void pack_multi_tiled_kernel(const DgePackDesc* __restrict__ descs, int nd, long long total_tiles) {
    __shared__ float tile[32 * PK_LDW];
    constexpr int NDL = 512;
    __shared__ long long tstart[NDL];
    const bool lds_search = nd <= NDL;
    if (lds_search) {
        for (int i = threadIdx.x; i < nd; i += 256) tstart[i] = descs[i].tile_start;
        __syncthreads();
    }
    for (long long tix = blockIdx.x; tix < total_tiles; tix += gridDim.x) {
        int lo = 0, hi = nd - 1;
        // (entries without tiles share their successor's tile_start: "last entry with tile_start <= tix" skips them)
        if (lds_search) { while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tstart[mid] <= tix) lo = mid; else hi = mid - 1; } }
        else { while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (descs[mid].tile_start <= tix) lo = mid; else hi = mid - 1; } }
        const DgePackDesc e = descs[lo];
        const int local = (int)(tix - e.tile_start);
        const int ntap = e.ks * e.ks, mode = e.mode & 0xff, frag = e.mode & 0x100;
        const bool paired = e.out2 != nullptr;
        const int kt = paired ? e.tiles_i : (e.kdim + 31) / 32;
        const int n0 = (local / kt) * 32, k0 = (local % kt) * 32;
        const bool staged = (mode == 0 || mode == 2) && ntap <= 9;
        __syncthreads();                                   // the previous tile's readers are done
        if (staged) {
            // source rows = output channels o of w [Cout][Cin][taps]; columns = (i, tap) for 32 consecutive i
            const int o0 = mode == 0 ? n0 : k0, i0 = mode == 0 ? k0 : n0;
            const int run = 32 * ntap;
            const int iv = min(32, e.cin - i0);            // valid input channels of the block (<= 0: none)
            if ((e.cin & 3) == 0 && ntap == 9) {
                // 16-byte loads: row r holds iv * 9 valid floats (a multiple of 4, 16-byte aligned: cin % 4 == 0, i0 % 32 == 0); 72 units per full row
                const int upr = iv > 0 ? iv * 9 / 4 : 0;
                float4 v[9];
                int rr[9], cc[9];
#pragma unroll
                for (int q = 0; q < 9; q++) {
                    const int u = threadIdx.x + q * 256;       // 32 rows x 72 units
                    rr[q] = u / 72; cc[q] = u - rr[q] * 72;
                    const int o = o0 + rr[q];
                    v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (o < e.cout && cc[q] < upr) v[q] = *(const float4*)(e.w + ((size_t)o * e.cin + i0) * 9 + cc[q] * 4);
                }
#pragma unroll
                for (int q = 0; q < 9; q++) {
                    float* t = tile + rr[q] * PK_LDW + cc[q] * 4;
                    t[0] = v[q].x; t[1] = v[q].y; t[2] = v[q].z; t[3] = v[q].w;
                }
            } else {
                for (int idx = threadIdx.x; idx < 32 * run; idx += 256) {
                    const int r = idx / run, c = idx - r * run;
                    const int o = o0 + r, i = i0 + c / ntap;
                    tile[r * PK_LDW + c] = (o < e.cout && i < e.cin) ? e.w[((size_t)o * e.cin + i0) * ntap + c] : 0.f;
                }
            }
            __syncthreads();
        }
        if (staged && e.dtype == DGE_BF16 && (e.kdim & 7) == 0) {
            if (mode == 0) pack_emit_bf16<false>(tile, ntap, n0, k0, (bf16_t*)e.out, e.ntot, e.kdim, frag, e.scale);
            else pack_emit_bf16<true>(tile, ntap, n0, k0, (bf16_t*)e.out, e.ntot, e.kdim, frag, e.scale);
            // (paired: this entry is the forward copy - n = o, k = i; the partner is the data-gradient copy - n = i, k = o)
            if (paired) pack_emit_bf16<true>(tile, ntap, k0, n0, (bf16_t*)e.out2, e.ntot2, e.kdim2, e.mode2 & 0x100, e.scale);
            continue;
        }
        for (int idx = threadIdx.x; idx < 1024 * ntap; idx += 256) {
            const int kk = idx & 31, nn = (idx >> 5) & 31, tap = idx >> 10;
            const int n = n0 + nn, k = k0 + kk;
            if (k >= e.kdim) continue;
            float v;
            if (staged) v = mode == 0 ? tile[nn * PK_LDW + kk * ntap + tap] : tile[kk * PK_LDW + nn * ntap + (ntap - 1 - tap)];
            else v = pack_gather(e.w, mode, e.cout, e.cin, ntap, n, k, tap);
            v *= e.scale;
            const size_t o = pack_out_index(frag, tap, n, k, e.ntot, e.kdim);
            if (e.dtype == DGE_BF16) ((bf16_t*)e.out)[o] = f2bf(v);
            else ((float*)e.out)[o] = v;
        }
    }
}

// wsq[o][i] = scale^2 * sum_taps W[o][i][t]^2     (demodulation norm, :867-870)
__global__ void wsq_kernel(const float* __restrict__ w, float* __restrict__ wsq, int n_oi, int ntap, float scale2) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_oi) return;
    float s = 0.f;
    for (int t = 0; t < ntap; t++) { const float v = w[(size_t)idx * ntap + t]; s += v * v; }
    wsq[idx] = s * scale2;
}

// ------------------------------------------------------------------ small dense layers
// y[b][o] = act( (sum_i f(x[b][i]) * W[o][i]) * wscale + bias[o]*bscale + add ) * gain
// f = identity or square (square: demodulation d = rsqrt(s^2 . wsq + eps) with act = rsqrt).
// x rows may be strided (ldx) so a [B, L, 512] wp tensor can be indexed per layer.
enum { LIN_ACT_NONE = 0, LIN_ACT_LRELU = 1, LIN_ACT_RELU = 2, LIN_ACT_RSQRT = 3 };
__global__ void linear_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ W,
                              const float* __restrict__ bias, float* __restrict__ y, int ldy, int B, int I, int O,
                              float wscale, float bscale, float add, int act, float gain, int square) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B * O) return;
    const int b = wave / O, o = wave % O;
    const float* xr = x + (size_t)b * ldx;
    const float* wr = W + (size_t)o * I;
    float s = 0.f;
    for (int i = lane; i < I; i += 64) { float xv = xr[i]; if (square) xv *= xv; s += xv * wr[i]; }
    s = wave_sum(s);
    if (lane == 0) {
        float v = s * wscale + (bias ? bias[o] * bscale : 0.f) + add;
        if (act == LIN_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
        else if (act == LIN_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (act == LIN_ACT_RSQRT) v = rsqrtf(v);
        y[(size_t)b * ldy + o] = v * gain;
    }
}

// A chain of dense layers in ONE launch (the mapping network: pixel norm + 8 DenseBlocks of 512, :199-278 / :925-996; 16 launches of
// ~5.6 us per generator forward before).  One workgroup per sample, 16 waves; the activations ping-pong through LDS; wave v computes
// outputs v, v + 16, ... with exactly linear_kernel's arithmetic (lane-strided partial sums, wave_sum, scale / bias / act / gain), so
// the result is bit-identical to the per-layer launches.
struct DenseChain { const float* w[8]; const float* bias[8]; int I[8], O[8], act[8]; float wscale[8], bscale[8], add[8], gain[8]; int n; };
__global__ __launch_bounds__(1024) void dense_chain_kernel(const float* __restrict__ x, int ldx, DenseChain c, float* __restrict__ y, int ldy,
                                                           int pixelnorm, float eps) {
    __shared__ float buf[2][1024];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int I0 = c.I[0];
    if (pixelnorm) {                               // dge_pixelnorm's arithmetic (one wave: lane-strided sum of squares, wave_sum)
        if (wave == 0) {
            float s = 0.f;
            for (int i = lane; i < I0; i += 64) { const float v = x[(size_t)b * ldx + i]; s += v * v; }
            s = wave_sum(s);
            const float r = rsqrtf(s / I0 + eps);
            for (int i = lane; i < I0; i += 64) buf[0][i] = x[(size_t)b * ldx + i] * r;
        }
    } else {
        for (int i = tid; i < I0; i += 1024) buf[0][i] = x[(size_t)b * ldx + i];
    }
    __syncthreads();
    int cur = 0;
    for (int l = 0; l < c.n; l++) {
        const int I = c.I[l], O = c.O[l];
        const float* __restrict__ W = c.w[l];
        const float* __restrict__ bias = c.bias[l];
        const bool last = l == c.n - 1;
        for (int o = wave; o < O; o += 16) {
            const float* wr = W + (size_t)o * I;
            float s = 0.f;
            for (int i = lane; i < I; i += 64) s += buf[cur][i] * wr[i];
            s = wave_sum(s);
            if (lane == 0) {
                float v = s * c.wscale[l] + (bias ? bias[o] * c.bscale[l] : 0.f) + c.add[l];
                if (c.act[l] == LIN_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
                else if (c.act[l] == LIN_ACT_RELU) v = v > 0.f ? v : 0.f;
                v *= c.gain[l];
                if (last) y[(size_t)b * ldy + o] = v; else buf[cur ^ 1][o] = v;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

// All style vectors of a synthesis pass in ONE launch (17 modulated convs + 9 toRGB at 1024^2 = 26 dense layers,
// :858-864 / :990-996): row r of the concatenated weight matrix belongs to one layer and reads that layer's latent row
// x[b, row_xoff[r] .. +K); results are written per layer as contiguous [B, C_layer] blocks (the conv prologue indexes
// in_scale[b*Cin + c]):  y[row_ybase[r] + b*row_ybstride[r]] = wscale * <x_b, W_r> + bias[r]*bscale + add.
__global__ void linear_rows_kernel(const float* __restrict__ x, int ldx_b, const int* __restrict__ row_xoff,
                                   const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ y,
                                   const int* __restrict__ row_ybase, const int* __restrict__ row_ybstride, int B, int R, int K,
                                   float wscale, float bscale, float add) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B * R) return;
    const int b = wave / R, r = wave % R;
    const float* xr = x + (size_t)b * ldx_b + row_xoff[r];
    const float* wr = W + (size_t)r * K;
    float s = 0.f;
    for (int i = lane; i < K; i += 64) s += xr[i] * wr[i];
    s = wave_sum(s);
    if (lane == 0) y[(size_t)row_ybase[r] + (size_t)b * row_ybstride[r]] = s * wscale + bias[r] * bscale + add;
}

// All demodulation factors of a synthesis pass in one launch (:867-870): row r = output channel o of some layer,
// d = rsqrt(sum_c s[b,c]^2 * wsq[o,c] + eps) with that layer's style block s (contiguous [B, cin]) and wsq row.
__global__ void demod_rows_kernel(const float* __restrict__ s_all, const float* __restrict__ wsq_cat, const int* __restrict__ row_woff,
                                  const int* __restrict__ row_sbase, const int* __restrict__ row_cin, float* __restrict__ d_all,
                                  const int* __restrict__ row_dbase, const int* __restrict__ row_dbstride, int B, int R, float eps) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B * R) return;
    const int b = wave / R, r = wave % R;
    const int cin = row_cin[r];
    const float* sr = s_all + (size_t)row_sbase[r] + (size_t)b * cin;
    const float* wr = wsq_cat + (size_t)row_woff[r];
    float a = 0.f;
    for (int i = lane; i < cin; i += 64) { const float v = sr[i]; a += v * v * wr[i]; }
    a = wave_sum(a);
    if (lane == 0) d_all[(size_t)row_dbase[r] + (size_t)b * row_dbstride[r]] = rsqrtf(a + eps);
}

// pixel norm over rows: y = x / sqrt(mean(x^2) + eps)   (:550-553)
__global__ void pixelnorm_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int D, float eps) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B) return;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) { const float v = x[(size_t)wave * D + i]; s += v * v; }
    s = wave_sum(s);
    const float r = rsqrtf(s / D + eps);
    for (int i = lane; i < D; i += 64) y[(size_t)wave * D + i] = x[(size_t)wave * D + i] * r;
}

// truncation: wp[b][l][:] = w_avg + (w[b][:] - w_avg) * (l < layers ? psi : 1)      (:311-333)
// w is [B, D] (repeat) when w_is_wp == 0, else [B, L, D].
__global__ void truncation_kernel(const float* __restrict__ w, const float* __restrict__ w_avg, float* __restrict__ wp,
                                  int B, int L, int D, float psi, int layers, int w_is_wp, int apply) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)B * L * D) return;
    const int d = idx % D, l = (idx / D) % L, b = idx / ((long)D * L);
    const float v = w_is_wp ? w[idx] : w[(size_t)b * D + d];
    if (!apply) { wp[idx] = v; return; }
    const float coef = l < layers ? psi : 1.f;
    wp[idx] = w_avg[d] + (v - w_avg[d]) * coef;
}

// ------------------------------------------------------------------ toRGB + skip upsample
// img[b][c][y][x] (NCHW f32) = sum_i x[b,y,x,i] * (Wrgb[c][i]*wscale*s[b][i]) + bias[c]
//                              + up2(prev)[b][c][y][x]        (prev: [B,3,H/2,W/2] f32 or null)
// ModulateConvBlock k=1, demodulate=False, linear (:465-474) and UpsamplingLayer scale 2
// (:603-615): zero-insert, pad (2,1), 4x4 FIR (sum 4) == per-axis taps {.25,.75} / {.75,.25}.
template <typename T, int NC>
__global__ void torgb_kernel(const T* __restrict__ x, const float* __restrict__ wrgb, const float* __restrict__ s,
                             const float* __restrict__ bias, const float* __restrict__ prev, float* __restrict__ img,
                             int B, int H, int W, int Cin, float wscale) {
    extern __shared__ float wl[];                 // [NC][Cin] modulated weights of this sample
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < NC * Cin; i += blockDim.x) wl[i] = wrgb[i] * wscale * s[(size_t)b * Cin + i % Cin];
    __syncthreads();
    constexpr int EP16 = Elem<T>::PER16;
    const int HW = H * W;
    for (int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < HW; pix += gridDim.x * blockDim.x) {
        const T* xp = x + ((size_t)b * HW + pix) * Cin;
        float acc[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) acc[c] = 0.f;
        for (int i = 0; i < Cin; i += EP16) {
            const uint4 v = *(const uint4*)(xp + i);
            float f[EP16];
            unpack16(v, f, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < EP16; e++)
#pragma unroll
                for (int c = 0; c < NC; c++) acc[c] += f[e] * wl[c * Cin + i + e];
        }
        const int y = pix / W, xx = pix % W;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            float v = acc[c] + bias[c];
            if (prev) {
                const int h2 = H >> 1, w2 = W >> 1;
                const float* pp = prev + ((size_t)b * NC + c) * h2 * w2;
                const int my = y >> 1, mx = xx >> 1;
                // even output: .25*x[m-1] + .75*x[m];  odd output: .75*x[m] + .25*x[m+1]
                const int ya = (y & 1) ? my : my - 1, yb = (y & 1) ? my + 1 : my;
                const int xa = (xx & 1) ? mx : mx - 1, xb = (xx & 1) ? mx + 1 : mx;
                const float wya = (y & 1) ? 0.75f : 0.25f, wyb = 1.f - wya;
                const float wxa = (xx & 1) ? 0.75f : 0.25f, wxb = 1.f - wxa;
                auto at = [&](int yy, int xq) { return (yy >= 0 && yy < h2 && xq >= 0 && xq < w2) ? pp[yy * w2 + xq] : 0.f; };
                v += wya * (wxa * at(ya, xa) + wxb * at(ya, xb)) + wyb * (wxa * at(yb, xa) + wxb * at(yb, xb));
            }
            img[((size_t)b * NC + c) * HW + pix] = v;
        }
    }
}

// The same layer for the low resolutions (4^2 .. 64^2, Cin = 512): one WAVE per pixel - the lanes split the channels (one
// 16-byte load each per 64*EP16 channels), three wave reductions - instead of one thread walking all 512 channels of its pixel
// (a 4^2 image gave 16 busy threads per sample and ~30 us of pure latency per launch; six such launches per synthesis pass).
template <typename T, int NC>
__global__ __launch_bounds__(256) void torgb_wave_kernel(const T* __restrict__ x, const float* __restrict__ wrgb, const float* __restrict__ s,
                                                         const float* __restrict__ bias, const float* __restrict__ prev, float* __restrict__ img,
                                                         int B, int H, int W, int Cin, float wscale) {
    constexpr int EP16 = Elem<T>::PER16;
    const int lane = threadIdx.x & 63;
    const int HW = H * W;
    const long gp = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // global pixel = b * HW + pix (wave-uniform)
    if (gp >= (long)B * HW) return;
    const int b = (int)(gp / HW), pix = (int)(gp - (long)b * HW);
    const T* xp = x + (size_t)gp * Cin;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = 0.f;
    for (int i = lane * EP16; i < Cin; i += 64 * EP16) {
        float f[EP16], sv[EP16];
        unpack16(*(const uint4*)(xp + i), f, (T*)nullptr);
#pragma unroll
        for (int e4 = 0; e4 < EP16 / 4; e4++) *(float4*)&sv[e4 * 4] = *(const float4*)(s + (size_t)b * Cin + i + e4 * 4);
#pragma unroll
        for (int e = 0; e < EP16; e++) f[e] *= sv[e];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            float wv[EP16];
#pragma unroll
            for (int e4 = 0; e4 < EP16 / 4; e4++) *(float4*)&wv[e4 * 4] = *(const float4*)(wrgb + (size_t)c * Cin + i + e4 * 4);
#pragma unroll
            for (int e = 0; e < EP16; e++) acc[c] = fmaf(f[e], wv[e], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = wave_sum(acc[c]) * wscale;
    if (lane < NC) {
        const int c = lane;
        float v = bias[c];
#pragma unroll
        for (int k = 0; k < NC; k++) v += (k == c) ? acc[k] : 0.f;
        if (prev) {
            const int y = pix / W, xx = pix % W;
            const int h2 = H >> 1, w2 = W >> 1;
            const float* pp = prev + ((size_t)b * NC + c) * h2 * w2;
            const int my = y >> 1, mx = xx >> 1;
            const int ya = (y & 1) ? my : my - 1, yb = (y & 1) ? my + 1 : my;
            const int xa = (xx & 1) ? mx : mx - 1, xb = (xx & 1) ? mx + 1 : mx;
            const float wya = (y & 1) ? 0.75f : 0.25f, wyb = 1.f - wya;
            const float wxa = (xx & 1) ? 0.75f : 0.25f, wxb = 1.f - wxa;
            auto at = [&](int yy, int xq) { return (yy >= 0 && yy < h2 && xq >= 0 && xq < w2) ? pp[yy * w2 + xq] : 0.f; };
            v += wya * (wxa * at(ya, xa) + wxb * at(ya, xb)) + wyb * (wxa * at(yb, xa) + wxb * at(yb, xb));
        }
        img[((size_t)b * NC + c) * HW + pix] = v;
    }
}

// ------------------------------------------------------------------ layout conversion
// NCHW f32 (batch-broadcast when src_B == 1) -> NHWC T
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int HW, int src_B) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)B * C * HW) return;
    const int c = idx % C; const long r = idx / C; const int pix = r % HW; const int b = r / HW;
    const int sb = src_B == 1 ? 0 : b;
    Elem<T>::st(dst + idx, src[((size_t)sb * C + c) * HW + pix]);
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int C, int HW) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)B * C * HW) return;
    const int pix = idx % HW; const long r = idx / HW; const int c = r % C; const int b = r / C;
    dst[idx] = Elem<T>::ld(src + ((size_t)b * HW + pix) * C + c);
}

// out[b,l,d] = avg[l*avg_stride + d] + (w[b,d] - avg[...]) * coefs[l]
__global__ void lerp_layers_kernel(const float* __restrict__ w, const float* __restrict__ avg, int avg_stride,
                                   const float* __restrict__ coefs, float* __restrict__ out, int B, int L, int D) {
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= (long)B * L * D) return;
    const int d = idx % D, l = (idx / D) % L, b = idx / ((long)D * L);
    const float a = avg[(size_t)l * avg_stride + d];
    out[idx] = a + (w[(size_t)b * D + d] - a) * coefs[l];
}

// =================================================================== C ABI
#include "../../include/dge_hip.h"

extern "C" int dge_lerp_layers(const float* w, const float* avg, int avg_stride, const float* coefs, float* out, int B, int L,
                               int D, hipStream_t s) {
    const long n = (long)B * L * D;
    hipLaunchKernelGGL(lerp_layers_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, avg, avg_stride, coefs, out, B, L, D);
    DGE_LAUNCH_CHECK("lerp_layers");
    return 0;
}

extern "C" int dge_linear_rows(const float* x, int ldx_b, const int* row_xoff, const float* w, const float* bias, float* y,
                               const int* row_ybase, const int* row_ybstride, int B, int R, int K, float wscale, float bscale,
                               float add, hipStream_t s) {
    const long waves = (long)B * R;
    hipLaunchKernelGGL(linear_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, ldx_b, row_xoff, w, bias, y, row_ybase,
                       row_ybstride, B, R, K, wscale, bscale, add);
    DGE_LAUNCH_CHECK("linear_rows");
    return 0;
}

extern "C" int dge_demod_rows(const float* s_all, const float* wsq_cat, const int* row_woff, const int* row_sbase, const int* row_cin,
                              float* d_all, const int* row_dbase, const int* row_dbstride, int B, int R, float eps, hipStream_t s) {
    const long waves = (long)B * R;
    hipLaunchKernelGGL(demod_rows_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, s_all, wsq_cat, row_woff, row_sbase, row_cin,
                       d_all, row_dbase, row_dbstride, B, R, eps);
    DGE_LAUNCH_CHECK("demod_rows");
    return 0;
}

extern "C" int dge_pack_conv_weight(const float* w_oihw, void* out, int cout, int cin, int ksize, int mode, int dtype,
                                    float scale, hipStream_t s) {
    const int frag = mode & 0x100;
    const int mode_full = mode;
    mode &= 0xff;
    DGE_CHECK(mode >= 0 && mode <= 6 && (mode_full & ~0x1ff) == 0, "pack: bad mode %d", mode_full);
    DGE_CHECK((mode != 1 && mode < 3) || ksize == 3, "pack: up fold needs a 3x3 kernel");
    const int nvalid = (mode == 1 || mode == 4) ? 4 * cout : (mode >= 2 ? cin : cout);
    const int ntot = dge_packed_n(nvalid);
    const int kdim = mode == 2 ? cout : ((mode == 3 || mode == 5 || mode == 6) ? 4 * cout : cin);
    DGE_CHECK(!frag || (dtype == DGE_BF16 && ntot % 32 == 0 && kdim % 16 == 0), "pack: fragment order needs bf16, N %% 32 == 0, K %% 16 == 0");
    mode = mode_full;
    const long total = (long)ksize * ksize * ntot * kdim;
    DGE_CHECK(total < (1L << 31) - (4096L * 256), "pack: weight too large (%ld elements)", total);
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == DGE_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, w_oihw, (bf16_t*)out, cout, cin, ksize, ntot, mode, scale);
    else
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(grid), dim3(256), 0, s, w_oihw, (float*)out, cout, cin, ksize, ntot, mode, scale);
    DGE_LAUNCH_CHECK("pack_conv_weight");
    return 0;
}

// `table`: device array of `n` rows of 8 x int64: {w, out, cout | cin << 32, ksize | mode << 32, dtype, scale (float bits), -, -};
// the rows are validated on the host by the caller's wrapper (this entry cannot read device memory).  `descs`: device scratch of
// n * sizeof(DgePackDesc) bytes the call fills from `table_host` when `upload` is set.
extern "C" int dge_pack_conv_weights_multi(const long long* table_host, void* descs_dev, int n, int upload, hipStream_t s) {
    DGE_CHECK(n > 0 && n <= 4096, "pack_multi: bad entry count %d", n);
    static thread_local std::vector<DgePackDesc> host;
    host.resize(n);
    long long pairs = 0, tiles = 0;
    for (int i = 0; i < n; i++) {
        const long long* r = table_host + (size_t)i * 8;
        DgePackDesc& d = host[i];
        d.w = (const float*)r[0]; d.out = (void*)r[1];
        d.cout = (int)(r[2] & 0xffffffffLL); d.cin = (int)(r[2] >> 32);
        d.ks = (int)(r[3] & 0xffffffffLL); d.mode = (int)(r[3] >> 32);
        d.dtype = (int)r[4];
        const unsigned bits = (unsigned)r[5]; memcpy(&d.scale, &bits, 4);
        const int bm = d.mode & 0xff;
        DGE_CHECK(bm >= 0 && bm <= 6 && (d.mode & ~0x1ff) == 0, "pack_multi: bad mode %d", d.mode);
        DGE_CHECK((bm != 1 && bm < 3) || d.ks == 3, "pack_multi: up fold needs a 3x3 kernel");
        const int nvalid = (bm == 1 || bm == 4) ? 4 * d.cout : (bm >= 2 ? d.cin : d.cout);
        d.ntot = dge_packed_n(nvalid);
        d.kdim = bm == 2 ? d.cout : ((bm == 3 || bm == 5 || bm == 6) ? 4 * d.cout : d.cin);
        DGE_CHECK(!(d.mode & 0x100) || (d.dtype == DGE_BF16 && d.ntot % 32 == 0 && d.kdim % 16 == 0),
                  "pack_multi: fragment order needs bf16, N %% 32 == 0, K %% 16 == 0");
        d.pair_start = pairs;
        pairs += (long long)d.ntot * d.kdim;
        d.out2 = nullptr; d.ntot2 = d.kdim2 = d.mode2 = d.tiles_i = 0;
        DGE_CHECK(d.ntot % 32 == 0, "pack_multi: packed N is padded to the N tile (multiple of 32)");
    }
    // forward copy + data-gradient copy of one weight -> one entry of the tiled kernel (see DgePackDesc::out2)
    static int nopair = -1;
    if (nopair < 0) nopair = getenv("DGE_PACK_NOPAIR") ? 1 : 0;
    std::vector<char> absorbed(n, 0);
    if (!nopair)
        for (int i = 0; i < n; i++) {
            DgePackDesc& f = host[i];
            if ((f.mode & 0xff) != 0 || f.dtype != DGE_BF16 || f.ks != 3 || (f.kdim & 7) || absorbed[i]) continue;
            for (int j = 0; j < n; j++) {
                const DgePackDesc& g = host[j];
                if (j == i || absorbed[j] || g.out2 || (g.mode & 0xff) != 2 || g.w != f.w || g.dtype != DGE_BF16 || g.ks != 3 || (g.kdim & 7) ||
                    g.cout != f.cout || g.cin != f.cin || g.scale != f.scale)
                    continue;
                f.out2 = g.out; f.ntot2 = g.ntot; f.kdim2 = g.kdim; f.mode2 = g.mode;
                absorbed[j] = 1;
                break;
            }
        }
    for (int i = 0; i < n; i++) {
        DgePackDesc& d = host[i];
        d.tile_start = tiles;
        if (absorbed[i]) continue;
        if (d.out2) {
            const int to = std::max(d.ntot / 32, (d.kdim2 + 31) / 32);          // tiles along the source's output-channel axis
            d.tiles_i = std::max((d.kdim + 31) / 32, d.ntot2 / 32);
            tiles += (long long)to * d.tiles_i;
        } else {
            tiles += (long long)(d.ntot / 32) * ((d.kdim + 31) / 32);
        }
    }
    // upload = 0: descs_dev still holds the descriptors of an earlier call with the same table (the steady state of a
    // training loop: a pageable host -> device copy waits for the stream to drain, so it is paid once, not per step)
    if (upload) {
        // (a memcpy node of a stream capture would record the address of this thread's host table, which the next call rewrites:
        //  a replay would upload whatever table is current.  The table of a training loop is uploaded by its eager warm-up steps.)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess)
            DGE_CHECK(cs == hipStreamCaptureStatusNone, "pack_multi: the descriptor table changed inside a stream capture; run the step eagerly once first");
        hipError_t e = hipMemcpyAsync(descs_dev, host.data(), (size_t)n * sizeof(DgePackDesc), hipMemcpyHostToDevice, s);
        DGE_CHECK(e == hipSuccess, "pack_multi: descriptor upload failed: %s", hipGetErrorString(e));
    }
    static int untiled = -1;
    if (untiled < 0) untiled = getenv("DGE_PACK_UNTILED") ? 1 : 0;
    if (untiled) {
        long long blocks = (pairs + 255) / 256;
        const int grid = (int)(blocks > 8192 ? 8192 : blocks);
        hipLaunchKernelGGL(pack_multi_kernel, dim3(grid), dim3(256), 0, s, (const DgePackDesc*)descs_dev, n, pairs);
    } else {
        const int grid = (int)(tiles > 4096 ? 4096 : tiles);
        hipLaunchKernelGGL(pack_multi_tiled_kernel, dim3(grid), dim3(256), 0, s, (const DgePackDesc*)descs_dev, n, tiles);
    }
    DGE_LAUNCH_CHECK("pack_conv_weights_multi");
    return 0;
}
extern "C" int dge_pack_desc_bytes(void) { return (int)sizeof(DgePackDesc); }

extern "C" int dge_packed_n(int nvalid) {
    const int t = dge_conv_ntile(nvalid);
    return (nvalid + t - 1) / t * t;
}

extern "C" int dge_weight_sumsq(const float* w_oihw, float* wsq, int cout, int cin, int ksize, float scale, hipStream_t s) {
    const int n = cout * cin;
    hipLaunchKernelGGL(wsq_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w_oihw, wsq, n, ksize * ksize, scale * scale);
    DGE_LAUNCH_CHECK("weight_sumsq");
    return 0;
}

extern "C" int dge_linear(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int I, int O,
                          float wscale, float bscale, float add, int act, float gain, int square_input, hipStream_t s) {
    DGE_CHECK(B > 0 && I > 0 && O > 0, "linear: bad shape");
    const long waves = (long)B * O;
    hipLaunchKernelGGL(linear_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, ldx, w, bias, y, ldy, B, I, O,
                       wscale, bscale, add, act, gain, square_input);
    DGE_LAUNCH_CHECK("linear");
    return 0;
}

extern "C" int dge_pixelnorm(const float* x, float* y, int B, int D, float eps, hipStream_t s) {
    hipLaunchKernelGGL(pixelnorm_kernel, dim3((B + 3) / 4), dim3(256), 0, s, x, y, B, D, eps);
    DGE_LAUNCH_CHECK("pixelnorm");
    return 0;
}

extern "C" int dge_dense_chain(const float* x, int ldx, const dge_dense_layer* layers, int n, float* y, int ldy, int B, int pixelnorm,
                               float eps, hipStream_t s) {
    DGE_CHECK(x && y && layers && n >= 1 && n <= 8 && B >= 1, "dense_chain: 1 .. 8 layers");
    DenseChain c;
    c.n = n;
    for (int l = 0; l < n; l++) {
        const dge_dense_layer& L = layers[l];
        DGE_CHECK(L.w && L.I >= 1 && L.I <= 1024 && L.O >= 1 && L.O <= 1024 && (l == 0 || L.I == layers[l - 1].O), "dense_chain: layer %d: widths up to 1024, I = previous O", l);
        DGE_CHECK(L.act == DGE_ACT_NONE || L.act == DGE_ACT_LRELU || L.act == DGE_ACT_RELU, "dense_chain: activation %d", L.act);
        c.w[l] = L.w; c.bias[l] = L.bias; c.I[l] = L.I; c.O[l] = L.O;
        c.act[l] = L.act == DGE_ACT_LRELU ? LIN_ACT_LRELU : (L.act == DGE_ACT_RELU ? LIN_ACT_RELU : LIN_ACT_NONE);
        c.wscale[l] = L.wscale; c.bscale[l] = L.bscale; c.add[l] = L.add; c.gain[l] = L.gain;
    }
    hipLaunchKernelGGL(dense_chain_kernel, dim3(B), dim3(1024), 0, s, x, ldx, c, y, ldy, pixelnorm ? 1 : 0, eps);
    DGE_LAUNCH_CHECK("dense_chain");
    return 0;
}

extern "C" int dge_truncation(const float* w, const float* w_avg, float* wp, int B, int L, int D, float psi, int layers,
                              int w_is_wp, hipStream_t s) {
    const long n = (long)B * L * D;
    const int apply = (psi < 1.0f && layers > 0) ? 1 : 0;
    hipLaunchKernelGGL(truncation_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, w_avg, wp, B, L, D, psi,
                       layers, w_is_wp, apply);
    DGE_LAUNCH_CHECK("truncation");
    return 0;
}

extern "C" int dge_torgb(const void* x, const float* wrgb, const float* style, const float* bias, const float* prev,
                         float* img, int B, int H, int W, int cin, float wscale, int dtype, hipStream_t s) {
    const int esz = dtype == DGE_BF16 ? 2 : 4;
    DGE_CHECK(cin % (16 / esz) == 0, "torgb: Cin=%d not a multiple of %d", cin, 16 / esz);
    const int hw = H * W;
    if (cin >= 256 && cin % (64 * (16 / esz)) == 0 && (long)B * hw <= (1L << 16) && !dge_env().torgb_thread) {
        // low resolutions: one wave per pixel (the channels are the only parallelism a 4^2 .. 64^2 image offers)
        const unsigned nblk = (unsigned)(((long)B * hw + 3) / 4);
        if (dtype == DGE_BF16)
            hipLaunchKernelGGL((torgb_wave_kernel<bf16_t, 3>), dim3(nblk), dim3(256), 0, s, (const bf16_t*)x, wrgb, style, bias, prev, img, B, H, W, cin, wscale);
        else
            hipLaunchKernelGGL((torgb_wave_kernel<float, 3>), dim3(nblk), dim3(256), 0, s, (const float*)x, wrgb, style, bias, prev, img, B, H, W, cin, wscale);
        DGE_LAUNCH_CHECK("torgb");
        return 0;
    }
    int gx = (hw + 255) / 256; if (gx > 2048) gx = 2048;
    const size_t shm = (size_t)3 * cin * sizeof(float);
    if (dtype == DGE_BF16)
        hipLaunchKernelGGL((torgb_kernel<bf16_t, 3>), dim3(gx, B), dim3(256), shm, s, (const bf16_t*)x, wrgb, style, bias, prev, img, B, H, W, cin, wscale);
    else
        hipLaunchKernelGGL((torgb_kernel<float, 3>), dim3(gx, B), dim3(256), shm, s, (const float*)x, wrgb, style, bias, prev, img, B, H, W, cin, wscale);
    DGE_LAUNCH_CHECK("torgb");
    return 0;
}

// img += up2(prev): four consecutive output pixels of a row per thread (they read prev columns x/2-1 .. x/2+2 of two prev rows)
__global__ __launch_bounds__(256) void rgb_upsample_add_kernel(float* __restrict__ img, const float* __restrict__ prev, int H, int W) {
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x >= W) return;
    const int y = blockIdx.y, bc = blockIdx.z;
    const int h2 = H >> 1, w2 = W >> 1;
    const float* pp = prev + (size_t)bc * h2 * w2;
    const int my = y >> 1, mx = x >> 1;
    const int ya = (y & 1) ? my : my - 1, yb = ya + 1;
    const float wya = (y & 1) ? 0.75f : 0.25f, wyb = 1.f - wya;
    float v[4];                       // vertical blend of prev columns mx-1 .. mx+2
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int xq = mx - 1 + j;
        const bool xin = xq >= 0 && xq < w2;
        const float pa = (xin && ya >= 0) ? pp[(size_t)ya * w2 + xq] : 0.f, pb = (xin && yb < h2) ? pp[(size_t)yb * w2 + xq] : 0.f;
        v[j] = wya * pa + wyb * pb;
    }
    float4* ip = (float4*)(img + ((size_t)bc * H + y) * W + x);
    float4 o = *ip;
    // even output x: .25*p[m-1] + .75*p[m];  odd: .75*p[m] + .25*p[m+1]
    o.x += 0.25f * v[0] + 0.75f * v[1]; o.y += 0.75f * v[1] + 0.25f * v[2];
    o.z += 0.25f * v[1] + 0.75f * v[2]; o.w += 0.75f * v[2] + 0.25f * v[3];
    *ip = o;
}
extern "C" int dge_rgb_upsample_add(float* img, const float* prev, int BC, int H, int W, hipStream_t s) {
    DGE_CHECK(img && prev && BC > 0 && H > 0 && W % 4 == 0 && H % 2 == 0 && H <= 65535 && BC <= 65535, "rgb_upsample_add: needs W %% 4 == 0, even H");
    hipLaunchKernelGGL(rgb_upsample_add_kernel, dim3((W / 4 + 255) / 256, H, BC), dim3(256), 0, s, img, prev, H, W);
    DGE_LAUNCH_CHECK("rgb_upsample_add");
    return 0;
}

extern "C" int dge_nchw_to_nhwc(const float* src, void* dst, int B, int C, int HW, int src_B, int dtype, hipStream_t s) {
    const long n = (long)B * C * HW;
    if (dtype == DGE_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, (bf16_t*)dst, B, C, HW, src_B);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, (float*)dst, B, C, HW, src_B);
    DGE_LAUNCH_CHECK("nchw_to_nhwc");
    return 0;
}

extern "C" int dge_nhwc_to_nchw(const void* src, float* dst, int B, int C, int HW, int dtype, hipStream_t s) {
    const long n = (long)B * C * HW;
    if (dtype == DGE_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16_t*)src, dst, B, C, HW);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)src, dst, B, C, HW);
    DGE_LAUNCH_CHECK("nhwc_to_nchw");
    return 0;
}
