// dge_fold_multi: every per-sample weight image of one synthesis pass in ONE launch.
//
// The generator layers that run on conv_pp / up_pp / up_s4 read the reference's FUSED modulation (stylegan2_generator.py:858-875): one
// weight image per sample with style, demodulation and gain folded in (dge_pack_conv_pp, dge_pack_up_pp).  All styles and demodulation
// factors of a pass exist before its first layer runs (two grouped launches), so the eight folds of a StyleGAN2-1024 pass (layers 5 / 7 /
// 9 / 11 / 13 and 8 / 10 / 12: 150 MB of images at batch 8) need not be eight launches of 7 - 19 us between the layers: they are one
// grid whose blocks find their entry in a table passed BY VALUE (no device table, nothing to upload, safe under hipGraph capture).
// Bit-identical to the single launches (same association of the products; tests/test_fold_multi_gpu.py).
#include <stdlib.h>
#include "common.h"
#include "../../include/dge_hip.h"

namespace {

constexpr int MAXE = DGE_FOLD_MAX_ENTRIES;
struct FoldTab {
    dge_fold_entry e[MAXE];
    int first[MAXE + 1];          // first block of entry i (prefix sums)
    int nbx[MAXE], gy[MAXE];
    int n;
};

// slot-local unit order of up_pp's weight block -> dge_pack_upconv_weight's unit index (csrc/up_pp.hip: unit_q)
__device__ __forceinline__ int up_unit_q(int u) { return u == 0 ? 0 : u == 1 ? 4 : u == 2 ? 6 : u == 3 ? 8 : u == 4 ? 2 : u == 5 ? 5 : u == 6 ? 3 : u == 7 ? 1 : 7; }

// conv_pp image of one (N tile, K chunk, 16-row piece) for the samples by, by + gy, ...   (csrc/conv_pp.hip: conv_pp_pack_kernel, mode 0)
__device__ void fold_pp(const dge_fold_entry& E, int bid, int by, int gy, float (*wl)[9][36]) {
    const int N = E.N, K = E.K;
    const int nchunks = K / 32, ntn = N / 128;
    const int pc = bid % 8; bid /= 8;
    const int kc = bid % nchunks;
    const int nt = bid / nchunks;
    const int n0 = nt * 128 + pc * 16, k0 = kc * 32;
    const int tid = threadIdx.x;
    const float* w = (const float*)E.w;
    for (int idx = tid; idx < 16 * 288; idx += 256) {
        const int r = idx / 288, e = idx - r * 288;
        wl[r][e % 9][e / 9] = w[((size_t)(n0 + r) * K + k0) * 9 + e] * E.wscale;
    }
    __syncthreads();
    const int qd = (tid >> 4) & 3, r = tid & 15, tq = tid >> 6;
    bf16_t* out = (bf16_t*)E.out;
    for (int b = by; b < E.nb; b += gy) {
        float m[8];
        const float on = E.gain * (E.out_scale ? E.out_scale[(size_t)b * N + n0 + r] : 1.f);
#pragma unroll
        for (int j = 0; j < 8; j++) m[j] = (E.in_scale ? E.in_scale[(size_t)b * K + k0 + qd * 8 + j] : 1.f) * on;
        bf16_t* ob = out + ((((size_t)b * ntn + nt) * nchunks + kc) * 9 * 8 + pc) * 512 + (qd * 16 + r) * 8;
#pragma unroll
        for (int ti = 0; ti < 3; ti++) {
            const int t = tq + 4 * ti;
            if (t < 9) {
                const float4 a = *(const float4*)&wl[r][t][qd * 8], c = *(const float4*)&wl[r][t][qd * 8 + 4];
                const float v[8] = {a.x * m[0], a.y * m[1], a.z * m[2], a.w * m[3], c.x * m[4], c.y * m[5], c.z * m[6], c.w * m[7]};
                *(uint4*)(ob + (size_t)t * 8 * 512) = pack16(v, (bf16_t*)nullptr);
            }
        }
    }
}

// up_pp image: four 1 KiB pieces per block (csrc/up_pp.hip: up_pp_pack_kernel)
__device__ void fold_up(const dge_fold_entry& E, int bx, int by, int gy) {
    const int Cout = E.N, Cin = E.K;
    const int nchunks = Cin / 32, ntn = Cout / 32;
    int bid = bx * 4 + (threadIdx.x >> 6);
    const int l = threadIdx.x & 63, qd = l >> 4, r = l & 15;
    const int pc = bid % 18; bid /= 18;
    const int kc = bid % nchunks;
    const int nt = bid / nchunks;
    if (nt >= ntn) return;
    const int u = pc >> 1, o = nt * 32 + (pc & 1) * 16 + r, k0 = kc * 32 + qd * 8;
    const uint4 wv = *(const uint4*)((const bf16_t*)E.w + ((size_t)up_unit_q(u) * Cout + o) * Cin + k0);
    float f[8];
    unpack16(wv, f, (bf16_t*)nullptr);
    bf16_t* out = (bf16_t*)E.out;
    for (int b = by; b < E.nb; b += gy) {
        const float on = E.gain * (E.out_scale ? E.out_scale[(size_t)b * Cout + o] : 1.f);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = f[e] * ((E.in_scale ? E.in_scale[(size_t)b * Cin + k0 + e] : 1.f) * on);
        *(uint4*)(out + ((((size_t)b * ntn + nt) * nchunks + kc) * 18 + pc) * 512 + (qd * 16 + r) * 8) = pack16(v, (bf16_t*)nullptr);
    }
}

__global__ __launch_bounds__(256) void fold_multi_kernel(const FoldTab T) {
    __shared__ __attribute__((aligned(16))) float wl[16][9][36];
    int i = 0;
    while (i + 1 < T.n && (int)blockIdx.x >= T.first[i + 1]) i++;
    const int local = blockIdx.x - T.first[i];
    const int bx = local % T.nbx[i], by = local / T.nbx[i];
    if (T.e[i].kind == DGE_FOLD_CONV_PP) fold_pp(T.e[i], bx, by, T.gy[i], wl);
    else fold_up(T.e[i], bx, by, T.gy[i]);
}

}  // namespace

extern "C" int dge_fold_multi(const dge_fold_entry* entries, int n, hipStream_t s) {
    DGE_CHECK(entries && n >= 1 && n <= MAXE, "fold_multi: 1 .. %d entries, got %d", MAXE, n);
    FoldTab T;
    T.n = n;
    long total = 0;
    for (int i = 0; i < n; i++) {
        const dge_fold_entry& e = entries[i];
        DGE_CHECK(e.w && e.out && e.nb >= 1, "fold_multi: entry %d: null tensor / no copies", i);
        DGE_CHECK(e.nb == 1 || e.in_scale || e.out_scale, "fold_multi: entry %d: per-sample copies need a per-sample scale", i);
        long nbx;
        if (e.kind == DGE_FOLD_CONV_PP) {
            DGE_CHECK(e.N % 128 == 0 && e.K % 32 == 0, "fold_multi: entry %d: conv_pp image needs N=%d a multiple of 128 and K=%d of 32", i, e.N, e.K);
            nbx = (long)(e.N / 128) * (e.K / 32) * 8;
        } else {
            DGE_CHECK(e.kind == DGE_FOLD_UP_PP && e.N % 32 == 0 && e.K % 32 == 0, "fold_multi: entry %d: kind %d, up_pp image needs Cout=%d and Cin=%d multiples of 32", i, e.kind, e.N, e.K);
            nbx = ((long)(e.N / 32) * (e.K / 32) * 18 + 3) / 4;
        }
        // two samples per block on large images (the source piece is read once for both), one otherwise - as the single launches do
        const int gy = nbx >= 256 ? (e.nb + 1) / 2 : e.nb;
        T.e[i] = e; T.first[i] = (int)total; T.nbx[i] = (int)nbx; T.gy[i] = gy;
        total += nbx * gy;
        DGE_CHECK(total < (1l << 30), "fold_multi: grid too large");
    }
    T.first[n] = (int)total;
    hipLaunchKernelGGL(fold_multi_kernel, dim3((unsigned)total), dim3(256), 0, s, T);
    DGE_LAUNCH_CHECK("fold_multi");
    return 0;
}
