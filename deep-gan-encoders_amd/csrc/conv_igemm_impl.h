// Implicit-GEMM 3x3 / 1x1 convolution for gfx950 (MI355X), NHWC activations.
//
//   M = output pixels of a TH x TW spatial tile, N = output channels, K = taps x Cin.
//   The input halo tile ((TH+2) x (TW+2) x KC channels) is staged ONCE per K-chunk into LDS
//   and re-read at 9 shifted offsets (one per tap), so activations cross HBM/L2 once, not 9x.
//   Weights are pre-packed [tap][N][Cin] (K contiguous) so both MFMA operands are read from
//   LDS as 16-byte K-contiguous fragments (ds_read_b128, rows padded by 16 B: conflict free).
//   MFMA: v_mfma_f32_32x32x16_bf16 (bf16 storage) or v_mfma_f32_32x32x2_f32 (f32 storage,
//   exact f32 - the parity path); 64-wide wavefronts, 4 waves per workgroup.
//
//   Pipeline: one stage = one kernel ROW (3 taps) of one K chunk.  While the MFMAs of stage s run,
//   the weight tile of stage s+1 (and, at the first row of a chunk, the next chunk's input halo
//   tile) is in flight from L2/HBM into registers; it is written to the other LDS buffer after the
//   MFMAs and ONE barrier closes the stage.  A and B are both double-buffered in LDS.
//
// Fused prologue : per-(sample, in-channel) affine a*x+b applied while staging
//                  (StyleGAN2 style modulation s[b,i]; instance-norm apply in the encoder).
// Fused epilogue : per-(sample, out-channel) scale (demodulation d[b,o]), noise*weight, bias,
//                  lrelu/relu, gain, optional addend (residual blend), optional per-(b,c)
//                  sum / sum-of-squares for the next instance norm, depth-to-space x2 store
//                  (the folded transposed-conv+FIR up layer: N = 4 phases x Cout).
//
// Reference math: model/stylegan2_generator.py:855-922 (ModulateConvBlock.forward, shared-weight
// form :876-877,:908-909), model/E/E.py:50-85 (BEBlock.forward).
#pragma once
#include <type_traits>
#include <stdlib.h>
#include "common.h"
#include "conv_params.h"
#include "conv_epilogue.h"

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&a, *(const bf16x8_t*)&b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

// N uint4 values with compile-time-only indexing (keeps prefetch buffers in VGPRs: a plain array
// indexed inside a lambda was being demoted to scratch / LDS by the compiler).
template <int N> struct Regs {
    uint4 v; Regs<N - 1> rest;
    template <int I> __device__ __forceinline__ uint4& get() { if constexpr (I == 0) return v; else return rest.template get<I - 1>(); }
};
template <> struct Regs<0> { template <int I> __device__ __forceinline__ uint4& get(); };

// 16-byte-per-lane global -> LDS DMA issued from inline asm, so that hipcc does NOT track it: with the
// builtin form the compiler inserts s_waitcnt vmcnt(0) in front of the next ds_read (it cannot prove
// the read does not alias the DMA destination), which serialises the weight stream behind the MFMAs.
// The LDS destination is wave-uniform base (M0) + lane*16; completion is awaited by hand
// (s_waitcnt vmcnt(0) right before the stage barrier).
__device__ __forceinline__ void glds16_untracked(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
// the same with a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset: no per-lane 64-bit address arithmetic
__device__ __forceinline__ void glds16_saddr(unsigned voff, unsigned long long sbase, unsigned lds_dst_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ unsigned lds_offset_of(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

extern "C" int dge_get_deterministic(void);        // capi.hip

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int rup(int a, int b) { return (a + b - 1) / b * b; }

static constexpr int cmin(int a, int b) { return a < b ? a : b; }
// ---- transposed-accumulator epilogue (kernel MODE bit 5).  MFMA C layout with the weights as A operand: lane (pixel = lane & 31,
// kh = lane >> 5) holds rows 8j + 4kh + i in register 4j + i; with the weight rows permuted by tr_chan_of_row, registers 0-7 are
// channels 8kh .. 8kh+7 and registers 8-15 channels 16+8kh .. 16+8kh+7 of the 32-channel tile (conv_stream.hip uses the same map).
__device__ __forceinline__ int tr_chan_of_row(int m) {
    const int j = m >> 3, k = (m >> 2) & 1, i = m & 3;
    return 16 * (j >> 1) + 8 * k + 4 * (j & 1) + i;
}
// Stages: demodulation scale, noise, bias, activation, gain (model/stylegan2_generator.py:908-921; LPIPS conv + ReLU) - the
// forward menu of conv_epilogue MODE 0 without statistics and without the depth-to-space store.
template <typename T, class C, int TH, int TW, int BN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue_tr(const ConvParams& p, f32x16_t (&acc)[C::MT][C::NT], const float* ldsN,
                                                 int b, int x0, int y0, int bn0, int wave, int lane) {
    static_assert(sizeof(T) == 2, "bf16 storage");
    const int wm = wave / WN, wn = wave % WN;
    const int px = lane & 31, kh = lane >> 5;
    const float slope = p.act == DGE_ACT_LRELU ? 0.2f : (p.act == DGE_ACT_RELU ? 0.f : 1.f);
    T* __restrict__ Yb = (T*)p.y + (size_t)b * p.H * p.W * p.Cout;           // in-image offsets fit 32 bits
    StaticFor<C::NT>::run([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int n0 = bn0 + wn * C::WTN + j * 32;
        if (n0 >= p.Ntot_valid) return;                                        // wave-uniform
        float osc[16], bia[16], nw[16];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int c0 = n0 + 16 * h + 8 * kh;
            const bool cv = c0 < p.Cout;                                       // Cout is a multiple of 8: a run is valid or empty
#pragma unroll
            for (int e4 = 0; e4 < 2; e4++) {
                float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.out_scale && cv) s4 = *(const float4*)(p.out_scale + (size_t)b * p.Cout + c0 + e4 * 4);
                if (p.bias && cv) b4 = *(const float4*)(p.bias + c0 + e4 * 4);
                const int r = 8 * h + 4 * e4;
                osc[r] = s4.x * p.gain; osc[r + 1] = s4.y * p.gain; osc[r + 2] = s4.z * p.gain; osc[r + 3] = s4.w * p.gain;
                const float bg = p.bias_scale * p.gain;
                bia[r] = b4.x * bg; bia[r + 1] = b4.y * bg; bia[r + 2] = b4.z * bg; bia[r + 3] = b4.w * bg;
#pragma unroll
                for (int e = 0; e < 4; e++) nw[r + e] = (p.noise && cv) ? p.noise_w[(c0 + e4 * 4 + e) * p.noise_w_stride] * p.gain : 0.f;
            }
        }
        StaticFor<C::MT>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int m = wm * C::WTM + i * 32 + px;
            const int gy = y0 + m / TW, gx = x0 + m % TW;
            const float nz = p.noise ? ldsN[m] : 0.f;
            const f32x16_t a = acc[i][j];
            float v0[8], v1[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float u0 = fmaf(a[r], osc[r], fmaf(nw[r], nz, bia[r]));
                const float u1 = fmaf(a[8 + r], osc[8 + r], fmaf(nw[8 + r], nz, bia[8 + r]));
                v0[r] = fmaxf(u0, u0 * slope); v1[r] = fmaxf(u1, u1 * slope);
            }
            if ((gy < p.H) & (gx < p.W)) {
                T* yp = Yb + (gy * p.W + gx) * p.Cout + n0 + 8 * kh;
                if (n0 + 8 * kh < p.Cout) *(uint4*)yp = pack16(v0, (T*)nullptr);
                if (n0 + 16 + 8 * kh < p.Cout) *(uint4*)(yp + 16) = pack16(v1, (T*)nullptr);
            }
        });
    });
}

// ---- transposed-accumulator epilogue of the data-gradient launches (kernel MODE 33 = bit 5 + epilogue mode 1): residual addend,
// dot products against dot_src (sum f*d, sum f per (sample, channel); the per-channel scale is applied after them, as in
// conv_epilogue), ReLU backward of the layer below (mask_relu), statistics of results that include the addend.  A lane owns two
// runs of 8 channels of one pixel, so addend / dot_src arrive as the same 16-byte vectors the result leaves in, requested one
// tile ahead; the 2 x 16 per-lane sums are reduced over the 32 pixel lanes by halving (16 cross-lane moves per sum instead of
// 80) and leave as ONE atomic instruction per sum and N tile.  Launches with bias / noise / activation / up / prep / the
// deterministic mode keep conv_epilogue.
__device__ __forceinline__ float tr_lane_reduce16(float (&v)[16], int lane) {
    // v[q] summed over the 32 lanes of a half wave; returns the total of value q = 8 b4 + 4 b3 + 2 b2 + b1 (b = bits of lane & 31)
    const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
    float w[8], x[4], y[2];
#pragma unroll
    for (int q = 0; q < 8; q++) { const float snd = b4 ? v[q] : v[q + 8]; w[q] = (b4 ? v[q + 8] : v[q]) + __shfl_xor(snd, 16, 64); }
#pragma unroll
    for (int q = 0; q < 4; q++) { const float snd = b3 ? w[q] : w[q + 4]; x[q] = (b3 ? w[q + 4] : w[q]) + __shfl_xor(snd, 8, 64); }
#pragma unroll
    for (int q = 0; q < 2; q++) { const float snd = b2 ? x[q] : x[q + 2]; y[q] = (b2 ? x[q + 2] : x[q]) + __shfl_xor(snd, 4, 64); }
    const float snd = b1 ? y[0] : y[1];
    float z = (b1 ? y[1] : y[0]) + __shfl_xor(snd, 2, 64);
    z += __shfl_xor(z, 1, 64);
    return z;
}
template <typename T, class C, int TH, int TW, int BN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue_tr_da(const ConvParams& p, f32x16_t (&acc)[C::MT][C::NT], int b, int x0, int y0, int bn0,
                                                    int vbid, int wave, int lane) {
    static_assert(sizeof(T) == 2, "bf16 storage");
    const int wm = wave / WN, wn = wave % WN;
    const int px = lane & 31, kh = lane >> 5;
    T* __restrict__ Yb = (T*)p.y + (size_t)b * p.H * p.W * p.Cout;           // in-image offsets fit 32 bits
    const T* __restrict__ ADDb = p.addend ? (const T*)p.addend + (size_t)b * p.H * p.W * p.Cout : nullptr;
    const T* __restrict__ DOTb = p.dot_src ? (const T*)p.dot_src + (size_t)b * p.H * p.W * p.Cout : nullptr;
    const bool dot = DOTb != nullptr, add = ADDb != nullptr;
    const bool stats = p.stats != nullptr;
    float* __restrict__ STATS = stats ? p.stats + (size_t)(vbid % p.stats_slots) * p.B * p.Cout * 2 : nullptr;
    constexpr int NTILE = C::MT * C::NT;
    uint4 pd[2][2], pa[2][2];
    auto tile_off = [&](int i, int j, bool& ok) {
        const int m = wm * C::WTM + i * 32 + px;
        const int gy = y0 + m / TW, gx = x0 + m % TW;
        ok = (gy < p.H) & (gx < p.W);
        return (gy * p.W + gx) * p.Cout + bn0 + wn * C::WTN + j * 32 + 8 * kh;
    };
    auto issue = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int j = t / C::MT, i = t % C::MT;
        bool ok;
        const int off = tile_off(i, j, ok);
        const int c0 = bn0 + wn * C::WTN + j * 32 + 8 * kh;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            pd[t & 1][h] = make_uint4(0, 0, 0, 0); pa[t & 1][h] = make_uint4(0, 0, 0, 0);
            if (ok && c0 + 16 * h < p.Cout) {
                if (dot) pd[t & 1][h] = *(const uint4*)(DOTb + off + 16 * h);
                if (add) pa[t & 1][h] = *(const uint4*)(ADDb + off + 16 * h);
            }
        }
    };
    issue(std::integral_constant<int, 0>{});
    StaticFor<C::NT>::run([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int n0 = bn0 + wn * C::WTN + j * 32;
        float sc[16], ps0[16], ps1[16];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int c0 = n0 + 16 * h + 8 * kh;
            const bool cv = c0 < p.Cout && n0 < p.Ntot_valid;
#pragma unroll
            for (int e4 = 0; e4 < 2; e4++) {
                float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f);
                if (p.out_scale && cv) s4 = *(const float4*)(p.out_scale + (size_t)b * p.Cout + c0 + e4 * 4);
                const int r = 8 * h + 4 * e4;
                sc[r] = s4.x; sc[r + 1] = s4.y; sc[r + 2] = s4.z; sc[r + 3] = s4.w;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) { ps0[r] = 0.f; ps1[r] = 0.f; }
        StaticFor<C::MT>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int t = j * C::MT + i;
            if constexpr (t + 1 < NTILE) issue(std::integral_constant<int, t + 1>{});
            bool ok;
            const int off = tile_off(i, j, ok);
            const f32x16_t a = acc[i][j];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                float f[8], d[8], ad[8];
                unpack16(pd[t & 1][h], d, (T*)nullptr);
                unpack16(pa[t & 1][h], ad, (T*)nullptr);
                const bool rv = ok && (n0 + 16 * h + 8 * kh < p.Cout) && (n0 < p.Ntot_valid);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float v = rv ? a[8 * h + e] * p.gain : 0.f;
                    if (dot) { ps0[8 * h + e] = fmaf(v, d[e], ps0[8 * h + e]); ps1[8 * h + e] += v; }
                    v *= sc[8 * h + e];
                    if (add) v = fmaf(p.add_scale, ad[e], v);
                    if (dot && p.mask_relu) v = d[e] > 0.f ? v : 0.f;
                    if (!dot && stats && rv) { ps0[8 * h + e] += v; ps1[8 * h + e] = fmaf(v, v, ps1[8 * h + e]); }
                    f[e] = v;
                }
                if (rv) *(uint4*)(Yb + off + 16 * h) = pack16(f, (T*)nullptr);
            }
        });
        if (stats) {
            const float t0 = tr_lane_reduce16(ps0, lane), t1 = tr_lane_reduce16(ps1, lane);
            const int q = (lane >> 1) & 15;                                   // value index this lane's totals belong to
            const int c = n0 + 16 * (q >> 3) + 8 * kh + (q & 7);
            if (!(lane & 1) && c < p.Cout && n0 < p.Ntot_valid) {
                atomicAdd(STATS + ((size_t)b * p.Cout + c) * 2, t0);
                atomicAdd(STATS + ((size_t)b * p.Cout + c) * 2 + 1, t1);
            }
        }
    });
}

// OPT bit 1 (N1): single-phase noise tile - the launch is not in up mode (lets the 32 x 16 pixel tile keep two workgroups per CU).
// (A double-buffered halo tile - OPT bit 0 in the round-3 experiments: next chunk's tile written during the current chunk, no
//  boundary barrier - measured +-2 % on every deep-K 64-wide launch and was removed: DESIGN 6e.)
template <typename T, int TH, int TW, int BN, int KC, int KS, int WM, int WN, int OPT = 0>
struct ConvCfg {
    static constexpr int N1 = (OPT >> 1) & 1;                // single-phase noise tile (launch is not in up mode)
    static constexpr int NB2 = (OPT >> 2) & 1;               // weight ring 2 deep: with N1 the 64-wide tile fits three workgroups per CU
    static constexpr int BM = TH * TW;
    static constexpr int WTM = BM / WM, WTN = BN / WN;
    static constexpr int MT = WTM / 32, NT = WTN / 32;
    static constexpr int KCB = KC * (int)sizeof(T);
    static constexpr int CH = KCB / 16;
    static constexpr int PSTR = KCB + 16;
    static constexpr int HALO = KS / 2;
    static constexpr int HH = TH + 2 * HALO, HW = TW + 2 * HALO;
    static constexpr int RPITCH = rup(HW * PSTR, 256);
    static constexpr int A_BYTES = HH * RPITCH;              // one halo tile
    static constexpr int B_BYTES = KS * BN * KCB;            // one stage: KS taps (a kernel row), rows unpadded, 16-B chunks XOR-swizzled
    static constexpr int RP = 16 / CH;                       // weight rows per 256-byte LDS bank row
    static constexpr int B_PIECES = B_BYTES / 1024;          // 1 KiB wave-level LDS-DMA pieces per stage
    static constexpr int ESTR = 32 * 4 + 16;                  // epilogue staging is always f32
    static constexpr int E_BYTES = 4 * 32 * ESTR;
    // noise values of the tile (x4 phases in up mode; up mode has N = 4*Cout >= 64, so 32-wide N tiles never see it and
    // keep 3 KB: that puts the 16x16x32 configuration under the 3-workgroups-per-CU LDS line)
    static constexpr int N_BYTES = ((BN >= 64 && !N1) ? 4 : 1) * BM * 4;
    static constexpr int A_BUFS = 1;
    // weight-stage ring.  Large pixel tiles: 4 deep (DMA issued 3 stages ahead) when two workgroups still fit a CU, else
    // 2 deep.  Small pixel tiles serve the low-resolution layers, whose grids do not fill the chip and whose stages are
    // short (6-12 MFMAs per wave): there the serial K loop is bound by the L2 latency of the weight stream, so the ring
    // takes the whole LDS (one workgroup per CU) and runs up to 5 stages ahead.
    static constexpr int NBUF_SMALL = cmin(6, cmax(2, (150 * 1024 - A_BUFS * A_BYTES - N_BYTES) / B_BYTES));
    static constexpr int NBUF = NB2 ? 2 : ((BM <= 128) ? NBUF_SMALL : ((A_BUFS * A_BYTES + 4 * B_BYTES + N_BYTES <= 80 * 1024) ? 4 : 2));
    static constexpr int DPW = B_PIECES / 4;                   // DMA instructions every wave issues per stage (floor)
    static constexpr int LDS_BYTES = cmax(A_BUFS * A_BYTES + NBUF * B_BYTES, E_BYTES) + N_BYTES;
    static constexpr int NA_ITEMS = HH * HW * CH, NA_PER = (NA_ITEMS + 255) / 256;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(MT >= 1 && NT >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
    static_assert(KCB % 32 == 0 && 256 % CH == 0, "K chunk");
    static_assert(B_BYTES % 1024 == 0 && BN % 16 == 0 && (BN / RP) % CH == 0, "weight stage must be whole 1 KiB pieces");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    // workgroups per CU the LDS footprint allows (1..3) = waves per SIMD to ask the register allocator for
    // (a 128-wide N tile holds 128 accumulator registers per lane: never ask for more than 2 waves/SIMD there)
    static constexpr int MINW = (LDS_BYTES <= 53 * 1024 && MT * NT < 8) ? 3 : (LDS_BYTES <= 80 * 1024 ? 2 : 1);
};

template <typename T, int TH, int TW, int BN, int KC, int KS, int WM, int WN, int MODE>
__global__ __launch_bounds__(256, (ConvCfg<T, TH, TW, BN, KC, KS, WM, WN, ((MODE >> 3) & 3) | (((MODE >> 6) & 1) << 2)>::MINW)) void conv_igemm_kernel(ConvParams p) {
    using C = ConvCfg<T, TH, TW, BN, KC, KS, WM, WN, ((MODE >> 3) & 3) | (((MODE >> 6) & 1) << 2)>;
    // MODE: bits 0-1 = epilogue mode (conv_epilogue.h); bit 2 = in_t2d (phase-form adjoint of the up layer: compile time, because a
    // run-time choice between two unrolled step sequences in the main loop doubled its code and spilled the accumulators);
    // bits 3-4 = ConvCfg OPT
    constexpr bool T2D = (MODE & 4) != 0;
    // bit 5 = transposed accumulators (TR): the WEIGHT fragment is the MFMA A operand, its rows permuted (tr_chan_of_row) so that
    // a lane ends up with two runs of 8 consecutive channels of ONE pixel - the result leaves as 16-byte stores straight from
    // the registers (conv_epilogue_tr: no LDS transpose).  Offered for the plain forward epilogue (mode 0, no statistics, no up).
    constexpr bool TR = (MODE & 32) != 0;
    constexpr int EMODE = MODE & 3;
    constexpr int EP16 = Elem<T>::PER16;
    __shared__ __attribute__((aligned(256))) unsigned char lds[C::LDS_BYTES];
    unsigned char* ldsA = lds;                         // halo tile of the current K chunk
    unsigned char* ldsB = lds + C::A_BUFS * C::A_BYTES;   // weight stages
    float* ldsN = (float*)(lds + C::LDS_BYTES - C::N_BYTES);   // noise tile, lives until the epilogue
    const unsigned ldsB_off = lds_offset_of(ldsB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: keeps tile/DMA index math on the scalar unit
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (private L2 each), so workgroup id i runs on
    // XCD i % 8.  Give every XCD a CONTIGUOUS range of tiles: neighbouring tiles share halo rows and the same weight
    // slice, which then hit in that XCD's L2 instead of being fetched once per XCD.
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, per = nb >> 3;
        if (per > 0 && bid < (per << 3) && !(p.dbg & 32)) bid = (bid & 7) * per + (bid >> 3);
    }
    const int vbid = bid;             // tile id in (tx, ty, b, ntile) order: statistics slots are dealt by THIS id, so that the
                                      // workgroups sharing a slot stay spread over the samples (same-address atomics)
    const int tx_i = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty_i = bid % p.tiles_y; bid /= p.tiles_y;
    const int b = bid % p.B;
    const int ntile = bid / p.B;
    const int x0 = tx_i * TW, y0 = ty_i * TH;
    const int bn0 = ntile * BN;
    const T* __restrict__ X = (const T*)p.x;
    const T* __restrict__ Wp = (const T*)p.w;

    f32x16_t acc[C::MT][C::NT];
#pragma unroll
    for (int i = 0; i < C::MT; i++)
#pragma unroll
        for (int j = 0; j < C::NT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    int aoff[C::MT];
#pragma unroll
    for (int i = 0; i < C::MT; i++) {
        const int m = wm * C::WTM + i * 32 + (lane & 31);
        aoff[i] = (m / TW) * C::RPITCH + (m % TW) * C::PSTR + (lane >> 5) * 16;
    }
    int bbase[C::NT], bsw[C::NT];           // weight row byte offset and its chunk swizzle
#pragma unroll
    for (int j = 0; j < C::NT; j++) {
        const int n = wn * C::WTN + j * 32 + (TR ? tr_chan_of_row(lane & 31) : (lane & 31));
        bbase[j] = n * C::KCB; bsw[j] = (n / C::RP) % C::CH;
    }

    const int nchunks = p.Cin / KC;
    const bool affine = p.in_scale || p.in_shift;
    const int cphys = p.in_s2d ? (p.Cin >> 2) : p.Cin;
    const int achunk = tid % C::CH;                    // the 16-byte channel sub-range this thread stages

    Regs<C::NA_PER> areg;
    float asc[EP16], ash[EP16];                        // affine of the chunk held in areg
    unsigned inmask = 0;                               // bit i: halo item i lies inside the image

    // Source geometry of the fused read modes, folded into shifts so the per-item address math is
    // branch-free: plain (sl=sh=0), in_up2 (nearest x2: source pixel = (y>>1, x>>1)), in_s2d
    // (space-to-depth: logical channel (phase, c) lives at pixel (2y+py, 2x+px) of the fine grid).
    // in_t2d (phase-form adjoint of the up layer): the input tensor has one more row and column than the output grid
    // ([B,H+1,W+1,Cin] from dge_fir_t2d) and only the taps that read pixels m, m+1 per axis are computed (below).
    const int sl = p.in_s2d ? 1 : 0, sh = p.in_up2 ? 1 : 0;
    const int Hin = p.H + (T2D ? 1 : 0), Win = p.W + (T2D ? 1 : 0);
    const int Ws = (Win << sl) >> sh;
    const T* __restrict__ Xb = X + (size_t)b * ((Hin << sl) >> sh) * Ws * cphys;   // in-image offsets fit 32 bits

    // ---- input halo tile: global -> registers (zero outside the image)
    auto load_a = [&](int kc) {
        const int cbase = kc * KC;
        int ph = 0, cb = cbase;
        if (p.in_s2d) { ph = cbase / cphys; cb = cbase - ph * cphys; }
        const int ay = ph >> 1, ax = ph & 1;
        const int coff = cb + achunk * EP16;
        if (affine) {
            // (space-to-depth reads: the affine is per physical channel, [B, Cin/4] - the four phases of a channel share it)
            const int ci = b * cphys + cb + achunk * EP16;
            if (p.in_scale) {
#pragma unroll
                for (int e4 = 0; e4 < EP16 / 4; e4++) *(float4*)&asc[e4 * 4] = *(const float4*)(p.in_scale + ci + e4 * 4);
            } else {
#pragma unroll
                for (int e = 0; e < EP16; e++) asc[e] = 1.f;
            }
            if (p.in_shift) {
#pragma unroll
                for (int e4 = 0; e4 < EP16 / 4; e4++) *(float4*)&ash[e4 * 4] = *(const float4*)(p.in_shift + ci + e4 * 4);
            } else {
#pragma unroll
                for (int e = 0; e < EP16; e++) ash[e] = 0.f;
            }
        }
        inmask = 0;
        StaticFor<C::NA_PER>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int idx = tid + i * 256;
            const int pix = idx / C::CH;
            const int hx = pix % C::HW, hy = pix / C::HW;
            const int gy = y0 + hy - C::HALO, gx = x0 + hx - C::HALO;
            const bool inside = ((unsigned)gy < (unsigned)Hin) & ((unsigned)gx < (unsigned)Win) &
                                (C::NA_ITEMS % 256 == 0 || idx < C::NA_ITEMS);
            const int sy = ((gy << sl) >> sh) + ay, sx = ((gx << sl) >> sh) + ax;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (inside) v = *(const uint4*)(Xb + (sy * Ws + sx) * cphys + coff);
            inmask |= (inside ? 1u : 0u) << i;
            areg.template get<i>() = v;
        });
    };
    // ---- registers -> LDS with the fused per-(b,c) affine (inside the image only)
    auto store_a = [&](int abuf) {
        unsigned char* __restrict__ dstA = ldsA + abuf * C::A_BYTES;
        StaticFor<C::NA_PER>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int idx = tid + i * 256;
            if (C::NA_ITEMS % 256 == 0 || idx < C::NA_ITEMS) {
                const int pix = idx / C::CH;
                const int hx = pix % C::HW, hy = pix / C::HW;
                uint4 v = areg.template get<i>();
                if (affine && ((inmask >> i) & 1u)) {     // padding stays zero
                    float f[EP16];
                    unpack16(v, f, (T*)nullptr);
#pragma unroll
                    for (int e = 0; e < EP16; e++) f[e] = f[e] * asc[e] + ash[e];
                    if (p.in_relu) {
#pragma unroll
                        for (int e = 0; e < EP16; e++) f[e] = fmaxf(f[e], 0.f);
                    }
                    v = pack16(f, (T*)nullptr);
                }
                *(uint4*)(dstA + hy * C::RPITCH + hx * C::PSTR + achunk * 16) = v;
            }
        });
    };
    // ---- weight stage (kernel row `row` of chunk kc): KS taps x BN rows x KC, global -> LDS directly
    // (global_load_lds: no VGPR round trip; the LDS image is lane-linear per 1 KiB piece, so the
    // bank-conflict swizzle is applied to the per-lane SOURCE chunk and undone by the fragment reads)
    // The per-lane part of the source address (piece, tap within the row, weight row, swizzled chunk) does not depend on the
    // stage: it is computed once as a 32-bit byte offset per piece; a stage adds a scalar base (kernel row, K chunk), so a
    // piece costs one M0 write and one instruction (the address arithmetic used to be ~18 instructions per piece, ~15 % of
    // the 128-wide configuration's time).
    constexpr int NPW = (C::B_PIECES + 3) / 4;               // pieces per wave (the last one may not exist for every wave)
    unsigned boff[NPW];
    {
        constexpr int RPP = 1024 / C::KCB;                   // weight rows per piece
#pragma unroll
        for (int k = 0; k < NPW; k++) {
            const int pc = wave + 4 * k;
            const int r = pc * RPP + lane / C::CH;           // row within the stage = t*BN + n
            const int c = (lane % C::CH) ^ ((r / C::RP) % C::CH);
            const int t = r / BN, n = r % BN;
            boff[k] = (unsigned)((((size_t)t * p.Ntot + bn0 + n) * p.Cin + c * EP16) * sizeof(T));
        }
    }
    auto dma_b = [&](int kc, int row, int buf) {
        const unsigned long long sb = (unsigned long long)Wp + ((size_t)row * KS * p.Ntot * p.Cin + (size_t)kc * KC) * sizeof(T);
        StaticFor<NPW>::run([&](auto kcst) {
            constexpr int k = decltype(kcst)::value;
            const int pc = wave + 4 * k;
            if (C::B_PIECES % 4 == 0 || pc < C::B_PIECES)
                glds16_saddr(boff[k], sb, __builtin_amdgcn_readfirstlane(ldsB_off + buf * C::B_BYTES + pc * 1024));
        });
    };

    // ---- prologue: first halo tile, first NBUF-1 weight stages, noise tile
    const int nstages = nchunks * KS;
#pragma unroll
    for (int q = 0; q < C::NBUF - 1; q++)
        if (q < nstages) dma_b(q / KS, q % KS, q);
    load_a(0);
    // noise tile of the epilogue: this layer's plane (forward), or - fused tail backward, ConvParams::prep - the plane of the layer below
    const float* __restrict__ nz_src = p.prep ? p.prep_noise : p.noise;
    if (nz_src) {
        const int nz_bs = p.prep ? p.prep_noise_bstride : p.noise_bstride;
        const int OHn = p.up ? 2 * p.H : p.H, OWn = p.up ? 2 * p.W : p.W;
        const int nph = p.up ? 4 : 1;
        for (int idx = tid; idx < nph * C::BM; idx += 256) {
            const int m = idx % C::BM, ph = idx / C::BM;
            const int gy = y0 + m / TW, gx = x0 + m % TW;
            const int oy = p.up ? 2 * gy + (ph >> 1) : gy, ox = p.up ? 2 * gx + (ph & 1) : gx;
            ldsN[idx] = (gy < p.H && gx < p.W) ? nz_src[(size_t)b * nz_bs + (size_t)oy * OWn + ox] : 0.f;
        }
    }
    store_a(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the untracked weight DMAs
    __syncthreads();

    // One K chunk = KS stages (kernel rows), unrolled so that `row` is a compile-time constant; the
    // last chunk is a separate instantiation (LAST) so that the halo prefetch of the next chunk is
    // unconditional in the steady-state loop (a conditional prefetch turns the staging registers
    // into loop-carried copies and costs 2x their count in VGPRs).
    auto chunk = [&](int kc, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        StaticFor<KS>::run([&](auto rowc) {
            constexpr int row = decltype(rowc)::value;
            const int s = kc * KS + row;
            const int sbuf = s % C::NBUF;
            const int sn = s + C::NBUF - 1;                        // stage whose weights are requested now
            if (sn < nstages && !(p.dbg & 1)) dma_b(sn / KS, sn % KS, sn % C::NBUF);   // ring slot last read in stage s-1
            if (row == 0 && !LAST) load_a(kc + 1);                 // consumed after the last row of this chunk
            const unsigned char* aa = ldsA + row * C::RPITCH;
            const unsigned char* bb = ldsB + sbuf * C::B_BYTES;
            if (!(p.dbg & 2)) {
                // Software-pipelined fragment loads: the LDS reads of step q+1 are issued before the
                // MFMAs of step q, so their latency hides behind MT*NT matrix instructions instead of
                // being exposed in front of every pair of them (a step = one 32-byte K slice of one tap).
                constexpr int KSL = C::KCB / 32, NSTEP = KS * KSL;
                uint4 af[2][C::MT], bf[2][C::NT];
                auto frag = [&](int q, uint4 (&a)[C::MT], uint4 (&bq)[C::NT]) {
                    const int t = q / KSL, ks = q % KSL;
#pragma unroll
                    for (int i = 0; i < C::MT; i++) a[i] = *(const uint4*)(aa + aoff[i] + t * C::PSTR + ks * 32);
#pragma unroll
                    for (int j = 0; j < C::NT; j++)
                        bq[j] = *(const uint4*)(bb + t * BN * C::KCB + bbase[j] + (((ks * 2 + (lane >> 5)) ^ bsw[j]) << 4));
                };
                if constexpr (!T2D) {
                    frag(0, af[0], bf[0]);
                    StaticFor<NSTEP>::run([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        if (q + 1 < NSTEP) frag(q + 1, af[(q + 1) & 1], bf[(q + 1) & 1]);
#pragma unroll
                        for (int i = 0; i < C::MT; i++)
#pragma unroll
                            for (int j = 0; j < C::NT; j++) {
                                if constexpr (TR) Mma<T>::run(bf[q & 1][j], af[q & 1][i], acc[i][j]);
                                else Mma<T>::run(af[q & 1][i], bf[q & 1][j], acc[i][j]);
                            }
                    });
                } else if constexpr (row > 0) {
                    // in_t2d: kernel row 0 and tap 0 of the other rows hold zero weights (DGE_PACK_UPT2D_DGRAD) and are skipped:
                    // 4 of 9 taps.  (Their weight stages are still requested: the ring's vmcnt accounting counts every stage.)
                    frag(KSL, af[KSL & 1], bf[KSL & 1]);
                    StaticFor<NSTEP - KSL>::run([&](auto qc) {
                        constexpr int q = decltype(qc)::value + KSL;
                        if (q + 1 < NSTEP) frag(q + 1, af[(q + 1) & 1], bf[(q + 1) & 1]);
#pragma unroll
                        for (int i = 0; i < C::MT; i++)
#pragma unroll
                            for (int j = 0; j < C::NT; j++) {
                                if constexpr (TR) Mma<T>::run(bf[q & 1][j], af[q & 1][i], acc[i][j]);
                                else Mma<T>::run(af[q & 1][i], bf[q & 1][j], acc[i][j]);
                            }
                    });
                }
            }
            // Stage s+1's weights must have landed; DMA groups requested after it may stay in flight.
            // vmcnt retires in order, so allowing (groups still wanted in flight) x DPW outstanding ops is
            // exact for the DMAs and conservative w.r.t. the ordinary halo loads interleaved with them.
            {
                int fly = nstages - 2 - s; fly = fly < 0 ? 0 : (fly > C::NBUF - 2 ? C::NBUF - 2 : fly);
                static_assert(C::NBUF <= 6 && 4 * C::DPW <= 63, "vmcnt immediate");
                if (C::DPW == 0 || fly == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (fly == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(C::DPW) : "memory");
                else if (fly == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * C::DPW) : "memory");
                else if (fly == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(3 * C::DPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(4 * C::DPW) : "memory");
            }
            __syncthreads();                                       // ... for every wave: stage closed
            if (row == KS - 1 && !LAST) {                  // chunk boundary: replace the halo tile
                store_a(0);
                __syncthreads();
            }
        });
    };
    if (!(p.dbg & 8)) {
        for (int kc = 0; kc + 1 < nchunks; kc++) chunk(kc, std::false_type{});
        chunk(nchunks - 1, std::true_type{});
    }
    // ---------------------------------------------------------------- epilogue (conv_epilogue.h)
    // LDS is re-used as the transpose buffer from here (all reads done: barrier above); the noise tile at its end stays valid
    if constexpr (TR && EMODE == 1) conv_epilogue_tr_da<T, C, TH, TW, BN, WM, WN>(p, acc, b, x0, y0, bn0, vbid, wave, lane);
    else if constexpr (TR) conv_epilogue_tr<T, C, TH, TW, BN, WM, WN>(p, acc, ldsN, b, x0, y0, bn0, wave, lane);
    else conv_epilogue<T, C, TH, TW, BN, WM, WN, 256, EMODE>(p, acc, lds, ldsN, b, x0, y0, bn0, ntile, vbid, tx_i, ty_i, wave, lane, tid, true);
}

// ------------------------------------------------------------------------- dispatch
template <typename T, int TH, int TW, int BN, int KC, int KS, int WM, int WN>
static int launch_cfg(const ConvParams& p0, hipStream_t s) {
    ConvParams p = p0;
    p.dbg = dge_env().conv_dbg;
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    const int ntiles = (p.Ntot + BN - 1) / BN;
    const long grid = (long)p.tiles_x * p.tiles_y * p.B * ntiles;

    // epilogue mode (conv_epilogue.h): 0 plain, 1 addend / dot_src with prefetch, 2 = 1 + fused tail backward (3x3 only).
    // The f32 parity path has no prefetch, so its mode 1 is its mode 0 with the stages enabled: it always takes >= 1.
    if (p.stats && !p.up)          // deterministic mode: (sample, N tile) domains x (pixel tile, wave row) slots x (sum, sum2) per channel
        DGE_CHECK(dge_det_fits((long long)p.B * ntiles, (long long)p.tiles_x * p.tiles_y * WM, BN * 2),
                  "conv: the deterministic-mode workspace is too small for this launch (%d x %d x %d floats); reduce the batch", p.B * ntiles,
                  p.tiles_x * p.tiles_y * WM, BN * 2);
    const bool da = p.addend || p.dot_src;
    // transposed accumulators + direct stores (conv_epilogue_tr): the plain forward epilogue on the 16 x 16 / 32 x 16 pixel tiles
    const bool tr = sizeof(T) == 2 && !da && !p.prep && !p.stats && !p.up && !p.in_t2d && !(p.dbg & 64);
    // ... and the data-gradient epilogue in the same layout (conv_epilogue_tr_da), 64-wide tiles
    const bool tr_da = sizeof(T) == 2 && da && !p.prep && !p.up && !p.in_t2d && !p.bias && !p.noise && p.act == DGE_ACT_NONE &&
                       !dge_get_deterministic() && !(p.dbg & 512);
    dge_note_kernel("conv_igemm<%s,%d,%d,%d,%d,%d,%d,%d>%s%s%s", sizeof(T) == 2 ? "bf16" : "f32", TH, TW, BN, KC, KS, WM, WN,
                    p.in_t2d ? "+t2d" : "", p.prep ? "+prep" : "", ((tr && KS == 3 && TW == 16 && (TH == 16 || TH == 32)) || (tr_da && KS == 3 && TH == 16 && TW == 16 && BN == 64)) ? "+tr" : "");
    DGE_CHECK(!p.prep || (KS == 3 && p.dot_src), "conv: prep is offered for 3x3 data-gradient launches only");
#define DGE_GO(MODE) hipLaunchKernelGGL((conv_igemm_kernel<T, TH, TW, BN, KC, KS, WM, WN, MODE>), dim3((unsigned)grid), dim3(256), 0, s, p)
    if constexpr (KS == 3) {
        if (p.in_t2d) {         // data-gradient launches: always with addend / dot_src stages
            if constexpr (TH == 16 && TW == 16 && BN >= 64) { if (p.prep) DGE_GO(6); else DGE_GO(5); }
            else DGE_CHECK(false, "conv: in_t2d needs N >= 64 and a grid of at least 16 x 16");
        }
        else if (p.prep) DGE_GO(2);
        else if constexpr (sizeof(T) == 2 && TH == 32 && TW == 16 && BN == 64) {
            if (da) DGE_GO(17); else if (tr) DGE_GO(48); else DGE_GO(16);      // (offered without up mode only: single-phase noise tile)
        }
        else if constexpr (sizeof(T) == 2 && TH == 16 && TW == 16 && BN == 64) {
            // Three workgroups per CU (single-phase noise tile + 2-deep weight ring: 52 KB of LDS, 156 VGPRs) for the forward
            // epilogues, where that saves a round of workgroups (LPIPS conv3 / conv4 on the crops: 576 - 768 tiles are two rounds
            // of 512 slots but one of 768; measured 256->256 @48^2 59.9 -> 54.4 us, 128->64 @256^2 89.7 -> 82.0 us).  Grids that
            // fit either way keep the 4-deep ring (512->512 @32^2: 81.9 vs 84.8 us); the data-gradient epilogues need 193
            // registers (spilled at three waves per SIMD: 621 -> 713 us on the layer-15 adjoint) and stay at two.
            const bool w3 = !p.up && !da && !(p.dbg & 256) && (grid + 767) / 768 < (grid + 511) / 512;
            if (w3) { if (tr) DGE_GO(112); else DGE_GO(80); }
            else if (tr_da) DGE_GO(33);
            else if (da) DGE_GO(1);
            else if (tr) DGE_GO(32);
            else DGE_GO(0);
        }
        else if (da || sizeof(T) == 4) DGE_GO(1);
        else if constexpr (sizeof(T) == 2 && TH == 16 && TW == 16) { if (tr) DGE_GO(32); else DGE_GO(0); }
        else DGE_GO(0);
    } else {
        if (da || sizeof(T) == 4) DGE_GO(1);
        else DGE_GO(0);
    }
#undef DGE_GO
    DGE_LAUNCH_CHECK("conv_igemm");
    return 0;
}

// K-chunk (elements): 64 bytes when the channel count allows it, else 32 bytes.
static int kchunk(int cin, int esize) { return cin % (64 / esize) == 0 ? 64 / esize : 32 / esize; }

template <typename T, int KS>
static int launch_t(const ConvParams& p, hipStream_t s) {
    constexpr int E = (int)sizeof(T);
    constexpr int K0 = 64 / E, K1 = 32 / E;
    int bn = dge_conv_ntile(p.Ntot);
    if (bn == 128) {             // 128-wide N tiles only when they still give every CU two workgroups (measured: 223 vs 263 us
                                 // on 256->256 @128^2 B=8, but 117 vs 88 us on 512->512 @64^2 B=2)
        const long blocks128 = (long)((p.H + 15) / 16) * ((p.W + 15) / 16) * p.B * (p.Ntot / 128);
        if (blocks128 < 512) bn = 64;
    }
    if (dge_env().conv_bn > 0) bn = dge_env().conv_bn;
    if (p.up && bn < 64) bn = 64;       // the 32-wide configurations hold a single-phase noise tile
    int kc = kchunk(p.in_s2d ? p.Cin / 4 : p.Cin, E);
    if (dge_env().conv_kc > 0) kc = dge_env().conv_kc;
    const long work = (long)p.B * p.H * p.W * ((p.Ntot + bn - 1) / bn);
    bool small = (p.H <= 8 && p.W <= 8) || work < 256L * 256;
    if (dge_env().conv_small >= 0) small = dge_env().conv_small != 0;
    if (p.in_t2d) small = false;        // (offered on the 16 x 16 tile configurations only)
#define GO(TH, TW, BN, KC, WM, WN) return launch_cfg<T, TH, TW, BN, KC, KS, WM, WN>(p, s)
    if (small) {           // 8x8 pixel tiles, narrower N tiles: more workgroups for the low-resolution layers
        if (bn >= 64) {
            // 256- / 128-byte K chunks when the channel count allows: fewer stages / barriers / halo restages in the serial K
            // loop (measured 512->512 @8^2 B=8: 41 us with 64-byte chunks, 32 us with 128, 27 us with 256)
            if (kc == K0 && (p.in_s2d ? p.Cin / 4 : p.Cin) % (4 * K0) == 0 && !dge_env().conv_nok4) GO(8, 8, 64, 4 * K0, 2, 2);
            if (kc == K0 && (p.in_s2d ? p.Cin / 4 : p.Cin) % (2 * K0) == 0 && !dge_env().conv_nok2) GO(8, 8, 64, 2 * K0, 2, 2);
            if (kc == K0) GO(8, 8, 64, K0, 2, 2);
            GO(8, 8, 64, K1, 2, 2);
        }
        if (kc == K0) GO(8, 16, 32, K0, 4, 1); GO(8, 16, 32, K1, 4, 1);
    }
    if (bn == 128) { if (kc == K0) GO(16, 16, 128, K0, 2, 2); GO(16, 16, 128, K1, 2, 2); }
    if constexpr (E == 2 && KS == 3) {
        // 64-wide N tile on 32 x 16 pixels: each wave holds 128 pixels x 64 channels (6 fragment reads per 8 MFMAs instead of 4 per 4)
        if (bn == 64 && kc == K0 && !p.up && !p.in_t2d && !p.prep && (dge_env().conv_dbg & 128) &&
            (long)((p.H + 31) / 32) * ((p.W + 15) / 16) * p.B * (p.Ntot / 64) >= 1024) GO(32, 16, 64, K0, 4, 1);
    }
    if (bn == 64)  { if (kc == K0) GO(16, 16, 64, K0, 4, 1);  GO(16, 16, 64, K1, 4, 1); }
    if (kc == K0) GO(16, 16, 32, K0, 4, 1); GO(16, 16, 32, K1, 4, 1);
#undef GO
}

