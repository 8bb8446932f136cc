// Streaming form of the StyleGAN2 up layer for the narrow top of the generator (64 -> 32 channels, 512^2 -> 1024^2; bf16), gfx950.
// Reference math: model/stylegan2_generator.py:879-896 (conv_transpose2d stride 2 + 4x4 FIR), :908-921 (demodulation, noise,
// bias, lrelu * sqrt 2) - the same two-factor evaluation as upconv_fir_kernel (upconv_kernels.hip): the (2H+1)^2 transposed-conv
// result t in phase form on the MFMAs (9 tap-MACs per input pixel), then the separable [1,3,3,1]/4 FIR on the VALU.
//
// upconv_fir_kernel ran this layer at 313 us against a 100 us byte floor: a 16x16-pixel tile is one latency chain (weights + halo
// in: 80 us of the launch; K loop: 55 us; FIR + 537 MB of stores: 145 us) and the three phases of a workgroup do not overlap.
// Here the layer is a stream, built from the parts of conv_stream.hip:
//   * a 2-wave team owns a strip of 32 t-pixel columns (= 60 finished output columns: the FIR needs one t column on the left and
//     two on the right, strips advance by 30 input pixels) and marches down the image one INPUT row per step; the input rows live
//     in a team-shared LDS ring filled by LDS-DMA through one-row buffer descriptors (zero padding = out-of-range lanes);
//   * each wave holds 16 of the 32 output channels.  Its MFMA M tile stacks the two horizontal phases of those channels
//     (rows 0-15: t[., 2n], rows 16-31: t[., 2n+1]): 6 (row phase, row shift, column shift) units x 4 K slices = 24 MFMAs per
//     step, the weights of all of them resident in registers with style, demodulation and gain folded in (the reference's own
//     fused-modulation form, :858-864);
//   * the rows of the M tile are permuted so that a lane (pixel n, K half kh) holds both column phases of 8 CONSECUTIVE channels:
//     the pair is packed into one bf16x2 word (t is rounded to bf16 exactly where upconv_fir_kernel rounds it), the neighbour
//     columns come from the adjacent lanes by DPP wave shifts (no LDS round trip), the horizontal FIR is v_dot2_f32_bf16 on whole
//     words, the vertical FIR runs on a three-row register history, and the finished rows leave as 16-byte stores;
//   * noise rows ride the DMA stream through a small ring of their own.
#include <type_traits>
#include <stdlib.h>
#include "common.h"
#include "../../include/dge_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned rsrc_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
typedef __attribute__((ext_vector_type(2))) float f2_t;

__device__ __forceinline__ unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
__device__ __forceinline__ rsrc_t make_rsrc(unsigned long long base, unsigned bytes) {
    rsrc_t r;
    r[0] = rfl((unsigned)base); r[1] = rfl((unsigned)(base >> 32) & 0xffffu); r[2] = rfl(bytes); r[3] = 0x00020000u;
    return r;
}

struct UpsParams {
    const void* x; const void* w; void* y;
    const float* in_scale; const float* out_scale; const float* noise; const float* noise_w; const float* bias;
    int B, H, W;
    int noise_bstride, act;
    float bias_scale, gain;
    int nstrips, nseg, seg_rows, njobs, jobs_per_xcd;
    int dbg;                  // ablation switches for tuning (DGE_UP_DBG: 1 no stores, 2 no MFMAs, 4 no DMA in the loop), 0 in production
};

constexpr int CIN = 64, COUT = 32, KS = CIN / 16;
constexpr int PXB = CIN * 2;                      // bytes per input pixel
constexpr int HALO = 34, RB = HALO * PXB;         // one ring row: 34 pixels (33 used), 4352 B = 4 full 1 KB pieces + 16 lanes
constexpr int NR = 4, D = 2;                      // ring rows, rows in flight ahead of the newest live row
constexpr int XRING = NR * RB;
constexpr int N_OFF = (XRING + 1023) / 1024 * 1024;
constexpr int NZR = 4;                            // noise ring: one 1 KB piece (4 output rows x 64 columns f32) per step and wave
constexpr int DUMMY_OFF = N_OFF + 2 * NZR * 1024;
constexpr int LDS_BYTES = DUMMY_OFF + 1024;
constexpr int LPR = 4;                            // loads per wave and step: 3 x pieces + 1 noise piece

// 16-byte chunk swizzle of the ring image (8 chunks per pixel): conflict-free ds_read_b128 of 16 consecutive pixels
__device__ __forceinline__ int chunk_swz(int px) { return (px >> 1) & 7; }

#define UPS_P(off) "buffer_load_dwordx4 %1, %3, 0 offen offset:" #off " lds\n\t"
#define UPS_M0 "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
// one ring row: pieces 0, 2, 4 by wave 0 (piece 4 = the last 16 lanes' worth, under an EXEC mask, beyond the 12-bit offset field:
// second M0 value + third offset register), pieces 1, 3 + a zero-length piece by wave 1 (both waves count 3 loads per row)
__device__ __forceinline__ void dma_row(int wave, unsigned voff, unsigned voff_b, unsigned voff_c, rsrc_t rs, rsrc_t rs_null,
                                        unsigned m0v, unsigned m0_dummy) {
    unsigned long long keep;
    if (wave == 0)
        asm volatile(UPS_M0 UPS_P(0) UPS_P(2048)
                     "s_mov_b32 m0, %5\n\ts_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xffff\n\tbuffer_load_dwordx4 %4, %3, 0 offen lds\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(rs), "v"(voff_c), "s"(m0v + 4096u) : "memory");
    else
        asm volatile(UPS_M0 UPS_P(1024) UPS_P(3072) "s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %4, 0 offen lds"
                     : "=&s"(keep) : "v"(voff_b), "s"(m0v), "s"(rs), "s"(rs_null), "s"(m0_dummy) : "memory");
}
__device__ __forceinline__ void dma_piece(unsigned voff, rsrc_t rs, unsigned m0v) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(m0v), "s"(rs) : "memory");
}

// the six MFMA units of a step: row phase u, row shift a (input row m - a), column shift b (input column n - b).  The M tile
// holds column phase v = 0 in rows 0-15 (tap (wy, b ? 0 : 2)) and v = 1 in rows 16-31 (tap (wy, 1), only with b = 0);
// wy = u == 0 ? (a ? 0 : 2) : 1.  Index into the packed [9][Cout][Cin] weight (dge_pack_upconv_weight, upconv_kernels.hip):
// phase (0,0): q = 2a + b; (0,1): q = 4 + a; (1,0): q = 6 + b; (1,1): q = 8.
__host__ __device__ constexpr int unit_u(int k) { return k < 4 ? 0 : 1; }
__host__ __device__ constexpr int unit_a(int k) { return k < 4 ? (k >> 1) : 0; }
__host__ __device__ constexpr int unit_b(int k) { return k & 1; }
__host__ __device__ constexpr int unit_q0(int k) { return unit_u(k) == 0 ? 2 * unit_a(k) + unit_b(k) : 6 + unit_b(k); }
__host__ __device__ constexpr int unit_q1(int k) { return unit_b(k) ? -1 : (unit_u(k) == 0 ? 4 + unit_a(k) : 8); }

__global__ __launch_bounds__(128, 2) void upconv_stream_kernel(UpsParams p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = (int)rfl(threadIdx.x >> 6);
    const unsigned lds0 = lds_off(lds);
    const int n31 = lane & 31, kh = lane >> 5;

    // ---- job: the teams of one XCD take a contiguous range of (sample, segment, strip)
    int job = (blockIdx.x & 7) * p.jobs_per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= p.jobs_per_xcd || job >= p.njobs) return;
    const int strip = job % p.nstrips; job /= p.nstrips;
    const int seg = job % p.nseg;
    const int b = job / p.nseg;
    const int r0 = seg * p.seg_rows;                                   // input rows [r0, r1) -> output rows [2 r0, 2 r1)
    const int r1 = min(r0 + p.seg_rows, p.H);
    const int nsteps = r1 - r0 + 2;                                    // t-pixel rows m = r0 - 1 .. r1
    const int OH = 2 * p.H, OW = 2 * p.W;
    const int hx0 = 30 * strip - 2;                                    // input column of halo pixel 0 (t-pixel lane n <-> column hx0 + 1 + n)

    const unsigned xrow_bytes = (unsigned)p.W * PXB;
    const unsigned long long Xb = (unsigned long long)p.x + (unsigned long long)b * p.H * xrow_bytes;
    const unsigned long long NZb = p.noise ? (unsigned long long)(p.noise + (size_t)b * p.noise_bstride) : Xb;
    const unsigned nzrow_bytes = p.noise ? (unsigned)OW * 4 : 0;

    // ---- DMA lane offsets (as conv_stream: 16-byte unit = (halo pixel, chunk), source chunk swizzled; odd pieces start 8 pixels on)
    unsigned voff, voff_b, voff_c;
    {
        const int px = lane >> 3, cs = lane & 7;
        const int gx = hx0 + px;
        voff = (unsigned)(gx * PXB + ((cs ^ chunk_swz(px)) << 4));              // negative / beyond the row = out of range = zeros
        voff_b = (unsigned)(gx * PXB + ((cs ^ chunk_swz(px + 8)) << 4));
        voff_c = voff + 4096;
    }
    // noise piece: 4 output rows x 64 columns from column 60 strip (lane = (row, 4 columns))
    const unsigned nvoff = (unsigned)(((lane >> 4) * OW + 60 * strip + (lane & 15) * 4) * 4);
    const rsrc_t rs_null = make_rsrc(Xb, 0);

    // ring row h <-> input row r0 - 2 + h; step t (t-pixel row m = r0 - 1 + t) reads rows h = t (m - 1) and t + 1 (m)
    unsigned long long xptr = Xb + (unsigned long long)(long long)(r0 - 2) * xrow_bytes;
    auto issue = [&](int h) {
        const int gy = r0 - 2 + h;
        const bool xv = (unsigned)gy < (unsigned)p.H && h <= nsteps;
        const rsrc_t rx = make_rsrc(xptr, xv ? xrow_bytes : 0u);
        xptr += xrow_bytes;
        dma_row(wave, voff, voff_b, voff_c, rx, rs_null, lds0 + (unsigned)(h & (NR - 1)) * RB, lds0 + DUMMY_OFF);
    };
    // noise rows of step t: output rows 2 (r0 - 1 + t) - 2 .. + 3 (the step finishes the first two)
    auto issue_noise = [&](int t) {
        const int oy = 2 * (r0 - 1 + t) - 2;
        const bool v = p.noise != nullptr && oy >= 0 && oy < OH && t < nsteps;
        const rsrc_t rn = make_rsrc(NZb + (unsigned long long)(v ? oy : 0) * nzrow_bytes, v ? (unsigned)(OH - oy) * nzrow_bytes : 0u);
        dma_piece(nvoff, rn, lds0 + N_OFF + (unsigned)(wave * NZR + (t & (NZR - 1))) * 1024);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // rows 0 .. D and the noise of steps 0 .. D - 1 start before the weights are touched (4 loads per wave and "row")
    issue(0);
    for (int h = 1; h <= D; h++) { issue(h); issue_noise(h - 1); }

    // ---- weights -> registers: W'[o][k] = bf16(W[o][k] * in_scale[b][k] * out_scale[b][o] * gain)
    // M-tile row r = 8 j + 4 kk + i holds (v = j >> 1, channel 8 kk + 4 (j & 1) + i) of this wave's 16 channels, so that the
    // accumulator of lane (n, kh) is: registers 0-7 = v 0 of channels 8 kh .. 8 kh + 7, registers 8-15 = v 1 of the same
    uint4 wf[6][KS];
    {
        const bf16_t* __restrict__ Wp = (const bf16_t*)p.w;
        const int j = n31 >> 3, kk = (n31 >> 2) & 1, i = n31 & 3;
        const int v = j >> 1, o = wave * 16 + 8 * kk + 4 * (j & 1) + i;
        const float osc = (p.out_scale ? p.out_scale[b * COUT + o] : 1.f) * p.gain;
        float sc[KS][8];
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
            for (int e = 0; e < 8; e++) sc[ks][e] = (p.in_scale ? p.in_scale[b * CIN + ks * 16 + kh * 8 + e] : 1.f) * osc;
        StaticFor<6>::run([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const int q = v == 0 ? unit_q0(k) : unit_q1(k);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                uint4 wv = make_uint4(0, 0, 0, 0);
                if (q >= 0) wv = *(const uint4*)(Wp + ((size_t)(q * COUT + o) * CIN + ks * 16 + kh * 8));
                float f[8];
                unpack16(wv, f, (bf16_t*)nullptr);
#pragma unroll
                for (int e = 0; e < 8; e++) f[e] *= sc[ks][e];
                wf[k][ks] = pack16(f, (bf16_t*)nullptr);
            }
        });
    }
    // epilogue constants of this lane's 8 channels (gain folded: every supported activation is positively homogeneous)
    const int c0 = wave * 16 + 8 * kh;
    float bia[8];
    {   // two 16-byte loads (eight conditional scalar loads were eight dependent round trips of the job's prologue)
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        if (p.bias) { b0 = *(const float4*)(p.bias + c0); b1 = *(const float4*)(p.bias + c0 + 4); }
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; e++) bia[e] = bv[e] * p.bias_scale * p.gain;
    }
    const float nwv = p.noise ? p.noise_w[0] * p.gain : 0.f;
    const float slope = p.act == DGE_ACT_LRELU ? 0.2f : (p.act == DGE_ACT_RELU ? 0.f : 1.f);

    // ---- B-fragment lane offsets: halo pixel n31 + dx (dx = 1 - b), chunk (2 ks + kh) ^ swizzle
    unsigned loff[2][KS];
#pragma unroll
    for (int dx = 0; dx < 2; dx++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const int px = n31 + dx;
            loff[dx][ks] = lds0 + px * PXB + (((ks * 2 + kh) ^ chunk_swz(px)) << 4);
        }
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) u32x4_t* lds_u4p;
    typedef f2_t f32x2v_t;
    typedef const __attribute__((address_space(3))) f32x2v_t* lds_f2p;
    auto lds_u4 = [&](unsigned a) { const u32x4_t t = *(lds_u4p)a; return make_uint4(t[0], t[1], t[2], t[3]); };

    // output columns of this lane: X_e = 60 strip - 2 + 2 n31 (its own v = 0 column) and X_e + 1; lanes 1 .. 30 finish theirs
    const int Xe = 60 * strip - 2 + 2 * n31;
    const bool colv = n31 >= 1 && n31 <= 30 && Xe < OW && !(p.dbg & 1);
    const unsigned nzoff = lds0 + N_OFF + (unsigned)(wave * NZR) * 1024 + (unsigned)(max(2 * n31 - 2, 0) * 4);
    unsigned char* __restrict__ Yb = (unsigned char*)p.y + (size_t)b * OH * OW * (COUT * 2);
    const size_t yrow_bytes = (size_t)OW * COUT * 2;
    const unsigned ycol = (unsigned)Xe * (COUT * 2) + (unsigned)c0 * 2;

    // FIR history: h rows 2m-3, 2m-2, 2m-1 of this lane's (2 columns x 8 channels)
    f2_t hist[3][2][4];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) hist[r][c][i] = f2_t{0.f, 0.f};
    f2_t bia2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) bia2[i] = f2_t{bia[2 * i], bia[2 * i + 1]};
    f32x16_t zero16;
#pragma unroll
    for (int r = 0; r < 16; r++) zero16[r] = 0.f;
    const unsigned K0 = 0x3e80u, K1 = 0x3f40u;            // bf16 0.25, 0.75
    const unsigned cL = K0 << 16, cC_e = K1 | (K1 << 16), cR_e = K0, cC_o = K0 | (K1 << 16), cR_o = K1 | (K0 << 16);

    for (int t = 0; t < nsteps; t++) {
        // row t + 1 (and the noise of step t) have landed when at most (D - 1) later rows' loads are still in flight
        // (stores count in vmcnt too and only make this wait conservative; an exact allowance that includes them measured the same)
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"((D - 1) * LPR) : "memory");
        __builtin_amdgcn_s_barrier();                                  // the partner's pieces landed; it finished step t - 1
        if (!(p.dbg & 4)) { issue(t + 1 + D); issue_noise(t + D); }

        const unsigned rb_prev = (unsigned)(t & (NR - 1)) * RB, rb_cur = (unsigned)((t + 1) & (NR - 1)) * RB;
        // 16 fragments in consumption order: set 0 = (row m, dx 1) -> units 0, 4 | set 1 = (row m, dx 0) -> units 1, 5 |
        // set 2 = (row m - 1, dx 1) -> unit 2 | set 3 = (row m - 1, dx 0) -> unit 3; PF fragments are requested ahead of their MFMAs
        constexpr int NQ = 16, PF = 6;
        unsigned faddr[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) faddr[q] = loff[(q >> 2) & 1 ? 0 : 1][q & 3] + ((q >> 3) ? rb_prev : rb_cur);
        uint4 bq[PF];
        f32x16_t acc[2];
        acc[0] = zero16; acc[1] = zero16;
        if (!(p.dbg & 2)) {
        StaticFor<PF>::run([&](auto qc) { bq[decltype(qc)::value] = lds_u4(faddr[decltype(qc)::value]); });
        StaticFor<NQ>::run([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int set = q >> 2, ks = q & 3;
            const bf16x8_t bf = *(const bf16x8_t*)&bq[q % PF];
            constexpr int k0u = set == 0 ? 0 : (set == 1 ? 1 : (set == 2 ? 2 : 3));
            if constexpr (q == 0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&wf[k0u][ks], bf, zero16, 0, 0, 0);
            else acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&wf[k0u][ks], bf, acc[0], 0, 0, 0);
            if constexpr (set < 2) {
                if constexpr (q == 0) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&wf[4 + set][ks], bf, zero16, 0, 0, 0);
                else acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&wf[4 + set][ks], bf, acc[1], 0, 0, 0);
            }
            if constexpr (q + PF < NQ) bq[q % PF] = lds_u4(faddr[q + PF]);
        });
        // pin the issue order: PF reads, then per fragment its MFMAs and one read, then the last PF fragments' MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
        StaticFor<NQ - PF>::run([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            __builtin_amdgcn_sched_group_barrier(0x008, q < 8 ? 2 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        }

        // ---- t rows 2m (u = 0) and 2m + 1 (u = 1) -> horizontal FIR -> h rows of this lane's two columns (channel pairs: packed f32 math)
        f2_t hn[2][2][4];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const unsigned wc = pack2bf(acc[u][e], acc[u][8 + e]);                              // (t[X_e], t[X_e + 1])
                const unsigned wl = __builtin_amdgcn_mov_dpp(wc, 0x138, 0xf, 0xf, true) ;   // lane n - 1: (t[X_e - 2], t[X_e - 1])
                const unsigned wr = __builtin_amdgcn_mov_dpp(wc, 0x130, 0xf, 0xf, true) ;   // lane n + 1: (t[X_e + 2], t[X_e + 3])
                float he, ho;
                // (the first product of a chain in the non-accumulating VOP3P form: the accumulating one costs a v_mov 0 each)
                asm("v_dot2_f32_bf16 %0, %1, %2, 0" : "=v"(he) : "v"(wl), "v"(cL));
                he = __builtin_amdgcn_fdot2_f32_bf16(*(const bf2_t*)&wc, *(const bf2_t*)&cC_e, he, false);
                he = __builtin_amdgcn_fdot2_f32_bf16(*(const bf2_t*)&wr, *(const bf2_t*)&cR_e, he, false);
                asm("v_dot2_f32_bf16 %0, %1, %2, 0" : "=v"(ho) : "v"(wc), "v"(cC_o));
                ho = __builtin_amdgcn_fdot2_f32_bf16(*(const bf2_t*)&wr, *(const bf2_t*)&cR_o, ho, false);
                hn[u][0][e >> 1][e & 1] = he;
                hn[u][1][e >> 1][e & 1] = ho;
            }

        // ---- vertical FIR + tail: output rows 2m - 2 (h rows 2m-3 .. 2m) and 2m - 1 (h rows 2m-2 .. 2m+1)
        const int oy0 = 2 * (r0 - 1 + t) - 2;
        if (t >= 2) {
            const unsigned nzb = nzoff + (unsigned)(t & (NZR - 1)) * 1024;
#pragma unroll
            for (int yy = 0; yy < 2; yy++) {
                const f32x2v_t nz = *(lds_f2p)(nzb + yy * 256);
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const float nzv = (c == 0 ? nz[0] : nz[1]) * nwv;
                    const f2_t nz2 = {nzv, nzv};
                    unsigned ow[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const f2_t f = yy == 0 ? (hist[0][c][i] + hn[0][c][i]) * 0.25f + (hist[1][c][i] + hist[2][c][i]) * 0.75f
                                               : (hist[1][c][i] + hn[1][c][i]) * 0.25f + (hist[2][c][i] + hn[0][c][i]) * 0.75f;
                        const f2_t uu = f + (nz2 + bia2[i]);
                        const f2_t lo = uu * slope;
                        ow[i] = pack2bf(fmaxf(uu[0], lo[0]), fmaxf(uu[1], lo[1]));
                    }
                    if (colv) *(uint4*)(Yb + (size_t)(oy0 + yy) * yrow_bytes + ycol + c * (COUT * 2)) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) { hist[0][c][i] = hist[2][c][i]; hist[1][c][i] = hn[0][c][i]; hist[2][c][i] = hn[1][c][i]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no DMA may land after the wave has given its LDS back
}

}  // namespace

// eligibility + launch (called from dge_upconv_fir, upconv_kernels.hip)
bool dge_upconv_stream_ok(int B, int H, int W, int Cin, int Cout, int dtype) {
    static int off = -1;
    if (off < 0) off = getenv("DGE_NO_UPSTREAM") ? 1 : 0;
    if (off || dtype != DGE_BF16 || Cin != CIN || Cout != COUT) return false;
    if (W % 2 != 0 || W < 64 || H < 32) return false;                      // (16-byte noise pieces: 60 strip * 4 B is 16-byte aligned, OW * 4 B needs W even)
    if ((long)W * PXB >= (1L << 31) || (long)H * W * 4 >= (1L << 31)) return false;
    return (long)B * H * W >= (1L << 16) || dge_env().force_stream;
}

int dge_upconv_stream_launch(const void* x, const void* w_packed, void* y, const float* in_scale, const float* out_scale, const float* noise,
                             int noise_bstride, const float* noise_w, const float* bias, float bias_scale, float gain, int act,
                             int B, int H, int W, hipStream_t s) {
    UpsParams p;
    p.dbg = dge_env().up_dbg;
    p.x = x; p.w = w_packed; p.y = y; p.in_scale = in_scale; p.out_scale = out_scale; p.noise = noise; p.noise_w = noise_w; p.bias = bias;
    p.B = B; p.H = H; p.W = W; p.noise_bstride = noise_bstride; p.act = act; p.bias_scale = bias_scale; p.gain = gain;
    static int caps[16] = {0};                                   // resident teams, per device (the attribute is per device too)
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& cap = caps[dev >= 0 && dev < 16 ? dev : 0];
    if (!cap) {
        (void)hipFuncSetAttribute((const void*)upconv_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        int occ = 0, ncu = 256;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)upconv_stream_kernel, 128, LDS_BYTES) != hipSuccess || occ < 1) occ = 1;
        cap = occ * ncu;
    }
    p.nstrips = (2 * W + 59) / 60;
    int nseg = cap / (B * p.nstrips);
    int maxseg = H / 16; if (maxseg < 1) maxseg = 1;
    if (nseg > maxseg) nseg = maxseg;
    if (nseg < 1) nseg = 1;
    p.seg_rows = (H + nseg - 1) / nseg;
    p.nseg = (H + p.seg_rows - 1) / p.seg_rows;
    p.njobs = B * p.nstrips * p.nseg;
    p.jobs_per_xcd = (p.njobs + 7) / 8;
    dge_note_kernel("upconv_stream<bf16,%d,%d>", CIN, COUT);
    hipLaunchKernelGGL(upconv_stream_kernel, dim3((unsigned)(p.jobs_per_xcd * 8)), dim3(128), LDS_BYTES, s, p);
    DGE_LAUNCH_CHECK("upconv_stream");
    return 0;
}
