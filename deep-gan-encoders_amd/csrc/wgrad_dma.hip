// Conv weight gradient, bf16, 3x3, streaming form for gfx950 (reference model/E/E.py:50-85 differentiated):
//   dW[o][i][tap] += sum_{b,p} g[b,p,o] * Xn[b,p+tap,i],   Xn = X*sc[b,i] + sh[b,i] inside the image, 0 outside.
//
// conv_wgrad_tr_kernel (enc_bwd_kernels.hip) staged every pixel tile global -> registers -> affine -> LDS with one tile of
// prefetch: the MFMA phase of a tile (0.5 us) is shorter than the loaded memory latency, so each workgroup spent ~4 us per
// tile for ~1 us of matrix work (MFMA busy 0.23).  This kernel keeps the same GEMM view (M = 32 output channels, N = 32 input
// channels, K = the pixels of a TH x 16 tile, all 9 taps per wave, transposing LDS reads) and changes how tiles arrive:
//
//   * both tiles come by LDS-DMA (buffer_load_dwordx4 ... offen lds) into a ring of NS stages, NS-1 tiles ahead of the MFMAs;
//     lanes that fall outside the image (zero padding, ragged tiles, channel tails) carry an out-of-range buffer offset and
//     the hardware writes zeros for them (pinned by tools/probes/probe_ldsdma.hip).  One barrier per tile;
//   * the instance-norm affine leaves the fill path.  The K split is aligned to the samples (a workgroup's tiles belong to ONE
//     sample b), so the SCALE sc[b,i] multiplies the staged sums of the flush, and the SHIFT contributes
//     sh[b,i] * G[b,o,tap], G = sum of g over the pixels whose tap neighbour lies inside the image
//     = total - excluded border row - excluded border column + corner: the row / column / corner sums are picked out of the
//     g fragments that are in registers anyway (v_dot2 in the shadow of the MFMAs).  No VALU instruction touches the 144
//     accumulators: <= 256 registers, two workgroups per CU;
//   * no rounding of the affine result to bf16: x enters the MFMA as stored;
//   * XCD-aware placement: the (o, i) tiles of one pixel range read 64-byte slices of the same lines and run on one XCD.
// Measured at batch 8 (tools/perf_wgrad_all.sh, old -> new): 16->16 @1024^2 239 -> 167 us, 16->32 278 -> 191, 32->64 @512^2
// 147 -> 123, 64->64 @256^2 100 -> 63, 64->128 173 -> 101, 128->128 @128^2 84 -> 62, 128->256 130 -> 100, 256->256 @64^2
// 73 -> 63, 256->512 131 -> 119.  The deep layers (<= 32^2: few tiles per sample) stay on conv_wgrad_tr_kernel.
//
// 16-channel tensors use a 32-byte pixel pitch in LDS (half the DMA instructions); the upper 16 fragment rows then alias the
// lower ones and land in accumulator rows / columns that the flush discards.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/dge_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned rsrc_t;
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s_t* lds_v4s_ptr;

__device__ __forceinline__ unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
// raw buffer descriptor (gfx9 layout): base, stride 0, num_records in bytes, DATA_FORMAT = 32
__device__ __forceinline__ rsrc_t make_rsrc(unsigned long long base, unsigned bytes) {
    rsrc_t r;
    r[0] = rfl((unsigned)base); r[1] = rfl((unsigned)(base >> 32) & 0xffffu); r[2] = rfl(bytes); r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void dma16(unsigned voff, rsrc_t rs, unsigned m0v) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(m0v), "s"(rs) : "memory");
}
constexpr unsigned DEAD = 0x7fffffffu;          // + any in-sample byte offset (< 2^30) stays beyond num_records (< 2^30)

template <int TH_, int GP_, int XP_, int NW_, int NS_>
struct WCfg {
    static constexpr int TH = TH_, GP = GP_, XP = XP_, NW = NW_, NS = NS_;
    static constexpr int TW = 16, HH = TH + 2, HWD = TW + 2;
    static constexpr int GBYTES = TH * TW * GP;                                        // whole 1 KB pieces
    static constexpr int XBYTES = ((HH * HWD + 4) * XP + 1023) / 1024 * 1024;         // +4 pixels: the 12-pixel window over-reads
    static constexpr int NGP = GBYTES / 1024, NXP = XBYTES / 1024, NP = NGP + NXP;   // 1 KB pieces of a tile: g first, then x
    static constexpr int PW = (NP + NW - 1) / NW;               // DMA instructions per wave and tile (waves >= NP % NW issue PW - 1)
    static constexpr int STAGE = GBYTES + XBYTES;
    static constexpr int RING = NS * STAGE;
    static constexpr int FLUSH = (NW * 1024 + 1024 * 9) * 4;
    static constexpr int MAIN = RING > FLUSH ? RING : FLUSH;
    static constexpr int GTAB_OFF = MAIN, TAB_OFF = GTAB_OFF + 9 * 32 * 4;
    static constexpr int RW = TH / NW;                                                 // tile rows per wave
    static_assert(TH % NW == 0 && (NS - 2) * PW <= 63 && GBYTES % 1024 == 0 && NS >= 2 && NS <= 5, "configuration");
    static int lds_bytes() { return TAB_OFF + 64 * 4; }
};

#ifdef DGE_WG_TIMING          // tuning builds: clock stamps of one workgroup's tile loop (tools/perf_wgrad_timing.py)
__device__ long long wg_tlog[64 * 6];
#define DGE_WT(k) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); \
                       if (tlogon && t - t_begin >= 8 && t - t_begin < 40) wg_tlog[(t - t_begin - 8) * 6 + k] = t_; } while (0)
#else
#define DGE_WT(k)
#endif

template <class C>
__global__ __launch_bounds__(C::NW * 64, 2) void wgrad_dma_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ X,
                                                                const float* __restrict__ sc, const float* __restrict__ sh,
                                                                float* __restrict__ dW, int B, int H, int W, int Co, int Ci,
                                                                int tiles_x, int tiles_y, int per_group, int gps,
                                                                const float* __restrict__ Wdot, float* __restrict__ dots, int dots_slots) {
    constexpr int TH = C::TH, GP = C::GP, XP = C::XP, NW = C::NW, NS = C::NS, NT = NW * 64;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    float* gtab = (float*)(lds + C::GTAB_OFF);          // [9 kinds][32 o]: total, top row, bottom row, left col, right col, 4 corners
    float* sctab = (float*)(lds + C::TAB_OFF);          // [32] scale of this workgroup's sample and input channels
    float* ktab = sctab + 32;                           // [32] shift
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_it = (Ci + 31) / 32;
    // XCD-aware placement: workgroup L of the dispatch order runs on XCD L % 8 (private L2 each).  The (o, i) tiles of one
    // pixel range read the same g / x lines (64-byte slices of them), so they are placed on ONE XCD: spread over the XCDs in
    // dispatch order every line crossed the fabric once per XCD (measured: the DMA stream alone at 2.2 TB/s on 64 -> 64 @ 256^2).
    // Group g runs on XCD g % 8; the launcher pads the group count (gridDim.y) to a multiple of 8, padding groups return at once.
    const int noi = gridDim.x, lin = blockIdx.x + noi * blockIdx.y;
    int grp_id = blockIdx.y, oi = blockIdx.x;             // fewer than 8 groups: dispatch order (every XCD reads everything anyway)
    if (!(gridDim.y & 7)) { const int slot = lin >> 3; grp_id = (slot / noi) * 8 + (lin & 7); oi = slot % noi; }
    const int o0 = (oi / n_it) * 32, i0 = (oi % n_it) * 32;
    const int tps = tiles_x * tiles_y;
    // group = (sample, split): `gps` workgroups share a sample, `per_group` tiles each
    if (grp_id >= B * gps) return;
    const int smp = grp_id / gps, part = grp_id - smp * gps;
    const int t_begin = smp * tps + part * per_group;
    const int t_end = t_begin + per_group < (smp + 1) * tps ? t_begin + per_group : (smp + 1) * tps;
    if (t_begin >= t_end) return;
    const bool aff = sc != nullptr;
    if (aff && tid < 32) {
        const int ch = i0 + tid;
        sctab[tid] = ch < Ci ? sc[smp * Ci + ch] : 1.f;
        ktab[tid] = ch < Ci ? sh[smp * Ci + ch] : 0.f;
    }
    __syncthreads();
    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    // ---- per-lane DMA geometry: piece k of a tile region = 64 lanes x 16 B = 1 KB of LDS; lane -> (pixel, 16-byte chunk).
    //      Wave w moves g pieces w, w + NW, ... (PG of them: NGP is a multiple of NW) and x halo pieces w, w + NW, ...
    //      (PX or PX - 1): which region an instruction serves is known at compile time, only the last x piece is conditional.
    constexpr int CG = GP / 16, CX = XP / 16, PPG = 1024 / GP, PPX = 1024 / XP;
    constexpr int PG = C::NGP / NW, PX = (C::NXP + NW - 1) / NW;
    static_assert(C::NGP % NW == 0 && PG + PX == C::PW, "piece split");
    const int cg = lane % CG, pg = lane / CG, cx = lane % CX, px = lane / CX;
    const bool g_alive = o0 + cg * 8 < Co, x_alive = i0 + cx * 8 < Ci;
    const bool last_x = wave + NW * (PX - 1) < C::NXP;          // this wave moves PX x pieces (else PX - 1); wave-uniform
    const int pw_mine = PG + PX - (last_x ? 0 : 1);
    unsigned rel_g[PG], rel_x[PX];
#pragma unroll
    for (int j = 0; j < PG; j++) {
        const int q = (wave + NW * j) * PPG + pg;
        rel_g[j] = g_alive ? (unsigned)((((q >> 4) * W + (q & 15)) * Co + o0 + cg * 8) * 2) : DEAD;
    }
#pragma unroll
    for (int j = 0; j < PX; j++) {
        const int q = (wave + NW * j) * PPX + px, hy = q / C::HWD, hx = q - hy * C::HWD;
        rel_x[j] = (q < C::HH * C::HWD && x_alive) ? (unsigned)((((hy - 1) * W + (hx - 1)) * Ci + i0 + cx * 8) * 2) : DEAD;
    }
    const unsigned lds0 = lds_off(lds);
    const unsigned sample_g = (unsigned)H * W * Co * 2, sample_x = (unsigned)H * W * Ci * 2;
    // issue side of the tile sequence
    int irem = t_begin - smp * tps, ity = irem / tiles_x, itx = irem - ity * tiles_x;
    auto issue = [&](int stage) {
        const unsigned mg = lds0 + stage * C::STAGE + wave * 1024, mx = mg + C::GBYTES;
        const int y0 = ity * TH, x0 = itx * 16;
        const rsrc_t rg = make_rsrc((unsigned long long)g + (unsigned long long)smp * sample_g, sample_g);
        const rsrc_t rx = make_rsrc((unsigned long long)X + (unsigned long long)smp * sample_x, sample_x);
        const unsigned gbase = (unsigned)((y0 * W + x0) * Co * 2), xbase = (unsigned)((y0 * W + x0) * Ci * 2);
        const bool interior = (y0 >= 1) & (y0 + TH + 1 <= H) & (x0 >= 1) & (x0 + 17 <= W);
        if (interior) {
#pragma unroll
            for (int j = 0; j < PG; j++) dma16(gbase + rel_g[j], rg, mg + j * NW * 1024);
#pragma unroll
            for (int j = 0; j < PX - 1; j++) dma16(xbase + rel_x[j], rx, mx + j * NW * 1024);
            if (last_x) dma16(xbase + rel_x[PX - 1], rx, mx + (PX - 1) * NW * 1024);
        } else {
#pragma unroll
            for (int j = 0; j < PG; j++) {
                const int q = (wave + NW * j) * PPG + pg;
                const bool ok = (rel_g[j] != DEAD) & (y0 + (q >> 4) < H) & (x0 + (q & 15) < W);
                dma16(ok ? gbase + rel_g[j] : DEAD, rg, mg + j * NW * 1024);
            }
#pragma unroll
            for (int j = 0; j < PX; j++) {
                const int q = (wave + NW * j) * PPX + px, hy = q / C::HWD, hx = q - hy * C::HWD;
                const bool ok = (rel_x[j] != DEAD) & ((unsigned)(y0 + hy - 1) < (unsigned)H) & ((unsigned)(x0 + hx - 1) < (unsigned)W);
                if (j < PX - 1 || last_x) dma16(ok ? xbase + rel_x[j] : DEAD, rx, mx + j * NW * 1024);
            }
        }
        if (++itx == tiles_x) { itx = 0; ++ity; }
    };
    // counted wait: this wave's pieces of the tile it is about to read have landed while `nfly` younger tiles stay in flight
    auto wait_tile = [&](int nfly) {
        const bool full = pw_mine == C::PW;
#define DGE_W(N) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory")
        if (nfly <= 0) DGE_W(0);
        else if (nfly == 1) { if (full) DGE_W(C::PW); else DGE_W(C::PW - 1); }
        else if (nfly == 2) { if (full) DGE_W(2 * C::PW > 63 ? 63 : 2 * C::PW); else DGE_W(2 * (C::PW - 1) > 63 ? 63 : 2 * (C::PW - 1)); }
        else { if (full) DGE_W(3 * C::PW > 63 ? 63 : 3 * C::PW); else DGE_W(3 * (C::PW - 1) > 63 ? 63 : 3 * (C::PW - 1)); }
#undef DGE_W
    };

    // ---- fragment addressing of the transposing reads (ds_read_b64_tr_b16: in each 16-lane group lane i points at pixel
    //      (i>>2), channels 4*(i&3)..+3 of a [4 pixel][16 channel] block and receives channel i of the 4 pixels)
    const int grp = lane >> 4, li = lane & 15, kg = lane >> 5;
    const int frag_g = (kg * 8 + (li >> 2)) * GP + (GP == 64 ? (grp & 1) * 32 : 0) + (li & 3) * 8;
    const int frag_x = (kg * 8 + (li >> 2)) * XP + (XP == 64 ? (grp & 1) * 32 : 0) + (li & 3) * 8;
    auto trd = [&](const unsigned char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)p); };

    // shift bookkeeping of the running sample (per lane: output channel o0 + (lane & 31), pixel half kg, this wave's rows)
    float s_tot = 0.f, s_top = 0.f, s_bot = 0.f, s_lft = 0.f, s_rgt = 0.f, s_tl = 0.f, s_tr = 0.f, s_bl = 0.f, s_br = 0.f;
    // compute side of the tile sequence
    int crem = t_begin - smp * tps, cty = crem / tiles_x, ctx = crem - cty * tiles_x;

    // the sample's border / total sums of g after the last tile: the pixel halves are combined by a lane exchange, the waves in
    // wave order through the (now free) ring memory - a fixed summation order, whatever the mode
    auto gather_sums = [&]() {
        float v[9] = {s_tot, s_top, s_bot, s_lft, s_rgt, s_tl, s_tr, s_bl, s_br};
        float* part = (float*)lds;                         // [NW][9][32]
        __syncthreads();                                   // every wave is done with the last stage
#pragma unroll
        for (int k = 0; k < 9; k++) {
            v[k] += __shfl_xor(v[k], 32, 64);
            if (lane < 32) part[(wave * 9 + k) * 32 + lane] = v[k];
        }
        __syncthreads();
        for (int idx = tid; idx < 9 * 32; idx += NT) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) t += part[w * 9 * 32 + idx];
            gtab[idx] = t;
        }
        __syncthreads();
    };

#pragma unroll
    for (int s = 0; s < NS - 1; s++) if (t_begin + s < t_end) issue(s);
    int stage = 0;
#ifdef DGE_WG_TIMING
    const bool tlogon = lin == 40 && tid == 0;
#endif
    for (int t = t_begin; t < t_end; t++) {
        const int younger = t_end - 1 - t;                 // tiles issued after tile t that may stay in flight
        DGE_WT(0);
        wait_tile(younger < NS - 2 ? younger : NS - 2);    // this wave's pieces of tile t have landed
        DGE_WT(1);
        __syncthreads();                                   // ... everybody's have, and everybody is done with tile t-1
        DGE_WT(2);
        if (t + NS - 1 < t_end) issue(stage == 0 ? NS - 1 : stage - 1);
        DGE_WT(3);
        const unsigned char* sg = lds + stage * C::STAGE;
        const unsigned char* sx = sg + C::GBYTES;
        const int y0 = cty * TH, x0 = ctx * 16;
        const bool edge_tile = (y0 == 0) | (y0 + TH >= H) | (x0 == 0) | (x0 + 16 >= W);
        // ---- MFMA: wave w owns tile rows RW*w .. RW*w + RW-1; fragments of row rr+1 are requested before the MFMAs of row rr
        uint4 av[2];
        uint4 xv[2][3];
        uint32_t xw[2][3];
        auto frags = [&](int rr, uint4& a, uint4 (&x3)[3], uint32_t (&w3)[3]) {
            const int row = C::RW * wave + rr;
            const unsigned char* ga = sg + row * 16 * GP + frag_g;
            const v4s_t a0 = trd(ga), a1 = trd(ga + 4 * GP);
            const uint2 a01 = *(const uint2*)&a0, a23 = *(const uint2*)&a1;
            a = make_uint4(a01.x, a01.y, a23.x, a23.y);
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const unsigned char* xa = sx + (row + dy) * C::HWD * XP + frag_x;
                const v4s_t q0 = trd(xa), q1 = trd(xa + 4 * XP), q2 = trd(xa + 8 * XP);
                const uint2 w01 = *(const uint2*)&q0, w23 = *(const uint2*)&q1;
                x3[dy] = make_uint4(w01.x, w01.y, w23.x, w23.y);
                w3[dy] = (*(const uint2*)&q2).x;
            }
        };
        frags(0, av[0], xv[0], xw[0]);
        StaticFor<C::RW>::run([&](auto rc) {
            constexpr int rr = decltype(rc)::value;
            if constexpr (rr + 1 < C::RW) frags(rr + 1, av[(rr + 1) & 1], xv[(rr + 1) & 1], xw[(rr + 1) & 1]);
            const uint4 a4 = av[rr & 1];
            const bf16x8_t a = *(const bf16x8_t*)&a4;
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const uint4 v0 = xv[rr & 1][dy];
                const uint32_t w4 = xw[rr & 1][dy];
                const uint4 b1 = make_uint4(__builtin_amdgcn_alignbit(v0.y, v0.x, 16), __builtin_amdgcn_alignbit(v0.z, v0.y, 16),
                                            __builtin_amdgcn_alignbit(v0.w, v0.z, 16), __builtin_amdgcn_alignbit(w4, v0.w, 16));
                const uint4 b2 = make_uint4(v0.y, v0.z, v0.w, w4);
                acc[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, *(const bf16x8_t*)&v0, acc[dy * 3 + 0], 0, 0, 0);
                acc[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, *(const bf16x8_t*)&b1, acc[dy * 3 + 1], 0, 0, 0);
                acc[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, *(const bf16x8_t*)&b2, acc[dy * 3 + 2], 0, 0, 0);
            }
            if (aff) {
                // sums of this row's g fragment (8 pixels of channel lane & 31; pixels outside the image are zeros)
                const bf16x2_t one2 = {(__bf16)1.0f, (__bf16)1.0f};
                float rs = __builtin_amdgcn_fdot2_f32_bf16(*(const bf16x2_t*)&a4.x, one2, 0.f, false);
                rs = __builtin_amdgcn_fdot2_f32_bf16(*(const bf16x2_t*)&a4.y, one2, rs, false);
                rs = __builtin_amdgcn_fdot2_f32_bf16(*(const bf16x2_t*)&a4.z, one2, rs, false);
                rs = __builtin_amdgcn_fdot2_f32_bf16(*(const bf16x2_t*)&a4.w, one2, rs, false);
                s_tot += rs;
                if (edge_tile) {                                    // wave-uniform: the tile touches the image border
                    const int gy = y0 + C::RW * wave + rr;
                    const float ftop = gy == 0 ? 1.f : 0.f, fbot = gy == H - 1 ? 1.f : 0.f;
                    const float fl = (x0 == 0 && kg == 0) ? __uint_as_float(a4.x << 16) : 0.f;
                    const int pc = W - 1 - x0;                      // tile column of the image's last column
                    float fr = 0.f;
                    if ((unsigned)pc < 16u) {
                        const int ws = (pc & 7) >> 1;
                        const uint32_t wv = ws == 0 ? a4.x : (ws == 1 ? a4.y : (ws == 2 ? a4.z : a4.w));
                        fr = (pc >> 3) == kg ? __uint_as_float((pc & 1) ? (wv & 0xffff0000u) : (wv << 16)) : 0.f;
                    }
                    s_top = fmaf(ftop, rs, s_top); s_bot = fmaf(fbot, rs, s_bot);
                    s_lft += fl; s_rgt += fr;
                    s_tl = fmaf(ftop, fl, s_tl); s_tr = fmaf(ftop, fr, s_tr);
                    s_bl = fmaf(fbot, fl, s_bl); s_br = fmaf(fbot, fr, s_br);
                }
            }
        });
#ifdef DGE_WG_TIMING
        asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[8][15]));
#endif
        DGE_WT(4);
        if (++ctx == tiles_x) { ctx = 0; ++cty; }
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
    if (aff) gather_sums();
    // ---- flush: the waves' partial sums are combined in LDS and staged as [o][i][tap] (the layout of dW): consecutive lanes
    //      add to consecutive addresses (see wgrad_flush in enc_bwd_kernels.hip)
    float* red = (float*)lds;
    float* stg = red + NW * 1024;
    float d_s2 = 0.f, d_s1 = 0.f;          // (every element a thread stages has the same input channel il = tid & 31)
    static_assert(NT % 64 == 0, "flush: a thread's elements share the input channel");
#pragma unroll
    for (int t = 0; t < 9; t++) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave * 1024 + r * 64 + lane] = acc[t][r];
        __syncthreads();
        for (int e = tid; e < 1024; e += NT) {
            const int r = e >> 6, ln = e & 63;
            const int ol = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), il = ln & 31;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) s += red[w * 1024 + e];
            if (aff) {                 // the sample's affine on the staged sum: scale, and the shift times G[o][tap] (header)
                const int dy = t / 3, dx = t - dy * 3;
                float G = gtab[ol];
                if (dy != 1) G -= gtab[(dy == 0 ? 32 : 64) + ol];
                if (dx != 1) G -= gtab[(dx == 0 ? 96 : 128) + ol];
                if (dy != 1 && dx != 1) G += gtab[160 + 32 * ((dy >> 1) * 2 + (dx >> 1)) + ol];
                if (dots) {
                    // the two sums the instance-norm backward of the layer's INPUT needs (model/E/E.py:51-53 differentiated), taken
                    // here instead of in the data-gradient epilogue: with g_x = conv^T(g, w) (zero padding), sum_p g_x[p,i]*X[p,i] =
                    // sum_{o,tap} w[o,i,tap] * (raw correlation of this flush) and sum_p g_x[p,i] = sum_{o,tap} w[o,i,tap] * G[o,tap]
                    // - known BEFORE the data gradient runs, so that its epilogue can apply the backward itself.  w as the data
                    // gradient reads it: rounded to bf16.
                    const float wv = (o0 + ol < Co && i0 + il < Ci) ? bf2f(f2bf(Wdot[((size_t)(o0 + ol) * Ci + i0 + il) * 9 + t])) : 0.f;
                    d_s2 = fmaf(wv, s, d_s2); d_s1 = fmaf(wv, G, d_s1);
                }
                s = fmaf(s, sctab[il], ktab[il] * G);
            }
            stg[(ol * 32 + il) * 9 + t] = s;
        }
    }
    __syncthreads();
    if (dots) {                            // NT / 32 partial pairs per input channel -> one pair of atomics per channel and workgroup
        red[tid * 2] = d_s2; red[tid * 2 + 1] = d_s1;
        __syncthreads();
        if (tid < 64) {
            const int il = tid & 31, k = tid >> 5;
            float v = 0.f;
            for (int j = 0; j < NT / 32; j++) v += red[(il + 32 * j) * 2 + k];
            if (i0 + il < Ci) atomicAdd(dots + ((size_t)(grp_id % dots_slots) * B * Ci + (size_t)smp * Ci + i0 + il) * 2 + k, v);
        }
    }
    const int ni = (Ci - i0 < 32 ? Ci - i0 : 32) * 9;      // valid floats of one o row of this tile (contiguous in dW)
    if (det_on()) {
        const int L = 1024 * 9, nslots = B * gps;
        float* dslot = det_slot(oi, noi, grp_id, nslots, L);
        for (int idx = tid; idx < L; idx += NT) dslot[idx] = stg[idx];
        if (det_arrive_wg(oi, nslots)) {
            for (int idx = tid; idx < L; idx += NT) {
                const int ol = idx / (32 * 9), j = idx - ol * (32 * 9);
                if (o0 + ol < Co && j < ni) dW[((size_t)(o0 + ol) * Ci + i0) * 9 + j] += det_sum(oi, nslots, L, idx);
            }
        }
        return;
    }
    for (int idx = tid; idx < 1024 * 9; idx += NT) {
        const int ol = idx / (32 * 9), j = idx - ol * (32 * 9);
        if (o0 + ol < Co && j < ni) atomicAdd(dW + ((size_t)(o0 + ol) * Ci + i0) * 9 + j, stg[idx]);
    }
}

template <class C>
int launch(const void* g, const void* x, const float* sc, const float* sh, float* dw, int B, int H, int W, int cout, int cin,
           const float* wdot, float* dots, int dots_slots, hipStream_t s) {
    const int tx = (W + 15) / 16, ty = (H + C::TH - 1) / C::TH;
    const int tps = tx * ty;
    const int noi = ((cout + 31) / 32) * ((cin + 31) / 32);
    const int lds_bytes = C::lds_bytes();
    static int attr_bytes[16] = {0};          // dynamic LDS limit already granted to this instantiation, per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto kern = wgrad_dma_kernel<C>;
    if (dev < 0 || dev >= 16 || attr_bytes[dev] < lds_bytes) {
        const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) {
            dge_set_error("wgrad_dma: cannot raise the dynamic LDS limit to %d bytes: %s", lds_bytes, hipGetErrorString(e));
            return -3;
        }
        if (dev >= 0 && dev < 16) attr_bytes[dev] = lds_bytes;
    }
    const int wg_per_cu = 2 * lds_bytes <= 160 * 1024 ? 2 : 1;        // registers allow two 4-wave workgroups per CU
    const int groups_env = dge_env().wgrad_groups;
    // K split: `gps` workgroups per sample and (o, i) tile, one resident set of workgroups where the problem allows it
    int gps = groups_env > 0 ? groups_env : (256 * wg_per_cu + noi * B - 1) / (noi * B);
    gps = gps < 1 ? 1 : (gps > tps ? tps : gps);
    const int per = (tps + gps - 1) / gps;
    gps = (tps + per - 1) / per;
    const int groups = B * gps;
    const int groups8 = groups < 8 ? groups : (groups + 7) / 8 * 8;   // padding groups return at once (XCD-aware placement, see the kernel)
    // deterministic mode: (o, i) tile domains x (sample, group) slots x 32 x 32 x 9 floats; too large -> the caller's other kernel
    if (!dge_det_fits(noi, (long long)B * gps, 1024 * 9)) return 1;
    dge_note_kernel("wgrad_dma<%d,%d,%d,%d>", C::TH, C::GP, C::XP, C::NS);
    hipLaunchKernelGGL(kern, dim3(noi, groups8), dim3(C::NW * 64), lds_bytes, s, (const bf16_t*)g, (const bf16_t*)x, sc, sh, dw, B, H, W,
                       cout, cin, tx, ty, per, gps, wdot, dots, dots_slots < 1 ? 1 : dots_slots);
    DGE_LAUNCH_CHECK("wgrad_dma");
    return 0;
}

}  // namespace

#ifdef DGE_WG_TIMING
extern "C" int dge_wgrad_tlog(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wg_tlog), (size_t)n * sizeof(long long), 0, hipMemcpyDeviceToHost);
}
#endif
// bf16 3x3 weight gradient on the streaming kernel; returns 1 when the shape is not covered (the caller falls back to
// conv_wgrad_tr_kernel), 0 on success, < 0 on error.
// dots (optional, with wdot = the layer's weight [cout][cin][3][3] f32): [dots_slots][B][cin][2], pre-zeroed, += (sum g_x*x, sum g_x)
// of the data gradient g_x of the same (g, w) - see the flush.
int dge_wgrad_dma_try(const void* g, const void* x, const float* sc, const float* sh, float* dw, int B, int H, int W, int cout, int cin,
                      const float* wdot, float* dots, int dots_slots, hipStream_t s) {
    static int off = -1;
    if (off < 0) off = getenv("DGE_NO_WGRAD_DMA") ? 1 : 0;
    if (off) return 1;
    if (cout % 8 || cin % 8) return 1;
    if (dots && (!sc || !wdot || !dge_det_fits(1LL << 40, 1, 1))) return 1;     // (deterministic mode: the data gradient keeps producing the sums)
    if ((size_t)H * W * cout * 2 >= (1u << 30) || (size_t)H * W * cin * 2 >= (1u << 30)) return 1;      // 32-bit buffer offsets + DEAD
    const int tps = ((W + 15) / 16) * ((H + 15) / 16);
    const int noi = ((cout + 31) / 32) * ((cin + 31) / 32);
    // the sample-aligned K split needs enough tiles per sample to amortise a flush per (sample, (o, i) tile): 64^2 and up, and not
    // more workgroups than four resident sets (512 -> 512 @ 32^2: 131 us here against 72 us on conv_wgrad_tr_kernel)
    if (tps < 16 || noi * B > 1024) return 1;
    const bool g32 = cout <= 16, x32 = cin <= 16;
    {   // tuning builds (tools/perf_wgrad.py): alternative tile heights / ring depths for the 16- / 32-channel layers
        static const int cfg = getenv("DGE_WGRAD_CFG") ? atoi(getenv("DGE_WGRAD_CFG")) : 0;
#define WTRY(...) return launch<WCfg<__VA_ARGS__>>(g, x, sc, sh, dw, B, H, W, cout, cin, wdot, dots, dots_slots, s)
        if (cfg == 1) { if (g32 && x32) WTRY(8, 32, 32, 4, 5); if (!g32 && x32) WTRY(8, 64, 32, 4, 4); if (!g32 && !x32) WTRY(8, 64, 64, 4, 4); }
        if (cfg == 2) { if (g32 && x32) WTRY(16, 32, 32, 4, 4); if (!g32 && x32) WTRY(16, 64, 32, 4, 3); if (!g32 && !x32) WTRY(16, 64, 64, 4, 3); }
        if (cfg == 3) { if (g32 && x32) WTRY(8, 32, 32, 4, 3); if (!g32 && x32) WTRY(8, 64, 32, 4, 3); if (!g32 && !x32) WTRY(8, 64, 64, 4, 3); }
#undef WTRY
    }
    // ring depth: the deepest that leaves two workgroups per CU
    if (g32 && x32) return launch<WCfg<16, 32, 32, 4, 3>>(g, x, sc, sh, dw, B, H, W, cout, cin, wdot, dots, dots_slots, s);
    if (!g32 && x32) return launch<WCfg<16, 64, 32, 4, 2>>(g, x, sc, sh, dw, B, H, W, cout, cin, wdot, dots, dots_slots, s);
    if (g32 && !x32) return launch<WCfg<16, 32, 64, 4, 2>>(g, x, sc, sh, dw, B, H, W, cout, cin, wdot, dots, dots_slots, s);
    // 32 input channels (half-empty 64-byte x pixels): 8-row tiles with a 3-deep ring (measured at 512^2, batch 8: 32 -> 32 98 -> 77 us,
    // 32 -> 64 113 -> 108; 64 -> 64 @256^2 does not gain: 61 -> 63.5)
    if (cin <= 32 && cout <= 64) return launch<WCfg<8, 64, 64, 4, 3>>(g, x, sc, sh, dw, B, H, W, cout, cin, wdot, dots, dots_slots, s);
    return launch<WCfg<16, 64, 64, 4, 2>>(g, x, sc, sh, dw, B, H, W, cout, cin, wdot, dots, dots_slots, s);
}
