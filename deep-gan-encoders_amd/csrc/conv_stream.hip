// Streaming 3x3 convolution for the HBM-bound layers (Cin, Cout <= 64 at >= 128^2; bf16), gfx950.
//
// The 512^2 / 1024^2 layers of the generator (64 / 32 channels) and the first encoder blocks (16 / 32 channels) sit far
// under the MFMA ridge: 107-288 flop/B.  conv_igemm_kernel ran them at 2.5-4x their HBM floor because every 16x16 tile is
// one latency chain (halo load -> VALU prologue -> LDS -> MFMA -> LDS transpose -> store) with the weights re-fetched per tile.
// This kernel is organised around the byte stream instead:
//
//   * a workgroup owns a column strip (TW pixels wide) of one sample and marches DOWN it, RS rows per step: activations
//     enter a ring of NR = 4*RS halo rows in LDS, every row crosses HBM once per strip (horizontal halo only: (TW+2)/TW);
//   * rows arrive by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no VALU), three groups of RS rows ahead of the
//     MFMAs, retired with counted s_waitcnt vmcnt; out-of-image halo pixels are read from a zero page (no border pass);
//   * the prologue affine is folded into the WEIGHTS: W'[b][o][i] = bf16(W[o][i] * sc[b][i]) once per workgroup (a strip
//     lies inside one sample) - the reference's own fused-modulation form (stylegan2_generator.py:858-864).  An additive
//     in_shift (instance norm, model/E/E.py:57,68) becomes W'.(x + sh/sc): a per-(b,o) constant inside the image and a
//     9-entry table T[o][tap] subtracted at border pixels (zero padding follows the norm);
//   * the whole W' lives in REGISTERS as MFMA A operands (D[o][p] = sum_k W'[o][k] X[p][k]: out channels are the M rows,
//     pixels the N columns), so the K loop reads ONE ds_read_b128 per MFMA (activation fragment, XOR-swizzled image) and no
//     weights; each lane ends up with 16 channels of one pixel;
//   * epilogue from registers: demodulation scale, noise (DMA'd rows, LDS), bias, lrelu/relu, gain; bf16 pack;
//     v_permlane32_swap pairs give every lane 16 contiguous bytes -> two global_store_dwordx4 per 32x32 tile, no LDS
//     transpose; (sum, sum of squares) statistics accumulate in registers over the whole strip and leave as one atomic per
//     channel per workgroup; the data-gradient mode reads the `dot_src` rows from a third DMA ring.
//
// One barrier per step.  LDS 34-70 KB -> 2-4 workgroups per CU, each with two row groups in flight.
//
// Reference math: model/stylegan2_generator.py:855-922 (stride-1 branch), model/E/E.py:50-85.
#include <type_traits>
#include "common.h"
#include <stdlib.h>
#include "conv_params.h"

__device__ __attribute__((aligned(256))) unsigned char dge_zero_page[2048];     // source of every out-of-image DMA lane

namespace {

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

enum { SMODE_PLAIN = 0, SMODE_STATS = 1, SMODE_DOT = 2 };

template <int CIN, int COUT, int TW, int RS, int WSPLIT, int MODE>
struct SCfg {
    static constexpr int PXB = CIN * 2, CH = PXB / 16, LOGCH = ilog2(CH);
    static constexpr int HWP = TW + 2;
    static constexpr int ROWB = HWP * PXB;
    static constexpr int NG = 4;                                  // ring = NG groups of RS rows
    static constexpr int GROUPB = RS * ROWB;
    static constexpr int XPIECES = (GROUPB + 1023) / 1024, XPW = (XPIECES + 3) / 4;
    static constexpr int KS = CIN / 16;
    static constexpr int MT = (COUT + 31) / 32;
    static constexpr int NTW = TW / 32;
    static constexpr int NGROUPB = RS * TW * 4;                    // noise rows of a group (f32)
    static constexpr int NPW = 1;                                   // one DMA slot per wave per group (only wave 0's is real)
    static constexpr int CPB = COUT * 2;                            // dot_src bytes per pixel
    static constexpr int DGROUPB = RS * TW * CPB;
    static constexpr int DPIECES = (DGROUPB + 1023) / 1024, DPW = MODE == SMODE_DOT ? (DPIECES + 3) / 4 : 0;
    static constexpr int NDMA = XPW + NPW + DPW;                    // DMA instructions per wave per group (uniform)
    static constexpr int X_OFF = 0;
    static constexpr int N_OFF = X_OFF + NG * XPIECES * 1024;      // groups are padded to whole pieces
    static constexpr int D_OFF = N_OFF + NG * 1024;
    static constexpr int DUMMY_OFF = D_OFF + (MODE == SMODE_DOT ? NG * DPIECES * 1024 : 0);
    static constexpr int T_OFF = DUMMY_OFF + 1024;                  // T table [COUT][12] f32 (9 taps, sum) + reduce scratch
    static constexpr int T_BYTES = 64 * 12 * 4 + 3 * 64 * 4 + 4 * 64 * 4;   // + epilogue constants osc / bg / nwg [64] + border sums [4][64]
    static constexpr int R_OFF = T_OFF + T_BYTES;                   // statistics reduction scratch [4 waves][64 ch][2]
    static constexpr int LDS_BYTES = R_OFF + 4 * 64 * 2 * 4;
    static constexpr int XGROUP_STRIDE = XPIECES * 1024;
    static constexpr int DGROUP_STRIDE = DPIECES * 1024;
    static_assert(MT % WSPLIT == 0 && MT / WSPLIT == 1, "one M tile (32 out channels) per wave");
    static_assert(RS * WSPLIT == 4, "4 waves = RS pixel rows x WSPLIT channel halves");
    static_assert(NGROUPB <= 1024, "noise group is one DMA piece");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// swizzle of the 16-byte channel chunks of a pixel in the LDS image (conflict-free ds_read_b128 of 16 consecutive pixels)
template <int LOGCH> __device__ __forceinline__ int chunk_swz(int px) { return (px >> (4 - LOGCH)) & ((1 << LOGCH) - 1); }

template <int CIN, int COUT, int TW, int RS, int WSPLIT, int MODE>
__global__ __launch_bounds__(256, 2) void conv_stream_kernel(ConvParams p, int nseg, int seg_rows) {
    using C = SCfg<CIN, COUT, TW, RS, WSPLIT, MODE>;
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    const unsigned lds0 = lds_off(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int prow = wave % RS, mh = wave / RS;               // this wave's row within a step, its 32-channel half
    const int n31 = lane & 31, kh = lane >> 5;

    // ---- job: (sample, strip, segment)
    const int nstrips = (p.W + TW - 1) / TW;
    int job = blockIdx.x;
    const int seg = job % nseg; job /= nseg;
    const int strip = job % nstrips;
    const int b = job / nstrips;
    const int x0 = strip * TW;
    const int r0 = seg * seg_rows;
    const int rows = min(seg_rows, p.H - r0);                  // output rows of this segment (> 0 by construction)
    const int nsteps = (rows + RS - 1) / RS;
    const int glast = (rows + 1) / RS;                         // last group holding a needed halo row (halo index rows + 1)

    const bf16_t* __restrict__ Xb = (const bf16_t*)p.x + (size_t)b * p.H * p.W * CIN;
    bf16_t* __restrict__ Yb = (bf16_t*)p.y + (size_t)b * p.H * p.W * COUT;
    const bf16_t* __restrict__ DOTb = (MODE == SMODE_DOT) ? (const bf16_t*)p.dot_src + (size_t)b * p.H * p.W * COUT : nullptr;
    const float* __restrict__ NZb = p.noise ? p.noise + (size_t)b * p.noise_bstride : nullptr;

    // ---- weights -> registers, prologue affine folded in:  W'[o][k] = bf16(W[o][k] * sc[b][k])
    uint4 wf[9][C::KS];
    {
        const bf16_t* __restrict__ Wp = (const bf16_t*)p.w;
        const int o = mh * 32 + n31;
        float sc[C::KS][8], rt[C::KS][8];                       // scale and shift/scale of this lane's input channels
#pragma unroll
        for (int ks = 0; ks < C::KS; ks++) {
            const int k0 = ks * 16 + kh * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                sc[ks][e] = p.in_scale ? p.in_scale[b * CIN + k0 + e] : 1.f;
                const float sh = p.in_shift ? p.in_shift[b * CIN + k0 + e] : 0.f;
                rt[ks][e] = sc[ks][e] != 0.f ? sh / sc[ks][e] : 0.f;
            }
        }
        float* __restrict__ T = (float*)(lds + C::T_OFF);      // [64][12]: taps 0..8, [9] = sum over the taps
        float tall = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            float tt = 0.f;
#pragma unroll
            for (int ks = 0; ks < C::KS; ks++) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (o < p.Ntot) v = *(const uint4*)(Wp + ((size_t)(tap * p.Ntot + o) * CIN + ks * 16 + kh * 8));
                if (p.in_scale || p.in_shift) {
                    float f[8];
                    unpack16(v, f, (bf16_t*)nullptr);
#pragma unroll
                    for (int e = 0; e < 8; e++) f[e] *= sc[ks][e];
                    v = pack16(f, (bf16_t*)nullptr);
                    if (p.in_shift) {                           // T uses the ROUNDED weights: W'.(x + sh/sc) is then exact in W'
                        unpack16(v, f, (bf16_t*)nullptr);
#pragma unroll
                        for (int e = 0; e < 8; e++) tt += f[e] * rt[ks][e];
                    }
                }
                wf[tap][ks] = v;
            }
            if (p.in_shift) {
                tt += __shfl_xor(tt, 32, 64);                   // the two K halves of the row
                if (kh == 0) T[o * 12 + tap] = tt;
                tall += tt;
            }
        }
        if (p.in_shift && kh == 0) T[o * 12 + 9] = tall;
    }
    __syncthreads();

    // ---- per-channel epilogue constants in LDS (read back as float4 per group of 4 channels: keeps 48 registers free)
    float* __restrict__ EPC = (float*)(lds + C::T_OFF + 64 * 12 * 4);      // osc[64] | bg[64] | nwg[64]
    if (tid < 64) {
        const float* __restrict__ T = (const float*)(lds + C::T_OFF);
        const int ch = tid;
        const bool cv = ch < COUT;
        const float sc_o = (p.out_scale && cv) ? p.out_scale[b * COUT + ch] : 1.f;
        const float o_ = sc_o * p.gain;
        float bb = (p.bias && cv) ? p.bias[ch] * p.bias_scale * p.gain : 0.f;
        if (p.in_shift && cv) bb += T[ch * 12 + 9] * o_;                    // interior value of the folded shift
        EPC[ch] = o_;
        EPC[64 + ch] = bb;
        EPC[128 + ch] = (p.noise && cv) ? p.noise_w[ch * p.noise_w_stride] * p.gain : 0.f;
        if (p.in_shift) {
            float* __restrict__ BRD = EPC + 192;                             // [4][64]: left column, right column, top row, bottom row
            BRD[ch] = T[ch * 12 + 0] + T[ch * 12 + 3] + T[ch * 12 + 6];
            BRD[64 + ch] = T[ch * 12 + 2] + T[ch * 12 + 5] + T[ch * 12 + 8];
            BRD[128 + ch] = T[ch * 12 + 0] + T[ch * 12 + 1] + T[ch * 12 + 2];
            BRD[192 + ch] = T[ch * 12 + 6] + T[ch * 12 + 7] + T[ch * 12 + 8];
        }
    }
    __syncthreads();
    const float slope = p.act == DGE_ACT_LRELU ? 0.2f : (p.act == DGE_ACT_RELU ? 0.f : 1.f);

    // ---- B-fragment byte offsets inside a halo row (N tile 0; tile 1 adds 32 pixels), per dx.  The k step only changes the
    //      chunk bits: chunk = (ks*2 + kh) ^ swz(px), and every other term of the address is a multiple of the pixel pitch,
    //      so address(ks) = address(0) ^ (ks << 5): one v_xor per read instead of KS registers per dx.
    int boff[3];
#pragma unroll
    for (int dx = 0; dx < 3; dx++) {
        const int px = n31 + dx;
        boff[dx] = px * C::PXB + ((kh ^ chunk_swz<C::LOGCH>(px)) << 4);
    }
    static_assert(C::ROWB % C::PXB == 0 && (C::PXB & (C::PXB - 1)) == 0, "pixel pitch is a power of two dividing every row offset");

    // ---- DMA descriptors (per lane, per instruction slot; constant over the groups except the row)
    int x_hr[C::XPW], x_col[C::XPW];                             // row within the group, byte offset within the image row (-1: outside)
#pragma unroll
    for (int i = 0; i < C::XPW; i++) {
        const int pc = wave + 4 * i;
        const int u = pc * 64 + lane;
        const int per_row = C::HWP * C::CH;
        const int hr = u / per_row, rem = u - hr * per_row;
        const int px = rem >> C::LOGCH, cs = rem & (C::CH - 1);
        const int gx = x0 - 1 + px;
        const bool ok = (pc < C::XPIECES) && (hr < RS) && ((unsigned)gx < (unsigned)p.W);
        x_hr[i] = hr;
        x_col[i] = ok ? gx * C::PXB + ((cs ^ chunk_swz<C::LOGCH>(px)) << 4) : -1;
    }
    // noise: one piece per group (wave 0): lane -> 4 consecutive pixels of one row
    const int nz_row = lane / (TW / 4), nz_px = (lane % (TW / 4)) * 4;
    const bool nz_ok = (wave == 0) && p.noise && (nz_row < RS) && (x0 + nz_px < p.W);
    int d_hr[C::DPW > 0 ? C::DPW : 1], d_col[C::DPW > 0 ? C::DPW : 1];
    if constexpr (MODE == SMODE_DOT) {
#pragma unroll
        for (int i = 0; i < C::DPW; i++) {
            const int pc = wave + 4 * i;
            const int u = pc * 64 + lane;                           // 16-byte unit of the group's [RS][TW][COUT] bf16 image
            constexpr int per_row = TW * C::CPB / 16;
            const int hr = u / per_row, rem = u - hr * per_row;
            const int px = rem / (C::CPB / 16), cs = rem % (C::CPB / 16);
            const bool ok = (pc < C::DPIECES) && (hr < RS) && (x0 + px < p.W);
            d_hr[i] = hr;
            d_col[i] = ok ? (x0 + px) * C::CPB + cs * 16 : -1;
        }
    }
    const unsigned char* zero = dge_zero_page + lane * 16;
    const size_t xrow_bytes = (size_t)p.W * C::PXB, drow_bytes = (size_t)p.W * C::CPB;

    auto issue_group = [&](int g) {
        const int slot = g & (C::NG - 1);
        const bool live = g <= glast;
        const int hrow0 = r0 - 1 + g * RS;                          // image row of the group's first halo row
#pragma unroll
        for (int i = 0; i < C::XPW; i++) {
            const int pc = wave + 4 * i;
            const int row = hrow0 + x_hr[i];
            const bool ok = live && x_col[i] >= 0 && (unsigned)row < (unsigned)p.H;
            const unsigned char* src = ok ? (const unsigned char*)Xb + (size_t)row * xrow_bytes + x_col[i] : zero;
            const unsigned dst = pc < C::XPIECES ? lds0 + C::X_OFF + slot * C::XGROUP_STRIDE + pc * 1024 : lds0 + C::DUMMY_OFF;
            glds16(src, __builtin_amdgcn_readfirstlane(dst));
        }
        {   // noise rows of the OUTPUT rows r0 + g*RS .. (they are consumed at step g)
            const int row = r0 + g * RS + nz_row;
            const bool ok = live && nz_ok && row < p.H;
            const unsigned char* src = ok ? (const unsigned char*)(NZb + (size_t)row * p.W + x0 + nz_px) : zero;
            const unsigned dst = wave == 0 ? lds0 + C::N_OFF + slot * 1024 : lds0 + C::DUMMY_OFF;
            glds16(src, __builtin_amdgcn_readfirstlane(dst));
        }
        if constexpr (MODE == SMODE_DOT) {
#pragma unroll
            for (int i = 0; i < C::DPW; i++) {
                const int pc = wave + 4 * i;
                const int row = r0 + g * RS + d_hr[i];
                const bool ok = live && d_col[i] >= 0 && row < p.H;
                const unsigned char* src = ok ? (const unsigned char*)DOTb + (size_t)row * drow_bytes + d_col[i] : zero;
                const unsigned dst = pc < C::DPIECES ? lds0 + C::D_OFF + slot * C::DGROUP_STRIDE + pc * 1024 : lds0 + C::DUMMY_OFF;
                glds16(src, __builtin_amdgcn_readfirstlane(dst));
            }
        }
    };

    float s0[16], s1[16];                                         // statistics over the whole strip segment (registers)
#pragma unroll
    for (int r = 0; r < 16; r++) { s0[r] = 0.f; s1[r] = 0.f; }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // parameter loads are done before the counted DMA stream starts
    issue_group(0); issue_group(1); issue_group(2);

    for (int s = 0; s < nsteps; s++) {
        // groups s and s+1 must have landed (step s reads halo rows s*RS .. s*RS+RS+1); group s+2 may stay in flight.
        // vmcnt counts this wave's stores too: they only make the wait conservative (loads retire in order among loads).
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(C::NDMA) : "memory");
        __builtin_amdgcn_s_barrier();                              // every wave's pieces landed; every wave finished step s-1
        if (!(p.dbg & 1)) issue_group(s + 3);                      // into the slot of group s-1, dead since the barrier
        else { for (int i = 0; i < C::NDMA; i++) glds16(zero, __builtin_amdgcn_readfirstlane(lds0 + C::DUMMY_OFF)); }

        const int ro = s * RS + prow;                              // output row (segment relative); halo rows ro .. ro+2
        const int gy = r0 + ro;
        if (ro < rows) {
            unsigned rowbase[3];
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const int hr = ro + dy;
                rowbase[dy] = C::X_OFF + ((hr / RS) & (C::NG - 1)) * C::XGROUP_STRIDE + (hr % RS) * C::ROWB;
            }
            const float* __restrict__ nzrow = (const float*)(lds + C::N_OFF + (s & (C::NG - 1)) * 1024) + prow * TW;
            const unsigned char* __restrict__ drow = lds + C::D_OFF + (s & (C::NG - 1)) * C::DGROUP_STRIDE + prow * TW * C::CPB;
#pragma unroll 1
            for (int nt = 0; nt < C::NTW; nt++) {
                f32x16_t acc;
#pragma unroll
                for (int r = 0; r < 16; r++) acc[r] = 0.f;
                const unsigned char* base = lds + nt * 32 * C::PXB;
                // software pipelined: the fragment of step q+1 is read before the MFMA of step q
                constexpr int NQ = 9 * C::KS;
                uint4 bfr[2];
                bfr[0] = *(const uint4*)(base + rowbase[0] + boff[0]);
                if (!(p.dbg & 2))
                StaticFor<NQ>::run([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    if constexpr (q + 1 < NQ) {
                        constexpr int t1 = (q + 1) / C::KS, k1 = (q + 1) % C::KS;
                        bfr[(q + 1) & 1] = *(const uint4*)(base + ((rowbase[t1 / 3] + boff[t1 % 3]) ^ (k1 << 5)));
                    }
                    constexpr int t = q / C::KS, k = q % C::KS;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&wf[t][k], *(const bf16x8_t*)&bfr[q & 1], acc, 0, 0, 0);
                });
                // ---------------- epilogue of the 32 (channels) x 32 (pixels) tile: this lane = pixel px, 16 channels
                const int px = nt * 32 + n31;
                const int gx = x0 + px;
                const bool pv = gx < p.W;
                const float nz = p.noise ? nzrow[px] : 0.f;
                float v[16];
                if (p.in_shift) {
                    // zero padding follows the norm: at border pixels the taps that fall outside do not carry the shift.
                    // BRD = [4][64] per-channel sums of T over the left / right tap column and the top / bottom tap row.
                    const float* __restrict__ BRD = (const float*)(lds + C::T_OFF + 64 * 12 * 4 + 3 * 64 * 4);
                    const float* __restrict__ T = (const float*)(lds + C::T_OFF);
                    const bool top = gy == 0, bot = gy == p.H - 1;                   // wave-uniform (one row per wave)
                    const float cl = gx == 0 ? 1.f : 0.f, cr = gx == p.W - 1 ? 1.f : 0.f;
                    if (__builtin_amdgcn_ballot_w64((cl + cr) != 0.f)) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int ch0 = mh * 32 + 8 * j + 4 * kh;
                            const float4 l4 = *(const float4*)(BRD + ch0), r4 = *(const float4*)(BRD + 64 + ch0);
                            acc[4 * j] -= cl * l4.x + cr * r4.x; acc[4 * j + 1] -= cl * l4.y + cr * r4.y;
                            acc[4 * j + 2] -= cl * l4.z + cr * r4.z; acc[4 * j + 3] -= cl * l4.w + cr * r4.w;
                        }
                    }
                    if (top | bot) {
                        const int side = top ? 2 : 3;
                        const int c0 = top ? 0 : 6;                                  // corner taps of that row: c0 (left), c0 + 2 (right)
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int ch = mh * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
                            float corr = BRD[side * 64 + ch] - cl * T[ch * 12 + c0] - cr * T[ch * 12 + c0 + 2];
                            if (top & bot) corr += BRD[3 * 64 + ch] - cl * T[ch * 12 + 6] - cr * T[ch * 12 + 8];   // H == 1
                            acc[r] -= corr;
                        }
                    }
                }
                if constexpr (MODE == SMODE_DOT) {
                    // data-gradient mode: (sum acc*dot_src, sum acc) of the raw accumulator, then the per-channel scale
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint2 dd = *(const uint2*)(drow + px * C::CPB + (mh * 32 + 8 * j + 4 * kh) * 2);
                        const float d0 = __uint_as_float(dd.x << 16), d1 = __uint_as_float(dd.x & 0xffff0000u);
                        const float d2 = __uint_as_float(dd.y << 16), d3 = __uint_as_float(dd.y & 0xffff0000u);
                        const float a0 = pv ? acc[4 * j] * p.gain : 0.f, a1 = pv ? acc[4 * j + 1] * p.gain : 0.f;
                        const float a2 = pv ? acc[4 * j + 2] * p.gain : 0.f, a3 = pv ? acc[4 * j + 3] * p.gain : 0.f;
                        s0[4 * j] += a0 * d0; s0[4 * j + 1] += a1 * d1; s0[4 * j + 2] += a2 * d2; s0[4 * j + 3] += a3 * d3;
                        s1[4 * j] += a0; s1[4 * j + 1] += a1; s1[4 * j + 2] += a2; s1[4 * j + 3] += a3;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int ch0 = mh * 32 + 8 * j + 4 * kh;
                    const float4 o4 = *(const float4*)(EPC + ch0), b4 = *(const float4*)(EPC + 64 + ch0), n4 = *(const float4*)(EPC + 128 + ch0);
                    const float oo[4] = {o4.x, o4.y, o4.z, o4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w}, nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float t = fmaf(acc[4 * j + q], oo[q], fmaf(nn[q], nz, bb[q]));
                        v[4 * j + q] = fmaxf(t, t * slope);
                    }
                }
                if constexpr (MODE == SMODE_STATS) {
#pragma unroll
                    for (int r = 0; r < 16; r++) { const float t = pv ? v[r] : 0.f; s0[r] += t; s1[r] += t * t; }
                }
                // bf16 pack; v_permlane32_swap pairs: lanes < 32 end up with channels 8k..8k+7, lanes >= 32 with 8(k+1)..8(k+1)+7
                unsigned w[8];
#pragma unroll
                for (int j = 0; j < 4; j++) { w[2 * j] = pack2bf(v[4 * j], v[4 * j + 1]); w[2 * j + 1] = pack2bf(v[4 * j + 2], v[4 * j + 3]); }
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    // groups k (channels 8k + 4kh ..) and k+1
                    auto r0_ = __builtin_amdgcn_permlane32_swap(w[2 * k], w[2 * k + 2], false, false);
                    auto r1_ = __builtin_amdgcn_permlane32_swap(w[2 * k + 1], w[2 * k + 3], false, false);
                    const uint4 o16 = make_uint4(r0_[0], r1_[0], r0_[1], r1_[1]);
                    const int chb = mh * 32 + 8 * (k + kh);                 // first of this lane's 8 contiguous channels
                    if (pv && chb < COUT && !(p.dbg & 4))
                        *(uint4*)(Yb + ((size_t)gy * p.W + gx) * COUT + chb) = o16;
                }
            }
        }
    }

    // ---- statistics: reduce over the 32 pixel lanes, combine the row waves in LDS, one atomic per channel per workgroup
    if constexpr (MODE != SMODE_PLAIN) {
        if (p.stats) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) { s0[r] += __shfl_xor(s0[r], m, 64); s1[r] += __shfl_xor(s1[r], m, 64); }
            }
            float* __restrict__ red = (float*)(lds + C::R_OFF);              // [wave][64 ch][2]
            if (n31 == 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int ch = mh * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
                    red[(wave * 64 + ch) * 2] = s0[r]; red[(wave * 64 + ch) * 2 + 1] = s1[r];
                }
            }
            __syncthreads();
            float* __restrict__ ST = p.stats + (size_t)(blockIdx.x % p.stats_slots) * p.B * COUT * 2 + (size_t)b * COUT * 2;
            for (int i = tid; i < COUT * 2; i += 256) {
                const int ch = i >> 1, k = i & 1;
                const int m = ch >> 5;                                       // waves holding this channel: mh == m
                float a = 0.f;
                for (int wv = 0; wv < 4; wv++) if (wv / RS == m) a += red[(wv * 64 + ch) * 2 + k];
                atomicAdd(ST + i, a);
            }
        }
    }
}

template <int CIN, int COUT, int TW, int RS, int WSPLIT, int MODE>
int launch_stream(const ConvParams& p0, hipStream_t s) {
    using C = SCfg<CIN, COUT, TW, RS, WSPLIT, MODE>;
    ConvParams p = p0;
    { const char* e = getenv("DGE_STREAM_DBG"); p.dbg = e ? atoi(e) : 0; }
    const int nstrips = (p.W + TW - 1) / TW;
    // segments: enough workgroups to fill the chip about twice at the LDS-limited residency, segments >= 8 steps long
    int nseg = (2 * 2 * 256 + p.B * nstrips - 1) / (p.B * nstrips);
    { const char* e = getenv("DGE_STREAM_NSEG"); if (e) nseg = atoi(e); }
    int maxseg = p.H / (8 * RS); if (maxseg < 1) maxseg = 1;
    if (nseg > maxseg) nseg = maxseg;
    if (nseg < 1) nseg = 1;
    int seg_rows = ((p.H + nseg - 1) / nseg + RS - 1) / RS * RS;
    nseg = (p.H + seg_rows - 1) / seg_rows;
    const long grid = (long)p.B * nstrips * nseg;
    auto kern = conv_stream_kernel<CIN, COUT, TW, RS, WSPLIT, MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_set = true;
    }
    dge_note_kernel("conv_stream<bf16,%d,%d,%d,%d,%s>", CIN, COUT, TW, RS, MODE == SMODE_PLAIN ? "plain" : (MODE == SMODE_STATS ? "stats" : "dot"));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), C::LDS_BYTES, s, p, nseg, seg_rows);
    DGE_LAUNCH_CHECK("conv_stream");
    return 0;
}

template <int CIN, int COUT, int TW, int RS, int WSPLIT>
int launch_mode(const ConvParams& p, hipStream_t s) {
    if (p.dot_src) return launch_stream<CIN, COUT, TW, RS, WSPLIT, SMODE_DOT>(p, s);
    if (p.stats) return launch_stream<CIN, COUT, TW, RS, WSPLIT, SMODE_STATS>(p, s);
    return launch_stream<CIN, COUT, TW, RS, WSPLIT, SMODE_PLAIN>(p, s);
}

}  // namespace

// compile-time loop helper shared with conv_igemm.hip (kept local: separate translation units)
// (StaticFor is defined in common.h)

// Eligibility of the streaming kernel for a launch (bf16, 3x3, stride 1, no fused resampling / addend / ReLU prologue).
bool dge_conv_stream_eligible(const ConvParams& p, int dtype, int ksize) {
    if (dtype != DGE_BF16 || ksize != 3 || p.up || p.in_s2d || p.in_up2 || p.in_relu || p.addend) return false;
    if (!(p.Cin == 16 || p.Cin == 32 || p.Cin == 64) || !(p.Cout == 16 || p.Cout == 32 || p.Cout == 64)) return false;
    if (p.W % 4 != 0 || p.W < 64 || p.H < 32) return false;
    if ((long)p.H * p.W < 128L * 128) return false;
    if (p.dot_src && !p.stats) return false;
    if (p.noise && p.noise_w == nullptr) return false;
    if (getenv("DGE_NO_STREAM")) return false;
    return true;
}

int dge_conv_stream_launch(const ConvParams& p, hipStream_t s) {
#define GO(CI, CO, TW, RS, WS) if (p.Cin == CI && p.Cout == CO) return launch_mode<CI, CO, TW, RS, WS>(p, s)
    GO(16, 16, 64, 4, 1); GO(16, 32, 64, 4, 1); GO(32, 16, 64, 4, 1); GO(32, 32, 64, 4, 1);
    GO(64, 16, 32, 4, 1); GO(64, 32, 32, 4, 1);
    GO(16, 64, 64, 2, 2); GO(32, 64, 64, 2, 2); GO(64, 64, 64, 2, 2);
#undef GO
    dge_set_error("conv_stream: unsupported channel configuration %d -> %d", p.Cin, p.Cout);
    return -1;
}
