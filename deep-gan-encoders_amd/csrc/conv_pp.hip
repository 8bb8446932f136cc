// conv_pp: 3x3 stride-1 convolution of the MFMA-bound layers (C >= 128, 32^2 .. 256^2) as a ping-pong implicit GEMM.
//
// Reference math: ModulateConvBlock.forward, stride-1 branch, in the reference's own FUSED-modulation form
// (model/stylegan2_generator.py:858-875: the style multiplies the weight, the demodulation divides it, one weight per sample;
// :898-904 conv; :911-921 noise, bias, lrelu*sqrt2), and the plain conv + bias + ReLU layers of LPIPS' VGG16.
//
// Why a second kernel next to conv_igemm: that one runs a workgroup as ONE serial pipeline (stage -> barrier -> stage) and gets
// its overlap from a second resident workgroup; measured 0.40-0.49 matrix-pipe utilisation with 38 % of wave time parked at
// s_waitcnt / barriers.  Here the two waves of a SIMD are given opposite roles on purpose:
//   * 8 waves = two groups of four (wave w and w+4 share a SIMD).  In every PHASE one group issues the 16 MFMAs of a cluster
//     (one tap of one 32-channel K chunk: 128 pixels x 64 channels per wave) while the other group reads the 12 fragments of
//     ITS next cluster from LDS and issues the LDS-DMAs of the stages to come; one s_barrier per phase flips the roles.
//   * every operand reaches LDS by DMA, nothing is staged through registers: the halo tile (18 x 34 pixels x 64 B per chunk)
//     through a buffer descriptor of the sample's image (out-of-range lanes write zeros = the zero padding), the weights as a
//     linear copy of an LDS image prepared once per forward by dge_pack_conv_pp - per SAMPLE when the layer is modulated (style,
//     demodulation and activation gain folded into the weight).  No VALU instruction touches an operand on its way in.
//   * the weight ring is a whole K chunk deep (9 taps: slot = tap, compile time), requested 8 clusters ahead; the next halo
//     tile is requested during the first taps of the current chunk; arrival is counted with s_waitcnt vmcnt(N).
//   * LDS images are "part-major" per 1 KiB DMA piece (16 rows x 4 parts of 16 B: byte = part*256 + row*16), which makes every
//     ds_read_b128 fragment read conflict free for any 32 consecutive pixels / the permuted weight rows without an XOR swizzle.
//   * accumulators are transposed (weights = MFMA A operand, rows permuted by tr_chan_of_row): a lane ends with two runs of 8
//     channels of one pixel and stores 16-byte vectors straight from registers.
#include <type_traits>
#include <stdlib.h>
#include "common.h"
#include "../../include/dge_hip.h"

typedef __attribute__((ext_vector_type(4))) unsigned rsrc_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

namespace {

struct PPParams {
    const bf16_t* x; const bf16_t* w; bf16_t* y;
    long long w_bstride;                      // bytes between the samples' weight images (0 = shared)
    const float* bias; const float* noise; const float* noise_w;
    int B, H, W, Cin, Cout;
    int noise_bstride, noise_w_stride, act;
    float bias_scale, gain;
    int tiles_x, tiles_y, ntn, nchunks;
    int dbg;
    int VH, VW;                               // extent of the halo source in (virtual) pixels: H x W, or (H + 1) x (W + 1) under in_t2d
    // halo source: plain (SH x SW pixels of SC = Cin channels) or space-to-depth (in_s2d: the image is [2H][2W][Cin/4], logical channel
    // (phase py*2+px, c) of pixel (y, x) lives at pixel (2y+py, 2x+px); a 32-channel K chunk lies inside one phase)
    int s2d, SW, SC, cpp;                     // cpp: K chunks per phase
    long long x_bstride;                      // bytes per sample of the source image
    // data-gradient epilogue (EPI == 2), the menu of conv_epilogue.h MODE 1 / 2 on the raw accumulator a:
    //   stats += (sum a*dot, sum a); v = a*out_scale + add_scale*addend; mask_relu: v *= [dot > 0];
    //   prep: v = g_z = v*prep_gain*lrelu'(dot), prep_stats += (sum g_z*(z - ns*noise), sum g_z)       (ConvParams::prep)
    const float* out_scale; const bf16_t* dot; const bf16_t* add; float add_scale;
    float* stats; float* prep_stats; int stats_slots;
    int prep, mask_relu, prep_noise_bstride;
    float prep_gain; const float* prep_noise; const float* prep_ns;
};

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
// one 1 KiB piece, global -> LDS, through the image descriptor: lanes whose offset is out of range write zeros
__device__ __forceinline__ void dma_buf(unsigned voff, rsrc_t rs, unsigned soff, unsigned m0v) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" : : "v"(voff), "s"(m0v), "s"(rs), "s"(soff) : "memory");
}
// one 1 KiB piece of the weight image: wave-uniform 64-bit base + lane*16
__device__ __forceinline__ void dma_lin(unsigned voff, unsigned long long sbase, unsigned m0v) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(m0v) : "memory");
}
__device__ __forceinline__ int tr_chan_of_row(int m) {
    const int j = m >> 3, k = (m >> 2) & 1, i = m & 3;
    return 16 * (j >> 1) + 8 * k + 4 * (j & 1) + i;
}

// development aid (DGE_CONV_DBG bit 8 = 256): shader-clock stamps of workgroup 0, waves 0 and 4, read back by dge_dbg_pp_prof
__device__ long long g_pp_prof[2][1024];

// LDS map: halo buffer 0 | weight slots 0-2 | halo buffer 1 (at 64 KiB: the buffer toggle is one XOR) | weight slots 3-8
constexpr int H1_OFF = 65536, WLO_OFF = 40960, WHI_OFF = 106496, PP_LDS = WHI_OFF + 6 * 8192;   // + 8 x 1 KiB of epilogue staging
__host__ __device__ constexpr int wslot_off(int t) { return t < 3 ? WLO_OFF + t * 8192 : WHI_OFF + (t - 3) * 8192; }

// Persistent: workgroup i (XCD i % 8) walks the tiles x*Tx + j + k*(workgroups per XCD) of its XCD's contiguous range; the cluster
// stream runs on across tile seams (the last chunk of a tile requests the first halo tile and the first weights of the next one),
// so a seam costs the epilogue's VALU work and the issue of its stores, not a drain + refill of the pipeline.  vmcnt counts
// stores too, in issue order: the first seven waits after an epilogue that issued all of its 16 stores allow 16 more outstanding
// operations (a wave with out-of-image rows issues fewer stores and keeps the strict counts, i.e. waits for them).
// The DMAs of a cluster are issued between the MFMAs of its COMPUTE phase (two short asm statements with precomputed scalar
// operands: in front of the LOAD phase they were on the critical path of the phase).  In the workgroup's last chunk the
// requests go on as dummies (own halo tile, own first weight slots: nobody reads them) so that the wait counts stay the same.
// T2D (phase-form adjoint of the up layers, dge_fir_t2d + ConvParams::in_t2d): the source image has one more row and column than the
// output grid and only the taps (dy, dx) in {1, 2}^2 exist: a chunk is 4 clusters.  The weight ring then holds two chunks (8 slots,
// slot offsets in SGPRs, swapped per chunk), requests run 7 clusters ahead, and the five halo pieces of a wave go out with taps 0-2.
template <int PT, bool DBG, int EPI, bool T2D = false>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(PPParams p_) {
    // EPI 0 / 1: forward epilogue (scalar noise strength / noise weight per channel); EPI 2 + 2*flavour + addend: data-gradient epilogue,
    // flavour 0 = statistics + fused tail backward (prep), 1 = statistics, 2 = ReLU mask
    constexpr bool NWC = EPI == 1, DG = EPI >= 2, ADD = DG && ((EPI - 2) & 1), PREP = DG && ((EPI - 2) >> 1) == 0, MASK = DG && ((EPI - 2) >> 1) == 2;
    constexpr int TH = 4 * PT, HH = TH + 2, HPIX = HH * 34, HPIECES = (HPIX + 15) / 16, HPW = (HPIECES + 7) / 8;
    static_assert(HPW * 8 * 1024 <= WLO_OFF && HPW <= 8, "halo tile (incl. the all-zero pieces of the waves that have one piece less)");
    constexpr int NTAP = T2D ? 4 : 9, WAHEAD = T2D ? 7 : 8;              // clusters per chunk; clusters the weight requests run ahead
    constexpr int VM_B = WAHEAD - 2;                                     // W pieces a wave has issued after the one cluster c + 1 needs
    // ... after its last halo piece of the chunk (9 taps: pieces with taps 0-4, then the W pieces of taps 4-7; T2D: last piece with tap 2)
    constexpr int VM_T8 = T2D ? 1 : ((VM_B + 3 - HPW) < VM_B ? (VM_B + 3 - HPW) : VM_B);
    constexpr int RLX = T2D ? 3 : 7;                                     // clusters after an epilogue whose W piece was requested before it
    constexpr int NSTORE = 4 * PT;                                       // stores of one epilogue, per wave
    __shared__ __attribute__((aligned(1024))) unsigned char lds[PP_LDS + 8192];
    const unsigned lds0 = lds_off(lds);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, kh = lane >> 5;
    const int wave = rfl(tid >> 6), g = wave >> 2, q = wave & 3;
    // the parameters are re-read from the kernel argument segment where they are used (tile seams, epilogue): held in SGPRs for
    // the whole kernel they push the main loop's scalars into spills
    typedef const __attribute__((address_space(4))) PPParams* kparg_t;       // (the struct is the kernel's only argument: offset 0)
    const kparg_t kp = (kparg_t)__builtin_amdgcn_kernarg_segment_ptr();
    auto P = [&]() { kparg_t r = kp; asm volatile("" : "+s"(r)); return r; };
    const int nchunks = p_.nchunks;
    const int dbg = DBG ? p_.dbg : 0;
    int pidx = 0;
    auto stamp = [&](int tag) {
        if constexpr (DBG) {
            if ((dbg & 256) && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && pidx < 1022) {
                g_pp_prof[g][pidx++] = ((long long)tag << 48) | (long long)(__builtin_readcyclecounter() & 0xffffffffffffll);
                g_pp_prof[g][1023] = pidx;
            }
        }
    };
    // tiles of this workgroup
    const int ntiles = p_.tiles_x * p_.tiles_y * p_.B * p_.ntn;
    const int per_xcd = (ntiles + 7) >> 3, stride = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    int tile = xcd * per_xcd + (blockIdx.x >> 3);
    const int tile_end = min((xcd + 1) * per_xcd, ntiles);
    if (tile >= tile_end) return;                                        // (whole workgroup)

    int x0, y0, b, nt;                         // tile whose accumulators are live
    rsrc_t rs;                                 // image descriptor the halo DMAs read through (the NEXT tile's during a tile's last chunk)
    unsigned hoff[HPW];
    unsigned long long wt;                     // weight image of the current tile (+ this wave's piece)
    auto decode = [&](int id, int& tx0, int& ty0, int& tb, int& tnt) {
        const auto p = P();
        tnt = id % p->ntn; id /= p->ntn;                                // N tiles of one pixel tile back to back: the halo stays in L2
        tx0 = (id % p->tiles_x) * 32; id /= p->tiles_x;
        ty0 = (id % p->tiles_y) * TH; tb = id / p->tiles_y;
    };
    auto halo_src = [&](int tx0, int ty0, int tb) {
        const auto p = P();
        const unsigned long long xb = (unsigned long long)p->x + (unsigned long long)tb * (unsigned long long)p->x_bstride;
        rs = make_rsrc(xb, (unsigned)p->x_bstride);
        const int mul = p->s2d ? 2 : 1;
        // source offsets of this wave's halo pieces (piece P = wave + 8k: 16 pixels x 4 parts, lane = part*16 + pixel)
        // (recomputed per tile on purpose: hoisted out of the tile loop these lane constants are spilled around it)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
#pragma unroll
        for (int k = 0; k < HPW; k++) {
            const int Pc = wave + 8 * k;
            const int hp = Pc * 16 + (lane_o & 15), qd = lane_o >> 4;
            const int hr = hp / 34, hx = hp - hr * 34;
            const int gy = ty0 + hr - 1, gx = tx0 + hx - 1;
            const bool ok = (hp < HPIX) & ((unsigned)gy < (unsigned)p->VH) & ((unsigned)gx < (unsigned)p->VW);
            hoff[k] = ok ? (unsigned)((((gy * p->SW + gx) * mul) * p->SC + qd * 8) * 2) : 0x80000000u;
        }
    };
    auto wbase = [&](int tb, int tnt) {
        const auto p = P();
        return (unsigned long long)p->w + (unsigned long long)tb * p->w_bstride + (unsigned long long)tnt * nchunks * ((unsigned long long)NTAP * 8192) +
               (unsigned)wave * 1024u;
    };
    const unsigned wvoff = lane * 16;
    const unsigned wm0 = rfl(lds0 + wave * 1024);                       // this wave's piece of a weight slot / halo buffer 0

    // fragment addresses.  Pixels: halo row rr of this wave's strip, column l31 + dx -> linear halo index hp, image position
    // (hp >> 4) KiB + (hp & 15) * 16 + part * 256 with part = 2 ks + kh (ks = 32-byte K slice: +512 as an immediate).
    unsigned atab[PT + 2][3];
#pragma unroll
    for (int rr = 0; rr < PT + 2; rr++)
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const int hp = (PT * q + rr) * 34 + l31 + dx;
            atab[rr][dx] = (unsigned)((hp >> 4) * 1024 + (hp & 15) * 16 + kh * 256);
        }
    unsigned wlo, whi, wl;
    {
        const int n = tr_chan_of_row(l31);
        wl = (unsigned)(g * 4096 + (n >> 4) * 1024 + (n & 15) * 16 + kh * 256);
        wlo = WLO_OFF + wl; whi = WHI_OFF + wl;
    }
    // T2D: slot offsets of the chunk being read (wcur) and of the other one (wx = wcur ^ other: the swap is four s_xor)
    unsigned wcur[4], wx[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { wcur[t] = wslot_off(t); wx[t] = wslot_off(t) ^ wslot_off(4 + t); }
    f32x16_t acc[PT][2];

    // ---- prologue: halo tile of chunk 0 of the first tile, weights of its clusters 0 .. 7
    decode(tile, x0, y0, b, nt);
    halo_src(x0, y0, b);
    wt = wbase(b, nt);
    StaticFor<HPW>::run([&](auto kc_) { constexpr int k = decltype(kc_)::value; dma_buf(hoff[k], rs, 0u, rfl(wm0 + k * 8192)); });
    StaticFor<WAHEAD>::run([&](auto cc) { constexpr int c = decltype(cc)::value; dma_lin(wvoff, wt + c * 8192ull, rfl(wm0 + wslot_off(c))); });
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (g == 1) asm volatile("s_barrier" ::: "memory");                 // group 1 runs one phase behind group 0
    unsigned long long wptr = wt + WAHEAD * 8192ull;                    // weight piece the next request reads (cluster c + WAHEAD)
    unsigned hm0 = wm0 + H1_OFF;                                        // this wave's first piece of the halo buffer being FILLED

    // One chunk = 9 clusters; ONE instance of this body in the kernel (copies of it at the merge points of the tile loop cost the
    // register allocator 250-600 spilled registers).  relaxed: first chunk after an epilogue with all NSTORE stores in the queue;
    // final: the workgroup's last chunk (no closing barrier).  wnext0: the piece requested after this chunk's tap-0 request (in a
    // tile's last chunk the stream jumps to the next tile's image); hsoff: channel byte offset of the halo tile requested here.
    auto chunk = [&](int relaxed, bool final, unsigned long long wnext0, unsigned hsoff) {
        StaticFor<NTAP>::run([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int dy = T2D ? 1 + t / 2 : t / 3, dx = T2D ? 1 + t % 2 : t % 3;
            // ------------------------------------------------ LOAD phase of cluster c0 + t
            stamp(1);
            uint4 xa[PT][2], wa[2][2];
#pragma unroll
            for (int i = 0; i < PT; i++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) xa[i][ks] = *(const uint4*)(lds + atab[i + dy][dx] + ks * 512);
            const unsigned wrd = T2D ? wl + wcur[t] : (t < 3 ? wlo + t * 8192 : whi + (t - 3) * 8192);
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) wa[j][ks] = *(const uint4*)(lds + wrd + j * 2048 + ks * 512);
            __builtin_amdgcn_sched_barrier(0);
            // the weights of the NEXT cluster (and, at the last tap, the next halo tile) have landed - for this wave's pieces;
            // the barrier makes it true for everybody's.  lgkmcnt(0): this phase's reads are done before anyone overwrites.
            constexpr int VM_N = t == NTAP - 1 ? VM_T8 : VM_B;
            if (t < RLX && DG && relaxed == NSTORE + 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(VM_B + NSTORE + 4) : "memory");
            else if (t < RLX && relaxed) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(VM_B + NSTORE) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(VM_N) : "memory");
            stamp(2);
            if (!(t == NTAP - 1 && final && g == 1)) asm volatile("s_barrier" ::: "memory");
            stamp(3);
            __builtin_amdgcn_sched_barrier(0);
            // ------------------------------------------------ COMPUTE phase: 16 MFMAs, the cluster's DMA requests in their shadow
            // halo pieces of the tile the next chunk reads: 9 taps - piece t behind MFMA 3 of taps 0 .. HPW-1; T2D - pieces 2t, 2t + 1
            // behind MFMAs 3 and 11.  Weight piece of cluster c + WAHEAD behind MFMA 7, into the slot cluster c - 1 was read from.
            // (macros, not lambdas: clang rejects the captures of a generic lambda nested this deep as inline-asm operands)
#define PP_HALO_REQ(HPI)                                                                                                            \
            if (!(DBG && (dbg & 4)))                                                                                                \
                asm volatile("s_add_u32 m0, %1, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"                         \
                             :: "v"(hoff[(HPI) < HPW ? (HPI) : 0]), "s"(hm0), "s"(rs), "s"(hsoff), "i"((HPI) * 8192) : "memory", "scc")
#define PP_W_REQ()                                                                                                                  \
            do {                                                                                                                    \
                if (!(DBG && (dbg & 1))) {                                                                                          \
                    if constexpr (T2D) {                                                                                            \
                        const unsigned tgt = t == 0 ? (wcur[3] ^ wx[3]) : wcur[t == 0 ? 0 : t - 1];                                 \
                        asm volatile("s_add_u32 m0, %1, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"                           \
                                     :: "v"(wvoff), "s"(wm0), "s"(wptr), "s"(tgt) : "memory", "scc");                               \
                    } else {                                                                                                        \
                        asm volatile("s_add_u32 m0, %1, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"                           \
                                     :: "v"(wvoff), "s"(wm0), "s"(wptr), "i"(wslot_off((t + 8) % 9)) : "memory", "scc");            \
                    }                                                                                                               \
                }                                                                                                                   \
                wptr = t == 0 ? wnext0 : wptr + 8192ull;                                                                            \
            } while (0)
            constexpr int HP0 = T2D ? 2 * t : t, HP1 = T2D ? 2 * t + 1 : HPW;          // (HPW = none)
            if (!(DBG && (dbg & 2))) {
                __builtin_amdgcn_s_setprio(1);
                StaticFor<16>::run([&](auto mc) {
                    constexpr int m = decltype(mc)::value, ks = m >> 3, i = (m >> 1) & (PT - 1), j = m & 1;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&wa[j][ks], *(const bf16x8_t*)&xa[i][ks],
                                                                        acc[i][j], 0, 0, 0);
                    if constexpr (m == 3 && HP0 < HPW) {
                        __builtin_amdgcn_sched_barrier(0);
                        PP_HALO_REQ(HP0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (m == 11 && HP1 < HPW) {
                        __builtin_amdgcn_sched_barrier(0);
                        PP_HALO_REQ(HP1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (m == 7) {
                        __builtin_amdgcn_sched_barrier(0);
                        PP_W_REQ();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                __builtin_amdgcn_s_setprio(0);
            } else {
                if constexpr (HP0 < HPW) { PP_HALO_REQ(HP0); }
                PP_W_REQ();
                if constexpr (HP1 < HPW) { PP_HALO_REQ(HP1); }
#pragma unroll
                for (int i = 0; i < PT; i++)
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) asm volatile("" :: "v"(xa[i][ks].x), "v"(xa[i][ks].w), "v"(wa[i & 1][ks].x), "v"(wa[i & 1][ks].w));
            }
#undef PP_HALO_REQ
#undef PP_W_REQ
            __builtin_amdgcn_sched_barrier(0);
            stamp(4);
            if (!(t == NTAP - 1 && final)) asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (T2D) {
#pragma unroll
            for (int t = 0; t < 4; t++) wcur[t] ^= wx[t];
        }
        hm0 ^= (unsigned)H1_OFF;                   // the next chunk reads the buffer just filled and fills the other one
#pragma unroll
        for (int rr = 0; rr < PT + 2; rr++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) atab[rr][dx] ^= (unsigned)H1_OFF;
    };

    // Additive terms of the epilogue (noise * strength + bias, stylegan2_generator.py:911-920, times the activation gain) are the
    // INITIAL value of the accumulators: their loads are issued at the start of the previous tile's epilogue and consumed at its
    // end, so no memory latency sits between a tile's last MFMA and its stores, and the epilogue is lrelu + pack + store.
    struct AddT { float ab[2][16], anw[NWC ? 2 : 1][NWC ? 16 : 1], anz[PT]; };
    auto add_load = [&](AddT& A, int tx0, int ty0, int tb, int tnt) {
        float (&ab)[2][16] = A.ab; float (&anw)[NWC ? 2 : 1][NWC ? 16 : 1] = A.anw; float (&anz)[PT] = A.anz;
        const auto p = P();
        const float bg = p->bias_scale * p->gain;
        const int gx = tx0 + l31;
#pragma unroll
        for (int i = 0; i < PT; i++) {
            const int gy = ty0 + PT * q + i;
            anz[i] = (p->noise && gy < p->H && gx < p->W) ? p->noise[(size_t)tb * p->noise_bstride + gy * p->W + gx] : 0.f;
        }
        if constexpr (!NWC) anw[0][0] = p->noise ? p->noise_w[0] * p->gain : 0.f;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int e4 = 0; e4 < 2; e4++) {
                    const int c0 = tnt * 128 + g * 64 + j * 32 + 8 * kh + 16 * h + 4 * e4, r = 8 * h + 4 * e4;
                    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p->bias) b4 = *(const float4*)(p->bias + c0);
                    ab[j][r] = b4.x * bg; ab[j][r + 1] = b4.y * bg; ab[j][r + 2] = b4.z * bg; ab[j][r + 3] = b4.w * bg;
                    if constexpr (NWC) {
                        float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (p->noise) n4 = *(const float4*)(p->noise_w + c0);
                        anw[j][r] = n4.x * p->gain; anw[j][r + 1] = n4.y * p->gain; anw[j][r + 2] = n4.z * p->gain; anw[j][r + 3] = n4.w * p->gain;
                    }
                }
    };
    auto add_apply = [&](const AddT& A) {
#pragma unroll
        for (int i = 0; i < PT; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = fmaf(A.anw[NWC ? j : 0][NWC ? r : 0], A.anz[i], A.ab[j][r]);
    };
    if constexpr (DG) {
#pragma unroll
        for (int i = 0; i < PT; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    } else { AddT A; add_load(A, x0, y0, b, nt); add_apply(A); }

    // epilogue staging: lanes write (pixel, 16-byte piece) of 16 pixels x 32 channels, read back lane-linear = 4 lanes per pixel
    const unsigned est_w = PP_LDS + wave * 1024 + (l31 & 15) * 64 + kh * 16, est_r = PP_LDS + wave * 1024 + lane * 16;

    int relaxed = 0;                            // memory operations of this wave's last epilogue that may still be in the queue (NSTORE stores, + 4 atomics)
    for (;;) {
        const int next = tile + stride;
        const bool has_next = next < tile_end;
        int nx0 = 0, ny0 = 0, nb_ = 0, nnt = 0;
        unsigned long long nwt = 0;
        for (int kc = 0; kc < nchunks; kc++) {
            const bool lastc = kc == nchunks - 1;
            // the chunk whose tap-0 request is the last one into this tile's weight image (9 taps: the request of tap t is for tap
            // t - 1 of the next chunk; T2D: of the chunk after the next)
            const bool wjump = T2D ? kc == nchunks - 2 : lastc;
            unsigned long long wnext0 = wptr + 8192ull;
            unsigned hsoff = (unsigned)(kc + 1) * 64u;
            if (P()->s2d) {
                const auto p = P();
                const int ph = (kc + 1) / p->cpp, wi = (kc + 1) - ph * p->cpp;
                hsoff = (unsigned)((((ph >> 1) * p->SW + (ph & 1)) * p->SC + wi * 32) * 2);
            }
            if (wjump) {                       // the weight stream runs on into the next tile's image (none: dummies from its own)
                if (has_next) { decode(next, nx0, ny0, nb_, nnt); nwt = wbase(nb_, nnt); }
                wnext0 = has_next ? nwt : wt;
            }
            if (lastc) {                       // ... and the halo requests into its first halo tile (none: dummies, slots nobody reads)
                if (has_next) halo_src(nx0, ny0, nb_);
                hsoff = 0u;
            }
            chunk(kc == 0 ? relaxed : 0, lastc && !has_next, wnext0, hsoff);
        }
        if constexpr (DG) {
            // ---------------------------------------------------- data-gradient epilogue (conv_epilogue.h MODE 1 / 2 on this tile form)
            // The raw f32 accumulators go through LDS one MFMA block (32 pixels x 32 channels) at a time (the halo buffer the last chunk
            // read is free until the next tile's first cluster requests chunk 1 into it: 5 KiB per wave, pixel pitch 144 B) and come
            // back in two rounds of 16 pixels: a lane holds 8 consecutive channels of one pixel = the 16 bytes of dot_src / addend / y
            // it loads and stores (64-byte runs per pixel).  Rounds run (channel block j, row i, half r); the requests of round
            // R + DEP are issued while round R is worked on.  No branch inside the rounds (the flavour is a template parameter, rows
            // and columns outside the image are out-of-range offsets of the buffer descriptors: loads return zeros, stores are
            // dropped, and the accumulators of such pixels are zeroed up front so that they stay out of the sums): every wave issues
            // the same NSTORE stores, and the compiler's wait counts for the prefetched operands stay exact.
            // Per-channel sums: 32 per lane and channel block (4 sums x 8 channels), reduce-scattered over the 16 pixel lanes (30
            // exchange-adds) so that TWO atomic instructions with 64 distinct addresses flush a block.  The arithmetic is written on
            // float pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): the seam is bound by VALU issue, two waves per SIMD.
            typedef float v2f __attribute__((ext_vector_type(2)));
            stamp(5);
            if (has_next) { if (g == 0) asm volatile("s_barrier" ::: "memory"); }
            else asm volatile("s_barrier" ::: "memory");           // (last tile: group 1 has read the halo buffer before anyone scribbles on it)
            const auto p = P();
            const int H_ = p->H, W_ = p->W, Co = p->Cout, cb2 = Co * 2;
            const unsigned ybytes = (unsigned)(H_ * W_) * (unsigned)cb2;
            const size_t boff = (size_t)b * H_ * W_ * Co;
            bf16_t* ybase = p->y + boff;
            const bool has_dot = p->dot != nullptr;
            const bool has_nz = PREP && p->prep_noise && p->prep_ns;
            const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(ybase, 0, ybytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(has_dot ? (bf16_t*)p->dot + boff : ybase, 0, has_dot ? ybytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(ADD ? (bf16_t*)p->add + boff : ybase, 0, ADD ? ybytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(
                has_nz ? (float*)p->prep_noise + (size_t)b * p->prep_noise_bstride : (float*)ybase, 0, has_nz ? (unsigned)(H_ * W_ * 4) : 0, 0x00020000);
            const float pg = p->prep_gain, pz = 1.f / p->prep_gain;
            const float pns = has_nz ? p->prep_ns[0] : 0.f;
            const float add_scale = p->add_scale;
            const float* osc = p->out_scale ? p->out_scale + (size_t)b * Co : nullptr;
            float* const t_st = p->stats ? p->stats + ((size_t)(tile % p->stats_slots) * p->B + b) * Co * 2 : nullptr;
            float* const t_pr = (PREP && p->prep_stats) ? p->prep_stats + ((size_t)(tile % p->stats_slots) * p->B + b) * Co * 2 : nullptr;
            // byte offsets of this lane's pixel (first row of the wave's strip) + its 16 bytes of the 64-channel group; the row (and
            // the channel block) is added per round, so a row below the image is past the descriptor's end by itself
            unsigned voff[2], nvoff[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int px = 16 * r + (lane >> 2);
                const bool ok = x0 + px < W_;
                voff[r] = ok ? (unsigned)(((y0 + PT * q) * W_ + x0 + px) * cb2 + (nt * 128 + g * 64) * 2 + (lane & 3) * 16) : 0x80000000u;
                nvoff[r] = ok ? (unsigned)(((y0 + PT * q) * W_ + x0 + px) * 4) : 0x80000000u;
            }
            if (y0 + TH > H_ || x0 + 32 > W_) {
                // ragged tile: the accumulators of pixels outside the image (computed from the zero-padded halo, not zero) must stay
                // out of the sums
                const bool cok = x0 + l31 < W_;
                const int rows_ok = H_ - (y0 + PT * q);
#pragma unroll
                for (int i = 0; i < PT; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[i][j][r] = (cok && i < rows_ok) ? acc[i][j][r] : 0.f;
            }
            const unsigned stg = (hm0 - wm0) + (unsigned)wave * 5120u;
            const unsigned st_w = stg + l31 * 144 + kh * 32, st_r = stg + (lane >> 2) * 144 + (lane & 3) * 32;
            const int cbase = nt * 128 + g * 64 + (lane & 3) * 8;            // + j*32: first of this lane's 8 channels on the read side
            const unsigned rowb = (unsigned)(W_ * cb2), rown = (unsigned)(W_ * 4);
            constexpr int DEP = 3;                       // rounds the requests run ahead (a round is ~0.7 us, an HBM miss 1.5-2.5)
            u32x4_t dq[DEP], aq[ADD ? DEP : 1];
            float nq[PREP ? DEP : 1];
            auto issue = [&](auto Rc) {
                constexpr int R = decltype(Rc)::value, j = R >> 3, i = (R >> 1) & 3, r = R & 1, sl = R % DEP;
                if (DBG && (dbg & 1024)) return;
                const unsigned vo = voff[r] + (unsigned)i * rowb;
                dq[sl] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, vo, j * 64, 0);
                if constexpr (ADD) aq[sl] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, vo, j * 64, 0);
                if constexpr (PREP) nq[sl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_n, nvoff[r] + (unsigned)i * rown, 0, 0));
            };
            StaticFor<DEP>::run([&](auto Rc) { issue(Rc); });
            v2f posc[4], ps0[4], ps1[4], pt0[4], pt1[4];
            StaticFor<16>::run([&](auto Rc) {
                constexpr int R = decltype(Rc)::value, j = R >> 3, i = (R >> 1) & 3, r = R & 1, sl = R % DEP;
                if constexpr ((R & 7) == 0) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { ps0[e] = 0.f; ps1[e] = 0.f; pt0[e] = 0.f; pt1[e] = 0.f; posc[e] = 1.f; }
                    if (osc) {
                        const float4 s0 = *(const float4*)(osc + cbase + j * 32), s1 = *(const float4*)(osc + cbase + j * 32 + 4);
                        posc[0] = v2f{s0.x, s0.y}; posc[1] = v2f{s0.z, s0.w}; posc[2] = v2f{s1.x, s1.y}; posc[3] = v2f{s1.z, s1.w};
                    }
                }
                if constexpr (r == 0) {
                    // all 64 lanes park the block (32 pixels x 32 channels, f32) - both halves r read it back
                    const f32x16_t a = acc[i][j];
#pragma unroll
                    for (int h = 0; h < 2; h++)
#pragma unroll
                        for (int e4 = 0; e4 < 2; e4++)
                            *(float4*)(lds + st_w + h * 64 + e4 * 16) = make_float4(a[8 * h + 4 * e4], a[8 * h + 4 * e4 + 1], a[8 * h + 4 * e4 + 2], a[8 * h + 4 * e4 + 3]);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                v2f f[4];
                {
                    const float4 f0 = *(const float4*)(lds + st_r + r * 2304), f1 = *(const float4*)(lds + st_r + r * 2304 + 16);
                    f[0] = v2f{f0.x, f0.y}; f[1] = v2f{f0.z, f0.w}; f[2] = v2f{f1.x, f1.y}; f[3] = v2f{f1.z, f1.w};
                }
                if constexpr (r == 1) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                const u32x4_t dv = dq[sl], av = aq[ADD ? sl : 0];
                const float nzs = PREP ? pns * nq[PREP ? sl : 0] : 0.f;
                if constexpr (R + DEP < 16) issue(std::integral_constant<int, R + DEP>{});
                v2f d[4];
#pragma unroll
                for (int e = 0; e < 4; e++) d[e] = v2f{__uint_as_float(dv[e] << 16), __uint_as_float(dv[e] & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if constexpr (!MASK) ps0[e] = __builtin_elementwise_fma(f[e], d[e], ps0[e]);
                    if constexpr (!MASK && !PREP) ps1[e] += f[e];           // (the synthesis chain has no use for the plain sum)
                    f[e] *= posc[e];
                }
                if constexpr (ADD) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const v2f ad = v2f{__uint_as_float(av[e] << 16), __uint_as_float(av[e] & 0xffff0000u)};
                        f[e] = __builtin_elementwise_fma(v2f{add_scale, add_scale}, ad, f[e]);
                    }
                }
                if constexpr (MASK) {
#pragma unroll
                    for (int e = 0; e < 4; e++) { f[e].x = d[e].x > 0.f ? f[e].x : 0.f; f[e].y = d[e].y > 0.f ? f[e].y : 0.f; }
                }
                if constexpr (PREP) {
                    if (!(DBG && (dbg & 2048))) {
                        // lrelu'(x) and the inverse activation from the stored x = lrelu(z)*gain: factor pairs (1, 1) for x > 0, (0.2, 5) else
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const bool p0 = d[e].x > 0.f, p1 = d[e].y > 0.f;
                            const v2f sf = v2f{p0 ? pg : 0.2f * pg, p1 ? pg : 0.2f * pg}, zf = v2f{p0 ? pz : 5.f * pz, p1 ? pz : 5.f * pz};
                            const v2f gz = f[e] * sf;
                            const v2f zt = __builtin_elementwise_fma(d[e], zf, v2f{-nzs, -nzs});
                            pt0[e] = __builtin_elementwise_fma(gz, zt, pt0[e]);
                            pt1[e] += gz;
                            f[e] = gz;
                            __builtin_amdgcn_sched_barrier(0);       // (keeps the compare masks and gz / zt of one pair short-lived)
                        }
                    }
                }
                u32x4_t o;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = pack2bf(f[e].x, f[e].y);
                if (DBG && (dbg & 128)) asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
                else __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, voff[r] + (unsigned)i * rowb, j * 64, 0);
                // One basic block: instruction selection is free to sink the sum updates down to the flush (keeping every round's
                // operands alive: 150 spilled registers) - the empty asm statements pin them to their round
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if constexpr (!MASK) asm volatile("" : "+v"(ps0[e]));
                    if constexpr (!MASK && !PREP) asm volatile("" : "+v"(ps1[e]));
                    if constexpr (PREP) asm volatile("" : "+v"(pt0[e]), "+v"(pt1[e]));
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr ((R & 7) == 7 && !MASK) {
                    if ((t_st || t_pr) && !(DBG && (dbg & 512))) {
                        float v[32];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            v[2 * e] = ps0[e].x; v[2 * e + 1] = ps0[e].y; v[8 + 2 * e] = ps1[e].x; v[9 + 2 * e] = ps1[e].y;
                            v[16 + 2 * e] = pt0[e].x; v[17 + 2 * e] = pt0[e].y; v[24 + 2 * e] = pt1[e].x; v[25 + 2 * e] = pt1[e].y;
                        }
                        StaticFor<4>::run([&](auto Lc) {
                            constexpr int L = decltype(Lc)::value, half = 16 >> L;
                            const bool up = (lane >> (2 + L)) & 1;
#pragma unroll
                            for (int k = 0; k < half; k++) {
                                const float keep = up ? v[half + k] : v[k], send = up ? v[k] : v[half + k];
                                v[k] = keep + __shfl_xor(send, 4 << L, 64);
                            }
                        });
                        // this lane's two sums: index (b2 b3 b4 b5 k) of (sum s = idx >> 3, channel e = idx & 7), bN = lane bit N
                        const int sidx = (((lane >> 2) & 1) << 1) | ((lane >> 3) & 1);
                        const int e0 = (((lane >> 4) & 1) << 2) | (((lane >> 5) & 1) << 1);
                        float* tab = (sidx & 2) ? t_pr : t_st;
                        if (tab) {
                            float* dst = tab + (cbase + j * 32 + e0) * 2 + (sidx & 1);
                            atomicAdd(dst, v[0]);
                            atomicAdd(dst + 2, v[1]);
                        }
                    }
                }
            });
#pragma unroll
            for (int i = 0; i < PT; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
            // the relaxed counts of the next tile's first chunk: NSTORE stores, + 4 atomic instructions when a statistics table
            // exists (lanes whose table is null are masked off; the instruction is issued for the others)
            relaxed = (DBG && (dbg & (128 | 512 | 16))) ? 0 : ((!MASK && (t_st || t_pr)) ? NSTORE + 4 : NSTORE);
            if (has_next && g == 1) asm volatile("s_barrier" ::: "memory");
            stamp(6);
            if (!has_next) break;
            tile = next; x0 = nx0; y0 = ny0; b = nb_; nt = nnt; wt = nwt;
            continue;
        }
        // ------------------------------------------------------------ epilogue: activation, pack, 64-byte runs through LDS
        stamp(5);
        if (DBG && (dbg & 8)) {
#pragma unroll
            for (int i = 0; i < PT; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) asm volatile("" :: "v"(acc[i][j][0]), "v"(acc[i][j][15]));
        } else {
            AddT A;                              // (consumed after the stores below; at the last tile: a second copy of its own, unused)
            add_load(A, has_next ? nx0 : x0, has_next ? ny0 : y0, has_next ? nb_ : b, has_next ? nnt : nt);
            // Seam choreography: group 0 lets group 1 finish its last cluster (one extra barrier here), then BOTH groups run their
            // epilogues in the same phase - two waves per SIMD interleave their LDS / store latencies - and group 1 adds its extra
            // barrier behind the epilogue, which puts it one phase behind group 0 again.  (Serial: 2 x 8.4 k cycles per seam.)
            if (has_next && g == 0) asm volatile("s_barrier" ::: "memory");
            const auto p = P();
            const float slope = p->act == DGE_ACT_LRELU ? 0.2f : (p->act == DGE_ACT_RELU ? 0.f : 1.f);
            const int cb2 = p->Cout * 2;
            // Stores go through a descriptor of the sample's output image: out-of-image lanes carry an out-of-range offset and are
            // dropped by the hardware, rows below the image use an empty descriptor - every wave issues exactly NSTORE stores
            // (the wait counts of the next tile's first chunk rely on it).  A lane's accumulators are two 16-byte runs of one
            // pixel; stored as they are, one instruction touches 32 lines with 32 bytes each and the store path pays per line.
            const unsigned ybytes = (unsigned)(p->H * p->W) * (unsigned)cb2;
            bf16_t* ybase = p->y + (size_t)b * p->H * p->W * p->Cout;
            const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(ybase, 0, ybytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_0 = __builtin_amdgcn_make_buffer_rsrc(ybase, 0, 0, 0x00020000);
            unsigned voff[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int px = 16 * r + (lane >> 2);
                voff[r] = (x0 + px < p->W) ? (unsigned)(px * cb2 + (lane & 3) * 16) : 0x80000000u;
            }
#pragma unroll
            for (int i = 0; i < PT; i++) {
                const int gy = y0 + PT * q + i;
                const unsigned rowoff = (unsigned)((gy * p->W + x0) * cb2 + (nt * 128 + g * 64) * 2);
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const f32x16_t a = acc[i][j];
                    float v0[8], v1[8];
#pragma unroll
                    for (int r = 0; r < 8; r++) { v0[r] = fmaxf(a[r], a[r] * slope); v1[r] = fmaxf(a[8 + r], a[8 + r] * slope); }
                    const uint4 o0 = pack16(v0, (bf16_t*)nullptr), o1 = pack16(v1, (bf16_t*)nullptr);
                    if (DBG && (dbg & 128)) { asm volatile("" :: "v"(o0.x), "v"(o0.w), "v"(o1.x), "v"(o1.w)); continue; }
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        if ((l31 >> 4) == r) {
                            *(uint4*)(lds + est_w) = o0;
                            *(uint4*)(lds + est_w + 32) = o1;
                        }
                        // lanes exchange data through LDS inside one wave: without the wave-level fence + barrier the compiler may
                        // (and did) run the read of the lanes that do not write in this round ahead of the other lanes' writes
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const u32x4_t o = *(const u32x4_t*)(lds + est_r);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        if (gy < p->H) __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, voff[r], rowoff + j * 64, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(o, rs_0, voff[r], 0, 0);
                    }
                }
            }
            relaxed = (DBG && ((dbg & 16) || (dbg & 128))) ? 0 : NSTORE;
            add_apply(A);
            if (has_next && g == 1) asm volatile("s_barrier" ::: "memory");
        }
        stamp(6);
        if (!has_next) break;
        tile = next; x0 = nx0; y0 = ny0; b = nb_; nt = nnt; wt = nwt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the dummy requests of the last chunk target this workgroup's LDS
}

// LDS image of the weights, one 8 KiB block per (sample, N tile, K chunk, tap): 8 pieces of 16 rows, part-major inside a piece.
// mode 0: GEMM (n, k) = (out channel, in channel), tap as stored.  mode 1 (data gradient): (n, k) = (in channel, out channel) of
// the forward weight, taps flipped.  in_scale [nb][K] / out_scale [nb][N] (either may be null) and `gain` are folded:
// W' = bf16((w*wscale) * (in_scale[k] * (gain*out_scale[n])))  - the association oracle/conv_ref.py:modconv_folded uses.
// One workgroup per (N tile, K chunk, 16-row piece): its 16 x 32 x 9 source weights are read once, coalesced, into LDS and
// written out as the nine 1 KiB pieces of every sample's image.
__global__ __launch_bounds__(256) void conv_pp_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int N, int K, float wscale,
                                    const float* __restrict__ in_scale, const float* __restrict__ out_scale, float gain, int nb, int mode,
                                    int src_rows, int in_period, int ntap) {
    __shared__ __attribute__((aligned(16))) float wl[16][9][32 + 4];         // [row][tap][k] (+4: the row / tap strided reads below)
    const int nchunks = K / 32, ntn = N / 128;
    int bid = blockIdx.x;
    const int pc = bid % 8; bid /= 8;
    const int kc = bid % nchunks;
    const int nt = bid / nchunks;
    const int n0 = nt * 128 + pc * 16, k0 = kc * 32;
    const int tid = threadIdx.x;
    if (mode == 0) {                       // w[n][k][tap]: per row 288 contiguous floats
        for (int idx = tid; idx < 16 * 288; idx += 256) {
            const int r = idx / 288, e = idx - r * 288;
            wl[r][e % 9][e / 9] = w[((size_t)(n0 + r) * K + k0) * 9 + e] * wscale;
        }
    } else if (mode == 1) {                // w[k][n][8 - tap]: per k 144 contiguous floats
        for (int idx = tid; idx < 32 * 144; idx += 256) {
            const int k = idx / 144, e = idx - k * 144;
            wl[e / 9][8 - e % 9][k] = w[((size_t)(k0 + k) * N + n0) * 9 + e] * wscale;
        }
    } else {                               // rows of a dge_pack_conv_weight copy in f32, [tap][src_rows][K]: 32 contiguous floats per (tap, row)
        for (int idx = tid; idx < 9 * 16 * 32; idx += 256) {
            const int k = idx & 31, r = (idx >> 5) & 15, t = idx >> 9;
            wl[r][t][k] = w[((size_t)t * src_rows + n0 + r) * K + k0 + k] * wscale;
        }
    }
    __syncthreads();
    // thread = (part qd, row r) of the piece, fixed over taps and samples: its 8 in-channel scales and its row's out-channel scale
    // are loaded once per sample; per (sample, tap) it reads 8 consecutive floats and writes one 16-byte group
    const int qd = (tid >> 4) & 3, r = tid & 15, tq = tid >> 6;          // taps tq, tq + 4, tq + 8
    for (int b = blockIdx.y; b < nb; b += gridDim.y) {
        float m[8];
        const float on = gain * (out_scale ? out_scale[(size_t)b * N + n0 + r] : 1.f);
#pragma unroll
        for (int j = 0; j < 8; j++) m[j] = (in_scale ? in_scale[(size_t)b * in_period + (k0 + qd * 8 + j) % in_period] : 1.f) * on;
        bf16_t* ob = out + ((((size_t)b * ntn + nt) * nchunks + kc) * ntap * 8 + pc) * 512 + (qd * 16 + r) * 8;   // (elements; tap stride 8 * 512)
#pragma unroll
        for (int ti = 0; ti < 3; ti++) {
            const int t = tq + 4 * ti;
            // ntap == 4 (phase-form adjoint, in_t2d): only the taps (dy, dx) in {1, 2}^2 exist: image tap tt <- source tap 4 + 3 (tt >> 1) + (tt & 1)
            const int ts = ntap == 4 ? 4 + 3 * (t >> 1) + (t & 1) : t;
            if (t < ntap) {
                const float4 a = *(const float4*)&wl[r][ts][qd * 8], c = *(const float4*)&wl[r][ts][qd * 8 + 4];
                const float v[8] = {a.x * m[0], a.y * m[1], a.z * m[2], a.w * m[3], c.x * m[4], c.y * m[5], c.z * m[6], c.w * m[7]};
                *(uint4*)(ob + (size_t)t * 8 * 512) = pack16(v, (bf16_t*)nullptr);
            }
        }
    }
}

}  // namespace

// development aid, not part of the C ABI (include/dge_hip.h): copies the stamps of the last DGE_CONV_DBG & 256 launch to the host
extern "C" int dge_dbg_pp_prof(long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pp_prof), sizeof(long long) * 2048, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

extern "C" int dge_conv_pp_supported(int B, int H, int W, int Cin, int Cout, int dtype) {
    static const bool off = getenv("DGE_NO_PP") != nullptr;          // read once: this is called several times per layer
    if (dtype != DGE_BF16 || off) return 0;
    if (Cin % 32 != 0 || Cout % 128 != 0 || Cin < 64) return 0;
    if ((long long)H * W * Cin * 2 >= (1ll << 31)) return 0;
    // one workgroup per CU and no overlap between tiles: the grid must fill the chip
    const long tiles = (long)B * ((H + 15) / 16) * ((W + 31) / 32) * (Cout / 128);
    // (measured, batch 16: 256 -> 256 at 48 x 64 = 192 tiles 58.4 -> 54.6 us, at 44^2 56.1 -> 51.8; 512 -> 512 at 32^2 = 128 tiles
    //  loses, 84.9 -> 91.5)
    static const int min_tiles = getenv("DGE_PP_MIN_TILES") ? atoi(getenv("DGE_PP_MIN_TILES")) : 192;
    return tiles >= min_tiles ? 1 : 0;
}

extern "C" int dge_pack_conv_pp(const float* w_oihw, void* out, int N, int K, float wscale, const float* in_scale, const float* out_scale,
                                float gain, int nb, int mode, hipStream_t s) {
    DGE_CHECK(w_oihw && out, "pack_conv_pp: null tensor");
    DGE_CHECK(N % 128 == 0 && K % 32 == 0 && nb >= 1 && (mode == 0 || mode == 1), "pack_conv_pp: N=%d must be a multiple of 128, K=%d of 32", N, K);
    DGE_CHECK(nb == 1 || in_scale || out_scale, "pack_conv_pp: per-sample copies need a per-sample scale");
    const long grid = (long)(N / 128) * (K / 32) * 8;
    hipLaunchKernelGGL(conv_pp_pack_kernel, dim3((unsigned)grid, (unsigned)(grid >= 256 ? (nb + 1) / 2 : nb)), dim3(256), 0, s, w_oihw, (bf16_t*)out, N, K, wscale, in_scale, out_scale, gain, nb, mode, 0, K, 9);
    DGE_LAUNCH_CHECK("pack_conv_pp");
    return 0;
}

extern "C" int dge_pack_conv_pp_rows(const float* w_rows, int src_rows, void* out, int N, int K, const float* in_scale, int in_period,
                                     const float* out_scale, float gain, int nb, int t2d, hipStream_t s) {
    DGE_CHECK(w_rows && out, "pack_conv_pp_rows: null tensor");
    DGE_CHECK(N % 128 == 0 && K % 32 == 0 && nb >= 1 && src_rows >= N, "pack_conv_pp_rows: N=%d must be a multiple of 128 (<= src_rows=%d), K=%d of 32", N, src_rows, K);
    DGE_CHECK(nb == 1 || in_scale || out_scale, "pack_conv_pp_rows: per-sample copies need a per-sample scale");
    DGE_CHECK(!in_scale || (in_period >= 1 && K % in_period == 0), "pack_conv_pp_rows: in_period=%d must divide K=%d", in_period, K);
    const long grid = (long)(N / 128) * (K / 32) * 8;
    hipLaunchKernelGGL(conv_pp_pack_kernel, dim3((unsigned)grid, (unsigned)(grid >= 256 ? (nb + 1) / 2 : nb)), dim3(256), 0, s, w_rows, (bf16_t*)out, N, K, 1.f, in_scale, out_scale, gain, nb, 2,
                       src_rows, in_scale ? in_period : K, t2d ? 4 : 9);
    DGE_LAUNCH_CHECK("pack_conv_pp_rows");
    return 0;
}

extern "C" int dge_conv_pp(const dge_conv_pp_desc* d, hipStream_t s) {
    DGE_CHECK(d && d->x && d->w_pp && d->y, "conv_pp: null tensor");
    DGE_CHECK(dge_conv_pp_supported(d->B, d->H, d->W, d->Cin, d->Cout, DGE_BF16), "conv_pp: %dx%d Cin=%d Cout=%d B=%d is not a shape dge_conv_pp_supported() accepts",
              d->H, d->W, d->Cin, d->Cout, d->B);
    DGE_CHECK(!d->noise || d->noise_w, "conv_pp: noise needs its weight");
    DGE_CHECK(d->gain > 0.f, "conv_pp: the gain is folded into scale / noise / bias and must be positive");
    PPParams p;
    p.x = (const bf16_t*)d->x; p.w = (const bf16_t*)d->w_pp; p.y = (bf16_t*)d->y;
    p.w_bstride = d->w_bstride * 2;
    p.bias = d->bias; p.noise = d->noise; p.noise_w = d->noise_w;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.noise_bstride = d->noise_batch > 1 ? d->H * d->W : 0; p.noise_w_stride = d->noise_w_per_channel ? 1 : 0; p.act = d->act;
    p.bias_scale = d->bias_scale; p.gain = d->gain;
    p.tiles_x = (d->W + 31) / 32; p.tiles_y = (d->H + 15) / 16; p.ntn = d->Cout / 128; p.nchunks = d->Cin / 32;
    p.dbg = dge_env().conv_dbg;
    p.s2d = d->in_s2d ? 1 : 0;
    p.VH = d->H; p.VW = d->W;
    DGE_CHECK(!(d->in_s2d && d->in_t2d), "conv_pp: in_s2d and in_t2d exclude each other");
    if (d->in_t2d) {
        DGE_CHECK(d->dgrad && d->prep, "conv_pp: in_t2d comes with the data-gradient form and its fused tail backward");
        DGE_CHECK(d->Cin >= 64, "conv_pp: in_t2d needs at least two K chunks");
        p.VH = d->H + 1; p.VW = d->W + 1; p.SW = d->W + 1; p.SC = d->Cin; p.cpp = p.nchunks;
        p.x_bstride = (long long)(d->H + 1) * (d->W + 1) * d->Cin * 2;
    } else if (p.s2d) {
        DGE_CHECK(d->dgrad, "conv_pp: in_s2d comes with the data-gradient form");
        DGE_CHECK(d->Cin % 128 == 0, "conv_pp: in_s2d needs Cin / 4 = %d channels per phase in whole 32-channel chunks", d->Cin / 4);
        p.SW = 2 * d->W; p.SC = d->Cin / 4; p.cpp = p.SC / 32;
    } else { p.SW = d->W; p.SC = d->Cin; p.cpp = p.nchunks; }
    if (!d->in_t2d) p.x_bstride = (long long)d->H * d->W * d->Cin * 2; // (space-to-depth: 2H x 2W x Cin/4 - the same bytes)
    p.out_scale = nullptr; p.dot = nullptr; p.add = nullptr; p.add_scale = 0.f; p.stats = nullptr; p.prep_stats = nullptr; p.stats_slots = 1;
    p.prep = 0; p.mask_relu = 0; p.prep_noise_bstride = 0; p.prep_gain = 1.f; p.prep_noise = nullptr; p.prep_ns = nullptr;
    if (d->dgrad) {
        DGE_CHECK(!d->bias && !d->noise && d->act == DGE_ACT_NONE, "conv_pp: the data-gradient form has no bias / noise / activation");
        DGE_CHECK(!((d->stats || d->prep) && dge_get_deterministic()), "conv_pp: the statistics of the data-gradient form are f32 atomics: not offered in deterministic mode (run dge_conv2d)");
        DGE_CHECK(!d->prep || (d->dot_src && d->prep_stats && d->prep_gain > 0.f), "conv_pp: prep needs dot_src, prep_stats and a positive gain");
        DGE_CHECK(!d->mask_relu || (d->dot_src && !d->prep), "conv_pp: mask_relu needs dot_src and excludes prep");
        DGE_CHECK(!d->stats || d->dot_src, "conv_pp: statistics are (sum a*dot_src, sum a): dot_src missing");
        DGE_CHECK(d->stats_slots >= 1 || (!d->stats && !d->prep_stats), "conv_pp: stats_slots");
        p.out_scale = d->out_scale; p.dot = (const bf16_t*)d->dot_src; p.add = (const bf16_t*)d->addend; p.add_scale = d->add_scale;
        p.stats = d->stats; p.prep_stats = d->prep ? d->prep_stats : nullptr; p.stats_slots = d->stats_slots >= 1 ? d->stats_slots : 1;
        p.prep = d->prep ? 1 : 0; p.mask_relu = d->mask_relu ? 1 : 0; p.prep_gain = d->prep ? d->prep_gain : 1.f;
        p.prep_noise = d->prep ? d->prep_noise : nullptr; p.prep_ns = d->prep ? d->prep_ns : nullptr;
        p.prep_noise_bstride = d->prep_noise_batch > 1 ? d->H * d->W : 0;
    } else {
        DGE_CHECK(!d->out_scale, "conv_pp: a per-sample output scale is folded into the weight image (dge_pack_conv_pp), not applied by the launch");
        DGE_CHECK(!d->dot_src && !d->addend && !d->stats && !d->prep && !d->mask_relu, "conv_pp: dot_src / addend / stats / prep / mask_relu belong to the data-gradient form (dgrad = 1)");
    }
    const long tiles = (long)p.tiles_x * p.tiles_y * p.B * p.ntn;
    // one workgroup per CU (160 KB of LDS each), 32 per XCD; each walks its share of its XCD's tile range
    int cus = 256;
    {   // CU count per device, queried once (hipGetDeviceProperties fills a multi-KB struct: not on the launch path)
        static int cu_of_dev[64] = {0};
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            if (cu_of_dev[dev] == 0) {
                int n = 0;
                cu_of_dev[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8) ? n / 8 * 8 : 256;
            }
            cus = cu_of_dev[dev];
        }
    }
    const long grid = tiles < cus ? (tiles + 7) / 8 * 8 : cus;
    dge_note_kernel(d->in_t2d ? "conv_pp<bf16,16,32,128>+dg+t2d+prep" : d->dgrad ? (d->in_s2d ? (d->prep ? "conv_pp<bf16,16,32,128>+dg+s2d+prep" : "conv_pp<bf16,16,32,128>+dg+s2d")
                                          : (d->prep ? "conv_pp<bf16,16,32,128>+dg+prep" : (d->mask_relu ? "conv_pp<bf16,16,32,128>+dg+mask" : "conv_pp<bf16,16,32,128>+dg")))
                             : "conv_pp<bf16,16,32,128>");
    // EPI 1: noise weight per channel (model/E/E.py:60-62) instead of StyleGAN2's scalar strength; EPI 2: data gradient
    const int epi = d->dgrad ? 2 + 2 * (d->prep ? 0 : (d->mask_relu ? 2 : 1)) + (d->addend ? 1 : 0) : (p.noise_w_stride ? 1 : 0);
#define PP_GO(E) do { if (p.dbg) hipLaunchKernelGGL((conv_pp_kernel<4, true, E>), dim3((unsigned)grid), dim3(512), 0, s, p); \
                      else hipLaunchKernelGGL((conv_pp_kernel<4, false, E>), dim3((unsigned)grid), dim3(512), 0, s, p); } while (0)
    if (d->in_t2d) {
        if (p.dbg) { if (d->addend) hipLaunchKernelGGL((conv_pp_kernel<4, true, 3, true>), dim3((unsigned)grid), dim3(512), 0, s, p);
                     else hipLaunchKernelGGL((conv_pp_kernel<4, true, 2, true>), dim3((unsigned)grid), dim3(512), 0, s, p); }
        else { if (d->addend) hipLaunchKernelGGL((conv_pp_kernel<4, false, 3, true>), dim3((unsigned)grid), dim3(512), 0, s, p);
               else hipLaunchKernelGGL((conv_pp_kernel<4, false, 2, true>), dim3((unsigned)grid), dim3(512), 0, s, p); }
    } else
    switch (epi) {
        case 0: PP_GO(0); break; case 1: PP_GO(1); break; case 2: PP_GO(2); break; case 3: PP_GO(3); break;
        case 4: PP_GO(4); break; case 5: PP_GO(5); break; case 6: PP_GO(6); break; default: PP_GO(7); break;
    }
#undef PP_GO
    DGE_LAUNCH_CHECK("conv_pp");
    return 0;
}
