// Streaming 3x3 convolution for the HBM-bound layers (Cin, Cout <= 64 at >= 128^2; bf16), gfx950.
//
// The 512^2 / 1024^2 layers of the generator (64 / 32 channels) and the first encoder blocks (16 / 32 channels) sit far
// under the MFMA ridge (107-288 flop/B): their floor is the byte stream.  conv_igemm_kernel ran them at 2.5-4x that floor
// because every 16x16 tile is one latency chain, and the first streaming version (a 4-wave workgroup sharing a strip, one
// barrier per step) was bound by instruction issue: ~1400 instructions per step against 36 MFMAs.  This version is built
// around the instruction count:
//
//   * ONE WAVE owns a strip 32 pixels wide (the MFMA N tile) and marches down it one output row per step; its halo rows
//     (34 pixels) live in a wave-private LDS ring of NR rows.  No workgroup barrier anywhere (Cout = 64 with Cin = 64 is
//     the exception: the 288 weight registers are split over a 2-wave team that shares the ring, one 2-wave barrier per step);
//   * rows arrive by LDS-DMA through a BUFFER descriptor of exactly one image row (buffer_load_dwordx4 ... offen lds):
//     lanes left / right of the image are out of range and the hardware writes zeros for them, rows above / below the
//     image use a zero-length descriptor - zero padding costs no instruction.  The immediate offset of the instruction
//     advances the global AND the LDS address, so a row is 2-5 instructions behind one M0 write; the partial last piece
//     runs under an EXEC mask (tools/probes/probe_ldsdma.hip pins all three properties on the hardware);
//   * the ring period is unrolled: every ds_read_b128 of the K loop is `lane offset register + immediate`, no address
//     arithmetic; fragments are prefetched PF reads ahead of their MFMA;
//   * every per-channel scale is folded into the WEIGHTS, held in registers as MFMA A operands:
//     W'[b][o][i] = bf16(W[o][i] * in_scale[b][i] * out_scale[b][o] * gain) - the reference's own fused-modulation form
//     (stylegan2_generator.py:858-864: the style multiplies the weight, the demodulation divides it).  The bias (times gain,
//     plus the interior value of a folded instance-norm shift, model/E/E.py:57,68) is the C operand of the first MFMA;
//   * the rows of the M tile are PERMUTED so that a lane's 16 accumulators are two runs of 8 consecutive channels:
//     bf16 pack -> two global_store_dwordx4 straight from registers (no LDS transpose, no cross-lane moves);
//   * noise rows and (data-gradient mode) `dot_src` rows ride the same DMA stream through their own small rings.
//
// Reference math: model/stylegan2_generator.py:855-922 (stride-1 branch), model/E/E.py:50-85.
#include <type_traits>
#include "common.h"
#include <stdlib.h>
#include "conv_params.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned rsrc_t;
#define DGE_LDS " lds\n\t"

__device__ __forceinline__ unsigned rfl(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

// raw buffer descriptor (gfx9 layout): base, stride 0, num_records in bytes, DATA_FORMAT = 32
__device__ __forceinline__ rsrc_t make_rsrc(unsigned long long base, unsigned bytes) {
    rsrc_t r;
    r[0] = rfl((unsigned)base); r[1] = rfl((unsigned)(base >> 32) & 0xffffu); r[2] = rfl(bytes); r[3] = 0x00020000u;
    return r;
}

// ---- LDS-DMA statements.  M0 (LDS base of the row) is written in the statement that uses it; the immediate offset moves the
//      global and the LDS address together.  The partial last piece of a row runs under an EXEC mask (saved / restored in
//      the statement; the code around it is wave-uniform).
// One halo row of 34 pixels: 1088 B (Cin 16), 2176 B (Cin 32), 4352 B (Cin 64) = 2 / 3 / 5 pieces of 1 KB, the last one
// partial.  TEAM = 2: the pieces alternate between the two waves (wave 1 pads with a zero-length piece where the counts
// differ, so that both waves count the same number of loads per row).
// Cin = 64: 8 pixels per piece while the chunk swizzle has a period of 16 pixels -> odd pieces use the second offset
// register; piece 4 lies beyond the 12-bit offset field -> second M0 value, third offset register.
#define DGE_P(off) "buffer_load_dwordx4 %1, %3, 0 offen offset:" #off " lds\n\t"
#define DGE_PB(off) "buffer_load_dwordx4 %4, %3, 0 offen offset:" #off " lds\n\t"
#define DGE_M0 "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
#define DGE_MASK(m) "s_mov_b64 %0, exec\n\ts_mov_b64 exec, " #m "\n\t"
#define DGE_UNMASK "s_mov_b64 exec, %0"
template <int CIN, int TEAM>
__device__ __forceinline__ void dma_row(int wave, unsigned voff, unsigned voff_b, unsigned voff_c, rsrc_t rs, rsrc_t rs_null,
                                        unsigned m0v, unsigned m0_dummy) {
    unsigned long long keep;
    if constexpr (CIN == 16 && TEAM == 1) {
        asm volatile(DGE_M0 DGE_P(0) DGE_MASK(0xf) DGE_P(1024) DGE_UNMASK : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(rs) : "memory");
    } else if constexpr (CIN == 16 && TEAM == 2) {
        if (wave == 0) asm volatile(DGE_M0 DGE_P(0) : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(rs) : "memory");
        else asm volatile(DGE_M0 DGE_MASK(0xf) DGE_P(1024) DGE_UNMASK : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(rs) : "memory");
    } else if constexpr (CIN == 32 && TEAM == 1) {
        asm volatile(DGE_M0 DGE_P(0) DGE_P(1024) DGE_MASK(0xff) DGE_P(2048) DGE_UNMASK : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(rs) : "memory");
    } else if constexpr (CIN == 32 && TEAM == 2) {
        if (wave == 0) asm volatile(DGE_M0 DGE_P(0) DGE_MASK(0xff) DGE_P(2048) DGE_UNMASK : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(rs) : "memory");
        else asm volatile(DGE_M0 DGE_P(1024) "s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %4, 0 offen lds"
                          : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(rs), "s"(rs_null), "s"(m0_dummy) : "memory");
    } else if constexpr (CIN == 64 && TEAM == 1) {
        asm volatile(DGE_M0 DGE_P(0) DGE_PB(1024) DGE_P(2048) DGE_PB(3072)
                     "s_mov_b32 m0, %6\n\t" DGE_MASK(0xffff) "buffer_load_dwordx4 %5, %3, 0 offen lds\n\t" DGE_UNMASK
                     : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(rs), "v"(voff_b), "v"(voff_c), "s"(m0v + 4096u) : "memory");
    } else {
        if (wave == 0)
            asm volatile(DGE_M0 DGE_P(0) DGE_P(2048)
                         "s_mov_b32 m0, %5\n\t" DGE_MASK(0xffff) "buffer_load_dwordx4 %4, %3, 0 offen lds\n\t" DGE_UNMASK
                         : "=&s"(keep) : "v"(voff), "s"(m0v), "s"(rs), "v"(voff_c), "s"(m0v + 4096u) : "memory");
        else
            asm volatile(DGE_M0 DGE_P(1024) DGE_P(3072) "s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %4, 0 offen lds"
                         : "=&s"(keep) : "v"(voff_b), "s"(m0v), "s"(rs), "s"(rs_null), "s"(m0_dummy) : "memory");
    }
}
// one full piece (8 noise rows of 32 pixels: lane = (row, 4 pixels))
__device__ __forceinline__ void dma_piece(unsigned voff, rsrc_t rs, unsigned m0v) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(m0v), "s"(rs) : "memory");
}
// dot_src row of this wave: NP pieces of 1 KB (32 pixels x 16 / 32 / 32 channels)
template <int NP> __device__ __forceinline__ void dma_dot(unsigned voff, unsigned voff1, rsrc_t rs, unsigned m0v) {
    if constexpr (NP == 1)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(m0v), "s"(rs) : "memory");
    else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds\n\t"
                     "buffer_load_dwordx4 %3, %2, 0 offen offset:1024 lds" : : "v"(voff), "s"(m0v), "s"(rs), "v"(voff1) : "memory");
}

enum { FL_GEN = 0, FL_ENC = 1, FL_ENC_STATS = 2, FL_DOT = 3, FL_DOT_PREP = 4, FL_GEN_RGB = 5, FL_ENC_POOL = 6, FL_DOT_IN = 7, FL_DOT_FR = 8, FL_DOT_INX = 9 };
// FL_DOT_INX:   the block-input form of FL_DOT_IN (conv_1 of an encoder block, Cin = Cout): y = A*acc + Bc*x + Cc + extra_scale *
//               extra[parent pixel] - instance-norm backward of the block input plus the pooled skip gradient (model/E/E.py:77-84
//               differentiated), no activation, no reductions; the pooled row rides the DMA stream (1 KB per wave and step).
// FL_DOT_FR:    the LAST data gradient of the encoder backward (conv_1 of block 0, 16 -> 16): x = dot_src is the FromRGB output x0 =
//               lrelu(W img + b) (model/utils/net.py:231-240).  g_x0 = A*acc + Bc*x0 + Cc + extra_scale*extra[parent pixel] (instance
//               norm backward + the pooled skip gradient, model/E/E.py:77-84 differentiated) has ONE reader, the FromRGB parameter
//               gradients: it is not stored but reduced in place, fr_out[b][c][0..3] += sum g_pre * (img_r, img_g, img_b, 1), g_pre =
//               g_x0 * lrelu'(x0).  Two more 512-byte rows ride the DMA stream per step: the pooled extra row and the image row in
//               pixel-major form [H][W][4] f32 (fr_img4, fourth component 1).
// FL_DOT_IN:    data-gradient mode with the instance-norm backward of the layer's input x = dot_src applied in the epilogue
//               (ConvParams::in_coef, model/E/E.py:51-62 differentiated): y = g_pre = (A*acc + Bc*x + Cc) * lrelu'(x); A is folded into
//               the weights, Cc is the C operand of the first MFMA; prep_stats (sum g_pre, sum g_pre*noise) = the bias / noise-weight
//               gradients of the layer that produced x; the noise ring carries that layer's plane (prep_noise).  No dot statistics:
//               the caller had them from the weight gradient (dge_conv_wgrad_dots) to build the coefficients.
// FL_ENC_POOL:  FL_ENC with the 2x2 average pool of the result (BEBlock: downscale2d after conv_2, model/E/E.py:75-76) taken in the
//               epilogue: the full-resolution activation is never stored - only its pooled value and, for the backward, the signs
// FL_GEN_RGB:   FL_GEN + the toRGB of the result (ConvParams::rgb_*): two more MFMAs per row on the packed output registers
// FL_GEN:       uniform noise weight (or none), no input shift, no statistics            (generator forward, LPIPS convs)
// FL_ENC:       per-channel noise weight, folded instance-norm shift with border terms    (encoder forward) - superset of GEN
// FL_ENC_STATS: + (sum, sum of squares) of the output per (sample, channel)
// FL_DOT:       data-gradient mode: y = acc * out_scale, statistics (sum acc * dot_src, sum acc); no bias / noise / activation
// FL_DOT_PREP:  FL_DOT + the tail backward of the layer that produced dot_src (ConvParams::prep): y = g_z, statistics
//               (sum acc * dot_src, -) and prep_stats (sum g_z * (z - ns*noise), sum g_z); the noise ring carries that layer's plane

template <int CIN, int COUT, int FL>
struct SC {
    static constexpr int PXB = CIN * 2, CH = PXB / 16, LOGCH = ilog2(CH), KS = CIN / 16;
    static constexpr int TEAM = COUT == 64 ? 2 : 1;                            // waves sharing one strip: 32 output channels each
    static constexpr int TPW = 4 / TEAM;                            // independent teams per workgroup (one wave per SIMD)
    static constexpr int MT = COUT / 32 > 0 ? COUT / 32 : 1;                   // M tiles of the strip (Cout 16: one half-empty tile)
    static constexpr int MTW = MT / TEAM;                                      // M tiles per wave
    static constexpr int HW = 34, RB = HW * PXB;
    static constexpr int PIECES = (RB + 1023) / 1024;
    static constexpr bool PREP = FL == FL_DOT_PREP, RGB = FL == FL_GEN_RGB, POOL = FL == FL_ENC_POOL, FRB = FL == FL_DOT_FR, INX = FL == FL_DOT_INX, INB = FL == FL_DOT_IN || FRB || INX;
    static_assert(!RGB || COUT == 32 || (COUT == 64 && TEAM == 2), "fused toRGB: 32 output channels per wave, one wave or a 2-wave team per pixel");
    static constexpr bool DOT = FL == FL_DOT || PREP || INB, STATS = FL == FL_ENC_STATS, ENC = FL == FL_ENC || FL == FL_ENC_STATS || POOL;
    static constexpr bool NOISE = !DOT || PREP || (INB && !FRB && !INX);
    static constexpr int NR = 6;                 // ring rows = unroll period
    static constexpr int dot_depth(int nr) { for (int d = 4; d >= 1; d--) if (d <= nr - 3 && nr % (d + 1) == 0) return d; return 1; }
    static constexpr int D = DOT ? dot_depth(6) : 2;            // rows in flight ahead of the newest live row (<= NR - 3)
    static constexpr int DR = DOT ? D + 1 : NR;                                // dot ring rows (must divide NR)
    static_assert(NR % DR == 0, "dot ring period");
    static constexpr int CPB = COUT * 2;
    static constexpr int CW = COUT / TEAM >= 32 ? 32 : 16;                      // channels of one wave's M tile set that are real (per tile)
    static constexpr int DCH = CW * MTW * 2 / 16;                                // 16-byte chunks of this wave's channels per pixel
    static constexpr int LOGDCH = ilog2(DCH);
    static constexpr int DROWB = 32 * DCH * 16, DPIECES = DROWB / 1024;
    static_assert(!DOT || MTW == 1, "data-gradient mode: one M tile per wave");
    static constexpr int XPW = TEAM == 2 ? (PIECES + 1) / 2 : PIECES;           // x loads per wave per row
    static constexpr int LPR = XPW + (DOT ? DPIECES : 0) + (FRB ? 2 : 0) + (INX ? 1 : 0);       // loads per wave per step (+ 1 noise piece when I == 0)
    static constexpr int X_OFF = 0;
    static constexpr int XRING = NR * RB;
    static constexpr int N_OFF = (XRING + 1023) / 1024 * 1024;                  // per wave: two noise buffers of 8 rows x 32 pixels f32
    static constexpr int NRING = NOISE ? 2048 : 0;
    static_assert(NR <= 8, "a noise piece holds 8 rows");
    static constexpr int D_OFF = N_OFF + TEAM * NRING;                           // per wave: dot ring DR x DROWB
    static constexpr int DRING = DOT ? DR * DROWB : 0;
    static constexpr int E_OFF = D_OFF + TEAM * DRING;                           // FRB: extra ring and image ring, DR rows of 512 B each
    static constexpr int ERING = FRB ? 2 * DR * 512 : (INX ? DR * 1024 : 0);      // (INX: per wave, 16 pooled pixels x 32 channels per row)
    static_assert(!INX || (CIN == COUT && CIN >= 32), "block-input flavour: 32 -> 32, 64 -> 64");
    static_assert(!FRB || (CIN == 16 && COUT == 16 && TEAM == 1), "FromRGB reduction flavour: 16 -> 16");
    static constexpr int T_OFF = E_OFF + TEAM * ERING;                           // per wave: T table [32*MTW][12] f32 + epilogue constants [3][32*MTW]
    static constexpr int TBYTES = 32 * MTW * 16 * 4;
    static constexpr int DUMMY_OFF = T_OFF + TEAM * TBYTES;
    static constexpr int RGBX_OFF = DUMMY_OFF + (TEAM == 2 ? 1024 : 0);             // fused toRGB of a team: wave 1's partial sums, [2 row parities][3][32] f32
    static constexpr int LDS_BYTES = RGBX_OFF + (RGB && TEAM == 2 ? 1024 : 0);
    static constexpr int NEED = 4 * 9 * KS * MTW + 32 * MTW + 16 + (ENC ? 16 * MTW : 0) + (STATS || DOT ? 32 * MTW : 0) + (DOT ? 16 : 0) + (PREP ? 16 : 0) + (RGB ? 11 : 0) + (POOL ? 20 * MTW : 0) + 36;
    // (the 64-channel prep flavour holds 144 weight + 48 sum registers: at two waves per SIMD it spilled 360 B per lane)
    static constexpr int WPE = ((PREP || INB) && CIN == 64) ? 1 : (NEED <= 128 ? 4 : (NEED <= 168 ? 3 : 2));
    static_assert(D <= NR - 3, "the slot of the row being fetched must be dead");
    static_assert((D - 1) * LPR + 1 < 64, "vmcnt range");
};

// swizzle of the 16-byte chunks of a pixel in an LDS image with 2^LOGC chunks per pixel (conflict-free ds_read_b128 of 16
// consecutive pixels): applied on the SOURCE address of the DMA and on the read address
template <int LOGC> __device__ __forceinline__ int chunk_swz(int px) { return (px >> (4 - LOGC)) & ((1 << LOGC) - 1); }

// channel of row m of the (permuted) M tile.  The accumulator of lane (pixel, kh) holds rows 8j + 4kh + i (j, i < 4) in
// register 4j + i; with this map registers 0-7 are channels 8kh .. 8kh+7 and registers 8-15 channels 16+8kh .. 16+8kh+7
// (16-channel tile: registers 0-7 = channels 8kh .. 8kh+7, rows >= 16 empty).
template <int CW> __device__ __forceinline__ int chan_of_row(int m) {
    const int j = m >> 3, k = (m >> 2) & 1, i = m & 3;
    if (CW == 16) return j < 2 ? 8 * k + 4 * j + i : -1;
    return 16 * (j >> 1) + 8 * k + 4 * (j & 1) + i;
}

template <int CIN, int COUT, int FL>
__global__ __launch_bounds__((64 * SC<CIN, COUT, FL>::TEAM * SC<CIN, COUT, FL>::TPW), (SC<CIN, COUT, FL>::WPE))
void conv_stream_kernel(ConvParams p, int nstrips, int nseg, int seg_rows, int njobs, int jobs_per_xcd) {
    using C = SC<CIN, COUT, FL>;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_all[];
#ifdef DGE_SC_TIMING          // tuning builds (tools/build_variant.sh ... main): per-job clock stamps, see the end of the kernel
    const long long tj0 = __builtin_readcyclecounter();
#endif
    const int lane = threadIdx.x & 63;
    const int wid = (int)rfl(threadIdx.x >> 6);
    const int team = wid / C::TEAM;                                              // teams of a workgroup never synchronise with each other
    const int wave = wid % C::TEAM;
    unsigned char* __restrict__ lds = lds_all + team * C::LDS_BYTES;
    const unsigned lds0 = lds_off(lds_all) + team * C::LDS_BYTES;
    const int n31 = lane & 31, kh = lane >> 5;

    // ---- job: blocks of one XCD (blockIdx % 8) take a contiguous range of (sample, segment, strip): neighbouring strips
    //      share their halo columns through that XCD's L2
    int job = ((blockIdx.x & 7) * jobs_per_xcd + (blockIdx.x >> 3)) * C::TPW + team;
    if ((blockIdx.x >> 3) >= jobs_per_xcd || job >= njobs) return;
    const int strip = job % nstrips; job /= nstrips;
    const int seg = job % nseg;
    const int b = job / nseg;
    const int x0 = strip * 32;
    const int r0 = seg * seg_rows;
    const int rows = min(seg_rows, p.H - r0);
    const bool full_strip = x0 + 32 <= p.W;
    const bool edge_strip = x0 == 0 || x0 + 32 >= p.W;

    const unsigned xrow_bytes = (unsigned)p.W * C::PXB, yrow_bytes = (unsigned)p.W * C::CPB;
    const unsigned long long Xb = (unsigned long long)p.x + (unsigned long long)b * p.H * xrow_bytes;
    unsigned char* __restrict__ Yb = (unsigned char*)p.y + (size_t)b * p.H * yrow_bytes;
    const float* __restrict__ nzsrc = (C::PREP || C::INB) ? p.prep_noise : p.noise;      // (prep: the plane of the layer whose tail is differentiated)
    const unsigned long long NZb = nzsrc ? (unsigned long long)(nzsrc + (size_t)b * ((C::PREP || C::INB) ? p.prep_noise_bstride : p.noise_bstride)) : Xb;
    const unsigned nzrow_bytes = nzsrc ? (unsigned)p.W * 4 : 0;
    const unsigned long long DOTb = C::DOT ? (unsigned long long)p.dot_src + (unsigned long long)b * p.H * yrow_bytes : Xb;

    // ---- DMA lane offsets.  x row: 16-byte unit u = lane (+ 64 per piece) = (halo pixel, chunk); source chunk swizzled
    unsigned voff, voff_b = 0, voff_c = 0;
    {
        const int px = lane >> C::LOGCH, cs = lane & (C::CH - 1);
        const int gx = x0 - 1 + px;
        voff = (unsigned)(gx * C::PXB + ((cs ^ chunk_swz<C::LOGCH>(px)) << 4));      // negative (left of the image) = out of range
        if constexpr (CIN == 64) {
            voff_b = (unsigned)(gx * C::PXB + ((cs ^ chunk_swz<C::LOGCH>(px + 8)) << 4));   // odd pieces: pixel + 8 (the offset adds the rest)
            voff_c = voff + 4096;                                                     // piece 4 (even), beyond the 12-bit offset field
        }
    }
    const unsigned nvoff = (unsigned)(((lane >> 3) * p.W + x0 + (lane & 7) * 4) * 4);   // noise piece: 8 rows x 32 pixels (pixels right of the image read the next row: never stored)
    unsigned dvoff = 0, dvoff1 = 0;
    if constexpr (C::DOT) {
        const int px = lane >> C::LOGDCH, cs = lane & (C::DCH - 1);
        const int chunk0 = wave * 4;                                                   // first chunk of this wave's channels
        dvoff = (unsigned)((x0 + px) * C::CPB + ((chunk0 + (cs ^ chunk_swz<C::LOGDCH>(px))) << 4));
        constexpr int PPP = 64 / C::DCH;                                                // pixels per piece
        dvoff1 = (unsigned)((x0 + px + PPP) * C::CPB + ((chunk0 + (cs ^ chunk_swz<C::LOGDCH>(px + PPP))) << 4)) - 1024u;
    }
    const rsrc_t rs_null = make_rsrc(Xb, 0);
    // FRB: rows of the pooled skip gradient [B][H/2][W/2][16] bf16 and of the pixel-major image [B][H][W][4] f32
    const unsigned exrow_bytes = (C::FRB || C::INX) ? (unsigned)(p.W >> 1) * C::CPB : 0u, imrow_bytes = C::FRB ? (unsigned)p.W * 16u : 0u;
    const unsigned long long EXb = ((C::FRB || C::INX) && p.in_extra) ? (unsigned long long)p.in_extra + (unsigned long long)b * (p.H >> 1) * exrow_bytes : Xb;
    const unsigned long long IMb = C::FRB ? (unsigned long long)p.fr_img4 + (unsigned long long)b * p.H * imrow_bytes : Xb;
    // (INX: lane = (pooled pixel, 16-byte chunk of this wave's 32 channels), chunk swizzled like the other rings)
    const unsigned evoff = C::INX ? (unsigned)(((x0 >> 1) + (lane >> 2)) * C::CPB + wave * 64 + (((lane & 3) ^ chunk_swz<2>(lane >> 2)) << 4))
                                  : (unsigned)((x0 >> 1) * 32 + (lane & 31) * 16);
    const unsigned ivoff = (unsigned)((x0 + (lane & 31)) * 16);

    // ---- issue of (x halo row h, noise row h-2, dot row h-2) into ring slot `slot` / `slot2`
    // row pointers of the NEXT row to fetch (advance one image row per issue; rows outside the image are never dereferenced:
    // their descriptor has length zero)
    unsigned long long xptr = Xb + (unsigned long long)(long long)(r0 - 1) * xrow_bytes;
    unsigned long long dptr = DOTb + (unsigned long long)(long long)(r0 - 2) * yrow_bytes;
    auto issue = [&](int h, int slot, int slot_d) {
        const int gy = r0 - 1 + h;
        const bool xv = (unsigned)gy < (unsigned)p.H && h <= rows + 1;
        const rsrc_t rx = make_rsrc(xptr, xv ? xrow_bytes : 0u);
        xptr += xrow_bytes;
        const unsigned m0x = lds0 + C::X_OFF + slot * C::RB;
        dma_row<CIN, C::TEAM>(wave, voff, voff_b, voff_c, rx, rs_null, m0x, lds0 + C::DUMMY_OFF);
        const bool ov = h >= 2 && h <= rows + 1;                                        // output row gy - 1 lies inside the segment
        if constexpr (C::DOT) {
            const rsrc_t rd = make_rsrc(dptr, ov ? yrow_bytes : 0u);
            dptr += yrow_bytes;
            dma_dot<C::DPIECES>(dvoff, dvoff1, rd, lds0 + C::D_OFF + wave * C::DRING + slot_d * C::DROWB);
        }
        if constexpr (C::INX) {
            const int oy = gy - 1;
            const rsrc_t re = make_rsrc(EXb + (unsigned long long)(ov ? (oy >> 1) : 0) * exrow_bytes, (ov && p.in_extra) ? exrow_bytes : 0u);
            dma_piece(evoff, re, lds0 + C::E_OFF + wave * C::ERING + slot_d * 1024);
        }
        if constexpr (C::FRB) {
            // output row gy - 1: its pooled parent row of `extra` (16 pixels x 32 B of this strip) and its image row (32 pixels x 16 B)
            const int oy = gy - 1;
            const rsrc_t re = make_rsrc(EXb + (unsigned long long)(ov ? (oy >> 1) : 0) * exrow_bytes, (ov && p.in_extra) ? exrow_bytes : 0u);
            const rsrc_t ri = make_rsrc(IMb + (unsigned long long)(ov ? oy : 0) * imrow_bytes, ov ? imrow_bytes : 0u);
            unsigned long long keep;
            asm volatile(DGE_MASK(0xffffffff) "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
                         "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %6, 0 offen lds\n\t" DGE_UNMASK
                         : "=&s"(keep) : "v"(evoff), "s"(lds0 + C::E_OFF + slot_d * 512), "s"(re), "s"(lds0 + C::E_OFF + C::DR * 512 + slot_d * 512), "v"(ivoff), "s"(ri) : "memory");
        }
    };
    // noise rows q .. q+7 of the segment (one piece; NR of them are used) into buffer `buf_off` (0 / 1024)
    auto issue_noise = [&](int q, unsigned buf_off) {
        const int gy = r0 + q;
        const bool v = nzsrc != nullptr && gy < p.H && q < rows;
        const rsrc_t rn = make_rsrc(NZb + (unsigned long long)(v ? gy : 0) * nzrow_bytes, v ? (unsigned)(p.H - gy) * nzrow_bytes : 0u);
        dma_piece(nvoff, rn, lds0 + C::N_OFF + wave * C::NRING + buf_off);
    };
    // rows 0 .. D+1 start before the weights are touched
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (C::NOISE) issue_noise(0, 0u);
    StaticFor<C::D + 2>::run([&](auto hc) { constexpr int h = decltype(hc)::value; issue(h, h % C::NR, (h + C::NR - 2) % C::NR % C::DR); });
    // (noise / dot rows -2 and -1 are zero-length; their slot index only has to be in range)

    // ---- weights -> registers, every scale folded in:  W'[o][k] = bf16(W[o][k] * in_scale[b][k] * out_scale[b][o] * gain)
    //      (data-gradient mode keeps out_scale for the epilogue: its statistics are those of the unscaled accumulator)
    uint4 wf[C::MTW][9][C::KS];
    float* __restrict__ TT = (float*)(lds + C::T_OFF + wave * C::TBYTES);              // [32*MTW][16]: taps 0..8 | 9: all taps | 10 / 11: left / right tap column | 12: bias' | 13: noise weight | 14: out scale
    {
        const bf16_t* __restrict__ Wp = (const bf16_t*)p.w;
        float sc[C::KS][8], rt[C::KS][8];
#pragma unroll
        for (int ks = 0; ks < C::KS; ks++) {
            const int k0 = ks * 16 + kh * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                sc[ks][e] = p.in_scale ? p.in_scale[b * CIN + k0 + e] : 1.f;
                const float sh = (C::ENC && p.in_shift) ? p.in_shift[b * CIN + k0 + e] : 0.f;
                rt[ks][e] = sc[ks][e] != 0.f ? sh / sc[ks][e] : 0.f;
            }
        }
#pragma unroll
        for (int mt = 0; mt < C::MTW; mt++) {
            const int cl = chan_of_row<C::CW>(n31);                                       // channel within the tile, -1: empty row
            const int o = (wave * C::MTW + mt) * 32 + cl;
            const bool ov = cl >= 0 && o < COUT;
            float osc = p.gain;
            if (!C::DOT && p.out_scale && ov) osc *= p.out_scale[b * COUT + o];
            if (C::INB && ov) osc *= p.in_coef[((size_t)b * COUT + o) * 3];
            // the row's epilogue constants: requested here, used behind the tap loop (they were three more dependent round trips there)
            const int oc = ov ? o : 0;
            float c_bias = (!C::DOT && p.bias) ? p.bias[oc] : 0.f;
            float c_nw = (C::NOISE && p.noise) ? p.noise_w[oc * p.noise_w_stride] : 0.f;
            float c_osc = (C::DOT && !C::INB && p.out_scale) ? p.out_scale[b * COUT + oc] : 1.f;
            float c_in1 = 0.f, c_in2 = 0.f;
            if constexpr (C::INB) { c_in1 = p.in_coef[((size_t)b * COUT + oc) * 3 + 1]; c_in2 = p.in_coef[((size_t)b * COUT + oc) * 3 + 2]; }
            float tall = 0.f, tleft = 0.f, tright = 0.f;
            // all 9 * KS loads of the row go out before the first one is used (rows that do not exist read row 0 and are zeroed below):
            // with the load inside the tap loop every tap waited for its own round trip - 9 dependent misses, ~19 k cycles of a job's
            // ~200 k (measured per job, tools/perf_stream_jobs.py)
            {
                const bf16_t* __restrict__ wrow = Wp + (size_t)(ov ? o : 0) * CIN + kh * 8;
#pragma unroll
                for (int tap = 0; tap < 9; tap++)
#pragma unroll
                    for (int ks = 0; ks < C::KS; ks++) wf[mt][tap][ks] = *(const uint4*)(wrow + (size_t)tap * p.Ntot * CIN + ks * 16);
            }
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                float tt = 0.f;
#pragma unroll
                for (int ks = 0; ks < C::KS; ks++) {
                    uint4 v = ov ? wf[mt][tap][ks] : make_uint4(0, 0, 0, 0);
                    float f[8];
                    unpack16(v, f, (bf16_t*)nullptr);
#pragma unroll
                    for (int e = 0; e < 8; e++) f[e] *= sc[ks][e] * osc;
                    v = pack16(f, (bf16_t*)nullptr);
                    if constexpr (C::ENC) {                 // T uses the ROUNDED weights: W'.(x + sh/sc) is then exact in W'
                        unpack16(v, f, (bf16_t*)nullptr);
#pragma unroll
                        for (int e = 0; e < 8; e++) tt += f[e] * rt[ks][e];
                    }
                    wf[mt][tap][ks] = v;
                }
                if constexpr (C::ENC) {
                    tt += __shfl_xor(tt, 32, 64);                                           // the two K halves of the row
                    if (kh == 0 && ov) TT[(mt * 32 + cl) * 16 + tap] = tt;
                    tall += tt;
                    if (tap % 3 == 0) tleft += tt;
                    if (tap % 3 == 2) tright += tt;
                }
            }
            if (kh == 0 && ov) {
                const int e = (mt * 32 + cl) * 16;
                float bb = (!C::DOT && p.bias) ? c_bias * p.bias_scale * p.gain : 0.f;
                if constexpr (C::ENC) { TT[e + 9] = tall; TT[e + 10] = tleft; TT[e + 11] = tright; bb += tall; }
                if constexpr (C::INB) bb = c_in2;
                TT[e + 12] = bb;
                TT[e + 13] = (C::NOISE && p.noise) ? c_nw * p.gain : 0.f;
                TT[e + 14] = C::INB ? c_in1 : c_osc;
            }
        }
    }
    // channel (within the wave's tile set) of accumulator register r of this lane
    auto chan_of_reg = [&](int mt, int r) { return mt * 32 + (C::CW == 16 ? 8 * kh + r : 16 * (r >> 3) + 8 * kh + (r & 7)); };
    constexpr int NREG = C::CW == 16 ? 8 : 16;                                               // live accumulator registers per tile
    f32x16_t biasv[C::MTW], zero16;
#pragma unroll
    for (int r = 0; r < 16; r++) zero16[r] = 0.f;
    float nwv[C::MTW][C::ENC ? 16 : 1], oscv[C::DOT ? 16 : 1];
#pragma unroll
    for (int mt = 0; mt < C::MTW; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const bool live = r < NREG;
            biasv[mt][r] = live ? TT[chan_of_reg(mt, r) * 16 + 12] : 0.f;
            if constexpr (C::ENC) nwv[mt][r] = live ? TT[chan_of_reg(mt, r) * 16 + 13] : 0.f;
            if constexpr (C::DOT) oscv[r] = live ? TT[chan_of_reg(mt, r) * 16 + 14] : 0.f;
            if constexpr (C::ENC) {
                // zero padding follows the norm: the tap COLUMN that falls outside the image does not carry the folded shift.  That term
                // depends on the lane (= pixel column) and the channel only: it goes into the lane's C operand here, once.  (It used to be
                // subtracted per row from a table in LDS - 16 dependent LDS round trips per row on the two edge strips of an image, 3 x
                // the step time of an interior strip; all strips of a launch are resident at once, so the edge strips set its duration.)
                const int gxl = x0 + n31;
                if (live && edge_strip && (gxl == 0 || gxl == p.W - 1)) {
                    const float* __restrict__ T = TT + chan_of_reg(mt, r) * 16;
                    biasv[mt][r] -= (gxl == 0 ? T[10] : 0.f) + (gxl == p.W - 1 ? T[11] : 0.f);
                }
            }
        }
    const float nw_uniform = (!C::ENC && C::NOISE && p.noise) ? p.noise_w[0] * p.gain : 0.f;
    const float slope = p.act == DGE_ACT_LRELU ? 0.2f : (p.act == DGE_ACT_RELU ? 0.f : 1.f);

    // ---- fused toRGB: the packed bf16 output registers of a lane ARE the B operand of a K = 32 product over the output channels
    //      (same permutation as the input fragments).  A rows 0..2 = bf16 hi part of the modulated f32 toRGB weights, rows 8..10 =
    //      their lo part (hi + lo carries 16 mantissa bits): lane (pixel, kh = 0) finds row k in register k, row 8 + k in 4 + k.
    uint4 rgbA[C::RGB ? 2 : 1];
    float rgbb[3] = {0.f, 0.f, 0.f};
    if constexpr (C::RGB) {
        const int row = n31 & 7, part = n31 >> 3;                       // part 0: hi, 1: lo
        const bool live = row < 3 && part < 2;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int c = wave * 32 + 16 * ks + 8 * kh + e;
                const float wv = live ? p.rgb_w[row * COUT + c] * p.rgb_wscale * p.rgb_style[b * COUT + c] : 0.f;
                const float hi = __uint_as_float(__float_as_uint(wv) & 0xffff0000u);      // (truncated hi: the lo part takes the rest)
                f[e] = part == 0 ? hi : wv - hi;
            }
            rgbA[ks] = pack16(f, (bf16_t*)nullptr);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) rgbb[k] = wave == 0 ? p.rgb_bias[k] : 0.f;
    }
    // 2-wave team: wave 1 leaves its partial sums of row s in LDS, wave 0 keeps its own and stores the sum one step later (after
    // the team barrier of step s + 1)
    float rgbh[3] = {0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) float* lds_wfp;
    auto rgb_flush = [&](int srow) {
        if (kh == 0 && x0 + n31 < p.W) {
            const unsigned a = lds0 + C::RGBX_OFF + (unsigned)(srow & 1) * 384 + n31 * 4;
            float* __restrict__ ro = p.rgb_out + ((size_t)b * 3 * p.H + (r0 + srow)) * p.W + x0 + n31;
            const size_t plane = (size_t)p.H * p.W;
#pragma unroll
            for (int k = 0; k < 3; k++) ro[k * plane] = rgbh[k] + *(const __attribute__((address_space(3))) float*)(a + k * 128);
        }
    };

    // ---- B-fragment lane offsets: halo pixel n31 + dx, chunk (ks*2 + kh) ^ swizzle
    unsigned loff[3][C::KS];
#pragma unroll
    for (int dx = 0; dx < 3; dx++)
#pragma unroll
        for (int ks = 0; ks < C::KS; ks++) {
            const int px = n31 + dx;
            loff[dx][ks] = lds0 + C::X_OFF + px * C::PXB + (((ks * 2 + kh) ^ chunk_swz<C::LOGCH>(px)) << 4);
        }
    const unsigned nzoff = lds0 + C::N_OFF + wave * C::NRING + n31 * 4;
    unsigned doff[2] = {0, 0};
    if constexpr (C::DOT) {
#pragma unroll
        for (int q = 0; q < 2; q++)
            doff[q] = lds0 + C::D_OFF + wave * C::DRING + n31 * C::DCH * 16 + (((2 * q + kh) ^ chunk_swz<C::LOGDCH>(n31)) << 4);
    }
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) u32x4_t* lds_u4p;
    typedef const __attribute__((address_space(3))) float* lds_fp;
    auto lds_u4 = [&](unsigned a) { const u32x4_t t = *(lds_u4p)a; return make_uint4(t[0], t[1], t[2], t[3]); };
    auto lds_f = [&](unsigned a) { return *(lds_fp)a; };

    const int gx = x0 + n31;
    const bool pv = gx < p.W;
    // output byte offset of this lane inside an image row: pixel, this wave's channels, this lane's first run of 8
    const unsigned yoff = (unsigned)gx * C::CPB + (unsigned)(wave * C::MTW * 32 * 2) + kh * 16;
    // pooled flavour: buffer descriptors of this sample's pooled output [H/2][W/2][COUT] bf16 and sign mask [H/2][W/2][COUT/8] words;
    // lane offsets of the even pixels inside the image (every other lane: out of range = the store is dropped)
    const bool pool_lane = C::POOL && (n31 & 1) == 0 && pv;
    const unsigned py_voff = pool_lane ? (unsigned)(gx >> 1) * C::CPB + (unsigned)(wave * C::MTW * 64) + kh * 16 : 0x80000000u;
    const unsigned pm_voff = pool_lane ? ((unsigned)(gx >> 1) * (COUT / 8) + (unsigned)(wave * C::MTW * 4) + kh) * 4u : 0x80000000u;
    const unsigned pool_px = C::POOL ? (unsigned)(p.H >> 1) * (unsigned)(p.W >> 1) : 0u;
    const __amdgpu_buffer_rsrc_t rs_pool = __builtin_amdgcn_make_buffer_rsrc((unsigned char*)p.y + (size_t)b * pool_px * C::CPB, 0, (int)(pool_px * C::CPB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mask = __builtin_amdgcn_make_buffer_rsrc((C::POOL && p.pool_mask) ? (unsigned char*)p.pool_mask + (size_t)b * pool_px * (COUT / 8) * 4 : (unsigned char*)p.y, 0,
                                                                               (C::POOL && p.pool_mask) ? (int)(pool_px * (COUT / 8) * 4) : 0, 0x00020000);

    float s0[C::MTW][(C::STATS || C::DOT) ? 16 : 1], s1[C::MTW][(C::STATS || C::DOT) ? 16 : 1], s2[(C::PREP || C::INB) ? 16 : 1];
    if constexpr (C::STATS || C::DOT) {
#pragma unroll
        for (int mt = 0; mt < C::MTW; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) { s0[mt][r] = 0.f; s1[mt][r] = 0.f; }
    }
    if constexpr (C::PREP || C::INB) {
#pragma unroll
        for (int r = 0; r < 16; r++) s2[r] = 0.f;
    }
    // pooled flavour: the even row of the current row pair (activated values and their sign bytes), this lane's pixel
    float prow[C::POOL ? C::MTW : 1][C::POOL ? 16 : 1];
    unsigned psign[C::POOL ? C::MTW : 1];
    // prep: x = lrelu(z)*gain of the layer below -> g_z = g * gain * lrelu'(x), z = x / (gain * lrelu'(x))
    const float pg_pos = p.prep_gain, pg_neg = 0.2f * p.prep_gain, pz_pos = 1.f / p.prep_gain, pz_neg = 1.f / (0.2f * p.prep_gain);
    const float pns = (C::PREP && p.prep_noise && p.prep_ns) ? p.prep_ns[0] : 0.f;
    const float fr_es = (C::FRB || C::INX) ? p.in_extra_scale : 0.f;

    // ---- one output row.  I = position in the ring period (compile time): halo rows in slots I, I+1, I+2 (mod NR)
#define DGE_T(k)
    auto step = [&](auto ic, int s, unsigned npar) {
        constexpr int I = decltype(ic)::value;
        DGE_T(0);
        // row s+2 (and the noise / dot row s) has landed when at most (D-1) rows' loads are still in flight
        // loads issued after row s+2's: the rows of the D-1 steps before this one, plus the noise piece if one of them was
        // a period start (ring position 0)
        constexpr int NAFTER = (C::D - 1) * C::LPR + ((C::NOISE && I >= 1 && I <= C::D - 1) ? 1 : 0);
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(NAFTER) : "memory");
        if constexpr (C::RGB && C::TEAM == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (wave 1's partial sums of row s-1 are in LDS)
        if constexpr (C::TEAM == 2) __builtin_amdgcn_s_barrier();     // the partner's pieces landed; it finished step s-1
        if constexpr (C::RGB && C::TEAM == 2) { if (wave == 0 && s > 0) rgb_flush(s - 1); }
        DGE_T(1);
        if constexpr (C::NOISE && I == 0) issue_noise(s + C::NR, npar ^ 1024u);   // the next period's rows into the other buffer
        issue(s + 2 + C::D, (I + 2 + C::D) % C::NR, (I + C::D) % C::DR);
        DGE_T(2);

        const int gy = r0 + s;
        constexpr int NQ = 9 * C::KS;
        constexpr int PF = NQ < 6 ? NQ : 6;
        auto frag_addr = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int tap = q / C::KS, ks = q % C::KS, dy = tap / 3, dx = tap % 3;
            return loff[dx][ks] + (unsigned)(((I + dy) % C::NR) * C::RB);
        };
        float nz = 0.f;
        if constexpr (C::NOISE) nz = lds_f(nzoff + npar + I * 128);
        uint4 bq[PF];
        StaticFor<PF>::run([&](auto qc) { bq[decltype(qc)::value] = lds_u4(frag_addr(qc)); });
        f32x16_t acc[C::MTW];
        StaticFor<NQ>::run([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int tap = q / C::KS, ks = q % C::KS;
            const bf16x8_t bf = *(const bf16x8_t*)&bq[q % PF];
#pragma unroll
            for (int mt = 0; mt < C::MTW; mt++) {
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&wf[mt][tap][ks], bf, q == 0 ? ((C::DOT && !C::INB) ? zero16 : biasv[mt]) : acc[mt], 0, 0, 0);
            }
            if constexpr (q + PF < NQ) bq[q % PF] = lds_u4(frag_addr(std::integral_constant<int, q + PF>{}));
        });

        // pin the issue order: PF (+ noise) reads, then one read per MTW MFMAs, then the last PF fragments' MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, PF + (C::NOISE ? 1 : 0), 0);
        StaticFor<NQ - PF>::run([&](auto) {
            __builtin_amdgcn_sched_group_barrier(0x008, C::MTW, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x008, PF * C::MTW, 0);
        DGE_T(3);
        // ---------------- epilogue: this lane = pixel gx, NREG channels per tile
        unsigned char* __restrict__ yrow = Yb + (size_t)gy * yrow_bytes;
#pragma unroll
        for (int mt = 0; mt < C::MTW; mt++) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = acc[mt][r];
            if constexpr (C::ENC) {
                // zero padding follows the norm: taps that fall outside the image do not carry the folded shift
                // (the left / right tap columns: in the C operand, see biasv)  Top / bottom image row: two rows per image take this path
                const bool top = gy == 0, bot = gy == p.H - 1;
                if (top | bot) {
                    const float cl = gx == 0 ? 1.f : 0.f, cr = gx == p.W - 1 ? 1.f : 0.f;
#pragma unroll
                    for (int r = 0; r < NREG; r++) {
                        const float* __restrict__ T = TT + chan_of_reg(mt, r) * 16;
                        float corr = 0.f;
                        if (top) corr += T[0] + T[1] + T[2] - cl * T[0] - cr * T[2];
                        if (bot) corr += T[6] + T[7] + T[8] - cl * T[6] - cr * T[8];
                        v[r] -= corr;
                    }
                }
            }
            if constexpr (C::DOT) {
                uint4 dd[2];
                dd[0] = lds_u4(doff[0] + (I % C::DR) * C::DROWB);
                if constexpr (NREG == 16) dd[1] = lds_u4(doff[1] + (I % C::DR) * C::DROWB);
                const float nzs = C::PREP ? pns * nz : 0.f;
                float exf[8], exx[C::INX ? 2 : 1][8];
                if constexpr (C::INX) {
#pragma unroll
                    for (int q = 0; q < NREG / 8; q++)
                        unpack16(lds_u4(lds0 + C::E_OFF + wave * C::ERING + (I % C::DR) * 1024 + (n31 >> 1) * 64 + (((2 * q + kh) ^ chunk_swz<2>(n31 >> 1)) << 4)), exx[q], (bf16_t*)nullptr);
                }
                float4 fr_im = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (C::FRB) {
                    unpack16(lds_u4(lds0 + C::E_OFF + (I % C::DR) * 512 + (n31 >> 1) * 32 + kh * 16), exf, (bf16_t*)nullptr);
                    const uint4 t4 = lds_u4(lds0 + C::E_OFF + C::DR * 512 + (I % C::DR) * 512 + n31 * 16);
                    fr_im = make_float4(__uint_as_float(t4.x), __uint_as_float(t4.y), __uint_as_float(t4.z), __uint_as_float(t4.w));
                }
#pragma unroll
                for (int q = 0; q < NREG / 8; q++) {
                    float d[8];
                    unpack16(dd[q], d, (bf16_t*)nullptr);
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int r = 8 * q + e;
                        if constexpr (C::INX) {
                            v[r] = fmaf(fr_es, exx[C::INX ? q : 0][e], fmaf(oscv[r], d[e], v[r]));      // v = A*acc + Cc; oscv = Bc
                            continue;
                        }
                        if constexpr (C::FRB) {
                            // v = A*acc + Cc; oscv = Bc; the four sums of channel r live in s0[0][r], s0[0][8 + r], s1[0][r], s1[0][8 + r]
                            const float gp = (fmaf(oscv[r], d[e], v[r]) + fr_es * exf[e]) * (d[e] > 0.f ? 1.f : 0.2f);
                            s0[0][r] = fmaf(gp, fr_im.x, s0[0][r]); s0[0][8 + r] = fmaf(gp, fr_im.y, s0[0][8 + r]);
                            s1[0][r] = fmaf(gp, fr_im.z, s1[0][r]); s1[0][8 + r] += gp;
                            continue;
                        }
                        if constexpr (C::INB) {
                            // v = A*acc + Cc (weights, C operand); oscv = Bc
                            const float o = fmaf(oscv[r], d[e], v[r]) * (d[e] > 0.f ? 1.f : 0.2f);
                            s1[mt][r] += o; s2[r] = fmaf(o, nz, s2[r]);
                            v[r] = o;
                            continue;
                        }
                        s0[mt][r] = fmaf(v[r], d[e], s0[mt][r]);
                        if constexpr (C::PREP) {
                            const bool pos = d[e] > 0.f;
                            const float gz = v[r] * oscv[r] * (pos ? pg_pos : pg_neg);
                            const float zt = d[e] * (pos ? pz_pos : pz_neg) - nzs;
                            s2[r] = fmaf(gz, zt, s2[r]); s1[mt][r] += gz;
                            v[r] = gz;
                        } else {
                            s1[mt][r] += v[r];
                            v[r] *= oscv[r];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < NREG; r++) {
                    float t;
                    if constexpr (C::ENC) t = fmaf(nwv[mt][r], nz, v[r]);
                    else t = fmaf(nw_uniform, nz, v[r]);
                    v[r] = fmaxf(t, t * slope);
                }
                if constexpr (C::STATS) {
#pragma unroll
                    for (int r = 0; r < NREG; r++) { s0[mt][r] += v[r]; s1[mt][r] = fmaf(v[r], v[r], s1[mt][r]); }
                }
            }
            if constexpr (C::POOL) {
                // sign bytes of the two channel runs (bit e = channel e of the run is positive: all the activation backward needs)
                unsigned sb = 0;
#pragma unroll
                for (int r = 0; r < NREG; r++) sb |= (v[r] > 0.f ? 1u : 0u) << r;
                // (row parity = parity of the ring position: segments start on even rows - launch_stream - and the ring period is even;
                //  a run-time `gy & 1` branch cost 17 register copies per even row)
                static_assert(!C::POOL || C::NR % 2 == 0, "pooled flavour: even ring period");
                if constexpr ((I & 1) == 0) {
#pragma unroll
                    for (int r = 0; r < NREG; r++) prow[mt][r] = v[r];
                    psign[mt] = sb;
                    continue;
                }
                // odd row: this pixel's column sum, plus the neighbour column's (lanes 2j / 2j+1: quad_perm [1,0,3,2])
#pragma unroll
                for (int r = 0; r < NREG; r++) {
                    const float cs = prow[mt][r] + v[r];
                    v[r] = 0.25f * (cs + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(unsigned, cs), 0xB1, 0xf, 0xf, true)));
                }
                // stores through buffer descriptors of the sample's pooled plane / mask plane: lane part of the offset fixed for the strip
                // (odd pixels and pixels right of the image carry an out-of-range offset: dropped), row part scalar.  (64-bit per-lane
                // pointers here were spilled, and every reload waited for ALL outstanding loads - the row prefetches - with vmcnt(0).)
                const unsigned prow_i = (unsigned)(gy >> 1);
                if (p.pool_mask) {
                    // mask word of a pooled pixel and 8-channel chunk: byte q = position (2oy, 2ox), (2oy, 2ox+1), (2oy+1, 2ox), (2oy+1, 2ox+1)
                    // (the layout of dge_blend_pool_mask / dge_act_bwd_mask)
                    const unsigned own = (psign[mt] & 0xffffu) | (sb << 16);       // runs: [7:0], [15:8] even row | [23:16], [31:24] odd row
                    const unsigned nbr = __builtin_amdgcn_mov_dpp(own, 0xB1, 0xf, 0xf, true);
                    const unsigned msoff = prow_i * (unsigned)(p.W >> 1) * (unsigned)(COUT / 8) * 4u;
                    __builtin_amdgcn_raw_buffer_store_b32((own & 0xffu) | ((nbr & 0xffu) << 8) | (((own >> 16) & 0xffu) << 16) | (((nbr >> 16) & 0xffu) << 24),
                                                          rs_mask, pm_voff + (unsigned)mt * 16u, msoff, 0);
                    if constexpr (NREG == 16)
                        __builtin_amdgcn_raw_buffer_store_b32(((own >> 8) & 0xffu) | (((nbr >> 8) & 0xffu) << 8) | (((own >> 24) & 0xffu) << 16) | (((nbr >> 24) & 0xffu) << 24),
                                                              rs_mask, pm_voff + (unsigned)mt * 16u + 8u, msoff, 0);
                }
                {
                    const unsigned ysoff = prow_i * (unsigned)(p.W >> 1) * (unsigned)C::CPB;
                    u32x4_t q0, q1;
                    q0[0] = pack2bf(v[0], v[1]); q0[1] = pack2bf(v[2], v[3]); q0[2] = pack2bf(v[4], v[5]); q0[3] = pack2bf(v[6], v[7]);
                    __builtin_amdgcn_raw_buffer_store_b128(q0, rs_pool, py_voff + (unsigned)mt * 64u, ysoff, 0);
                    if constexpr (NREG == 16) {
                        q1[0] = pack2bf(v[8], v[9]); q1[1] = pack2bf(v[10], v[11]); q1[2] = pack2bf(v[12], v[13]); q1[3] = pack2bf(v[14], v[15]);
                        __builtin_amdgcn_raw_buffer_store_b128(q1, rs_pool, py_voff + (unsigned)mt * 64u + 32u, ysoff, 0);
                    }
                }
                continue;
            }
            if constexpr (C::FRB) continue;                    // nothing is stored: the gradient w.r.t. x0 has been reduced
            uint4 o0 = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
            uint4 o1 = make_uint4(pack2bf(v[8], v[9]), pack2bf(v[10], v[11]), pack2bf(v[12], v[13]), pack2bf(v[14], v[15]));
            unsigned char* dst = yrow + yoff + mt * 64;
            if constexpr (C::RGB) {
                f32x16_t t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&rgbA[0], *(const bf16x8_t*)&o0, zero16, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&rgbA[1], *(const bf16x8_t*)&o1, t, 0, 0, 0);
                if constexpr (C::TEAM == 2) {
                    if (wave == 0) { rgbh[0] = t[0] + t[4] + rgbb[0]; rgbh[1] = t[1] + t[5] + rgbb[1]; rgbh[2] = t[2] + t[6] + rgbb[2]; }
                    else if (kh == 0) {
                        const unsigned a = lds0 + C::RGBX_OFF + (unsigned)(s & 1) * 384 + n31 * 4;
                        *(lds_wfp)(a) = t[0] + t[4]; *(lds_wfp)(a + 128) = t[1] + t[5]; *(lds_wfp)(a + 256) = t[2] + t[6];
                    }
                } else if (kh == 0 && pv) {
                    float* __restrict__ ro = p.rgb_out + ((size_t)b * 3 * p.H + gy) * p.W + gx;
                    const size_t plane = (size_t)p.H * p.W;
                    ro[0] = t[0] + t[4] + rgbb[0]; ro[plane] = t[1] + t[5] + rgbb[1]; ro[2 * plane] = t[2] + t[6] + rgbb[2];
                }
                if (p.rgb_skip_y) continue;
            }
            if (full_strip) {
                *(uint4*)dst = o0;
                if constexpr (NREG == 16) *(uint4*)(dst + 32) = o1;
            } else if (pv) {
                *(uint4*)dst = o0;
                if constexpr (NREG == 16) *(uint4*)(dst + 32) = o1;
            }
        }
        DGE_T(4);
    };

#ifdef DGE_SC_TIMING
    const int tjob_id = ((blockIdx.x & 7) * jobs_per_xcd + (blockIdx.x >> 3)) * C::TPW + team;
    __builtin_amdgcn_sched_barrier(0);
    const long long tj1 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
#endif
    unsigned npar = 0;                                                      // noise buffer of the current period (byte offset 0 / 1024)
    for (int s0_ = 0; s0_ < rows; s0_ += C::NR, npar ^= 1024u) {
        bool done = false;
        StaticFor<C::NR>::run([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if (!done) {
                // (pooled flavour: `rows` is even - an odd row follows its even row without a second test, so that the even row's values
                //  reach it in registers)
                if ((C::POOL && (I & 1)) || s0_ + I < rows) step(ic, s0_ + I, npar);
                else done = true;
            }
        });
    }
    if constexpr (C::RGB && C::TEAM == 2) {          // the last row's toRGB
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave == 0 && rows > 0) rgb_flush(rows - 1);
    }
    // no DMA may land after the wave has given its LDS back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef DGE_SC_TIMING
    // [job][4] int64 over the first bytes of the OUTPUT (the result is garbage there): entry, loop start, loop end, rows | realtime
    if (lane == 0 && wave == 0) {
        long long* tl = (long long*)p.y + (size_t)tjob_id * 4;
        tl[0] = tj0; tl[1] = tj1; tl[2] = __builtin_readcyclecounter(); tl[3] = ((long long)rows << 40) | (long long)(__builtin_amdgcn_s_memrealtime() & 0xffffffffffll);
    }
#endif

    // ---- statistics: reduce over the 32 pixel lanes, one atomic per channel per wave
    if constexpr (C::FRB) {
        // 32 sums per lane (8 channels x 4 components) over the 32 pixel lanes; lane n31 < 16 keeps channel 8 kh + (n31 & 7),
        // components n31 >> 3 (from s0) and 2 + (n31 >> 3) (from s1): two atomic instructions per wave
        float va = 0.f, vc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float a = pv ? s0[0][r] : 0.f, c = pv ? s1[0][r] : 0.f;
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) { a += __shfl_xor(a, m, 64); c += __shfl_xor(c, m, 64); }
            if (n31 == r) { va = a; vc = c; }
        }
        if (n31 < 16) {
            float* __restrict__ FR = p.fr_out + ((size_t)(blockIdx.x % p.stats_slots) * p.B + b) * COUT * 4 + (8 * kh + (n31 & 7)) * 4 + (n31 >> 3);
            atomicAdd(FR, va); atomicAdd(FR + 2, vc);
        }
    } else
    if constexpr (C::STATS || C::DOT) {
        if (p.stats || (C::INB && p.prep_stats)) {
            // lanes right of the image (ragged strips) accumulated values of pixels that do not exist: a lane is one pixel
            // column, so the whole column is dropped here instead of masking every step
#pragma unroll
            for (int mt = 0; mt < C::MTW; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++) { if (!pv) { s0[mt][r] = 0.f; s1[mt][r] = 0.f; if constexpr (C::PREP || C::INB) s2[r] = 0.f; } }
            float* __restrict__ ST = p.stats ? p.stats + (size_t)(blockIdx.x % p.stats_slots) * p.B * COUT * 2 + (size_t)b * COUT * 2 : nullptr;
            float* __restrict__ PST = ((C::PREP || C::INB) && p.prep_stats) ? p.prep_stats + (size_t)(blockIdx.x % p.stats_slots) * p.B * COUT * 2 + (size_t)b * COUT * 2 : nullptr;
            // deterministic mode: domain = sample, slot = (segment, strip), the team's waves fill disjoint channel ranges of it
            const bool det = det_on();
            float* dvec = det ? det_slot(b, p.B, seg * nstrips + strip, nseg * nstrips, COUT * 2) : nullptr;
            // after the butterflies every lane holds the totals: lane n31 = r keeps those of register r, so that ONE atomic
            // instruction carries all channels of the tile (an instruction per channel with two active lanes costs an L2 request
            // each: ~10 requests per ns on the whole device, 30-60 per wave here)
#pragma unroll
            for (int mt = 0; mt < C::MTW; mt++) {
                float va = 0.f, vc = 0.f, vt = 0.f;
#pragma unroll
                for (int r = 0; r < NREG; r++) {
                    float a = s0[mt][r], c = s1[mt][r], t2 = (C::PREP || C::INB) ? s2[r] : 0.f;
#pragma unroll
                    for (int m = 1; m < 32; m <<= 1) {
                        a += __shfl_xor(a, m, 64); c += __shfl_xor(c, m, 64);
                        if constexpr (C::PREP || C::INB) t2 += __shfl_xor(t2, m, 64);
                    }
                    if (n31 == r) { va = a; vc = c; vt = t2; }
                }
                if (n31 < NREG) {
                    const int ch = wave * C::MTW * 32 + chan_of_reg(mt, n31);
                    if constexpr (C::INB) {         // (sum g_pre, sum g_pre*noise); not offered in deterministic mode
                        if (PST) { atomicAdd(PST + ch * 2, vc); atomicAdd(PST + ch * 2 + 1, vt); }
                    } else if constexpr (C::PREP) {        // (the launcher refuses prep in deterministic mode)
                        atomicAdd(ST + ch * 2, va);
                        if (PST) { atomicAdd(PST + ch * 2, vt); atomicAdd(PST + ch * 2 + 1, vc); }
                    } else if (det) { dvec[ch * 2] = va; dvec[ch * 2 + 1] = vc; }
                    else { atomicAdd(ST + ch * 2, va); atomicAdd(ST + ch * 2 + 1, vc); }
                }
            }
            if (!C::PREP && !C::INB && det && det_arrive_wave(b, nseg * nstrips * C::TEAM)) {
                // last wave of the sample: ordered sum of all slots into copy 0 of the statistics buffer
                for (int idx = lane; idx < COUT * 2; idx += 64)
                    p.stats[(size_t)b * COUT * 2 + idx] = det_sum(b, nseg * nstrips, COUT * 2, idx);
            }
        }
    }
}

template <int CIN, int COUT, int FL>
int launch_stream(const ConvParams& p0, hipStream_t s) {
    using C = SC<CIN, COUT, FL>;
    ConvParams p = p0;
    auto kern = conv_stream_kernel<CIN, COUT, FL>;
    static int caps[16] = {0};                                   // resident workgroups of this instantiation, per device (so is the attribute)
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& cap = caps[dev >= 0 && dev < 16 ? dev : 0];
    if (!cap) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES * C::TPW);
        int occ = 0, ncu = 256;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, 64 * C::TEAM * C::TPW, C::LDS_BYTES * C::TPW) != hipSuccess || occ < 1) occ = 1;
        cap = occ * ncu * C::TPW;
    }
    const int nstrips = (p.W + 31) / 32;
    // segments: all workgroups resident at once when the strips alone do not fill the device; at least 16 rows each
    int nseg = cap / (p.B * nstrips);
    if (dge_env().stream_nseg > 0) nseg = dge_env().stream_nseg;
    int maxseg = p.H / 16; if (maxseg < 1) maxseg = 1;
    if (nseg > maxseg) nseg = maxseg;
    if (nseg < 1) nseg = 1;
    int seg_rows = (p.H + nseg - 1) / nseg;
    if (C::POOL) seg_rows += seg_rows & 1;                       // row pairs stay inside a segment
    nseg = (p.H + seg_rows - 1) / seg_rows;
    const int njobs = p.B * nstrips * nseg;
    const int nwg = (njobs + C::TPW - 1) / C::TPW;
    const int jobs_per_xcd = (nwg + 7) / 8;                        // workgroups per XCD
    const char* fl = FL == FL_GEN ? "gen" : (FL == FL_ENC ? "enc" : (FL == FL_ENC_STATS ? "enc_stats" : (FL == FL_DOT ? "dot" : (FL == FL_DOT_PREP ? "dot_prep" : (FL == FL_GEN_RGB ? "gen_rgb" : (FL == FL_DOT_IN ? "dot_in" : (FL == FL_DOT_FR ? "dot_fromrgb" : (FL == FL_DOT_INX ? "dot_inx" : "enc_pool"))))))));
    dge_note_kernel("conv_stream<bf16,%d,%d,%s>", CIN, COUT, fl);
    hipLaunchKernelGGL(kern, dim3((unsigned)(jobs_per_xcd * 8)), dim3(64 * C::TEAM * C::TPW), C::LDS_BYTES * C::TPW, s, p, nstrips, nseg, seg_rows, njobs, jobs_per_xcd);
    DGE_LAUNCH_CHECK("conv_stream");
    return 0;
}

template <int CIN, int COUT>
int launch_flavour(const ConvParams& p, hipStream_t s) {
    if (p.in_coef && p.fr_out) {
        if constexpr (CIN == 16 && COUT == 16) return launch_stream<CIN, COUT, FL_DOT_FR>(p, s);
        dge_set_error("conv_stream: the FromRGB reduction epilogue is built for 16 -> 16 only");
        return -1;
    }
    if (p.in_coef && !p.prep_stats) {
        if constexpr (CIN == COUT && CIN >= 32) return launch_stream<CIN, COUT, FL_DOT_INX>(p, s);
        dge_set_error("conv_stream: the block-input instance-norm backward epilogue is built for 32 -> 32 and 64 -> 64 only");
        return -1;
    }
    if (p.in_coef) {
        if constexpr ((CIN == 32 && COUT == 16) || (CIN == 64 && COUT == 32)) return launch_stream<CIN, COUT, FL_DOT_IN>(p, s);
        dge_set_error("conv_stream: the instance-norm backward epilogue is built for 32 -> 16 and 64 -> 32 only");
        return -1;
    }
    if (p.dot_src) {
        if constexpr (CIN == COUT && CIN >= 32) {        // the generator's stride-1 layers at 512^2 / 1024^2 (64 / 32 channels)
            if (p.prep) return launch_stream<CIN, COUT, FL_DOT_PREP>(p, s);
        }
        if (p.prep) { dge_set_error("conv_stream: the fused tail backward is built for 32->32 and 64->64 only"); return -1; }
        return launch_stream<CIN, COUT, FL_DOT>(p, s);
    }
    const bool enc = p.stats || p.in_shift || (p.noise && p.noise_w_stride != 0);
    if (p.rgb_out) {
        if constexpr ((CIN == 32 && COUT == 32) || (CIN == 64 && COUT == 64)) { if (!enc) return launch_stream<CIN, COUT, FL_GEN_RGB>(p, s); }
        dge_set_error("conv_stream: the fused toRGB is built for the generator flavour of 32 -> 32 and 64 -> 64 only");
        return -1;
    }
    if (p.pool_out) {
        if constexpr ((CIN == 16 && COUT == 32) || (CIN == 32 && COUT == 64)) { if (!p.stats) return launch_stream<CIN, COUT, FL_ENC_POOL>(p, s); }
        dge_set_error("conv_stream: the pooled epilogue is built for the encoder's 16 -> 32 and 32 -> 64 layers without statistics");
        return -1;
    }
    if constexpr (CIN == 64) {       // the encoder flavours of the 64-channel input do not fit the register file next to 144 weight registers
        if (enc) { dge_set_error("conv_stream: encoder flavour with Cin = 64 is not built"); return -1; }
    } else {
        if (p.stats) return launch_stream<CIN, COUT, FL_ENC_STATS>(p, s);
        if (enc) return launch_stream<CIN, COUT, FL_ENC>(p, s);
    }
    return launch_stream<CIN, COUT, FL_GEN>(p, s);
}

}  // namespace

// Eligibility of the streaming kernel for a launch (bf16, 3x3, stride 1, no fused resampling / addend / ReLU prologue).
bool dge_conv_stream_eligible(const ConvParams& p, int dtype, int ksize) {
    if (dtype != DGE_BF16 || ksize != 3 || p.up || p.in_s2d || p.in_up2 || p.in_relu || p.addend || p.mask_relu || p.in_t2d) return false;
    if (!(p.Cin == 16 || p.Cin == 32 || p.Cin == 64) || !(p.Cout == 16 || p.Cout == 32 || p.Cout == 64)) return false;
    if (p.W % 4 != 0 || p.W < 64 || p.H < 32) return false;
    if ((long)p.H * p.W < 128L * 128) return false;
    {   // enough rows per wave to amortise the weight fold (and the shift table of the encoder flavours): measured break-even
        // against conv_igemm (profiles/r02_conv_stream_*): data gradient always, generator >= 2^18 pixels, encoder >= 2^20
        const long px = (long)p.B * p.H * p.W;
        const bool enc = p.stats || p.in_shift || (p.noise && p.noise_w_stride != 0);
        if (!dge_env().force_stream) {           // (tests run the kernel on small ragged shapes)
            if (!p.dot_src && px < (enc ? (1L << 20) : (1L << 18))) return false;
            if (p.dot_src && px < (1L << 16)) return false;
        }
    }
    if ((long)p.W * p.Cin * 2 >= (1L << 31) || (long)p.H * p.W * 64 * 2 >= (1L << 40)) return false;
    if (p.dot_src && !p.stats && !p.in_coef) return false;
    if (p.in_coef && !p.fr_out && p.prep_stats && !(p.dot_src && !p.prep && !p.stats && ((p.Cin == 32 && p.Cout == 16) || (p.Cin == 64 && p.Cout == 32)))) return false;
    if (p.in_coef && !p.fr_out && !p.prep_stats && !(p.dot_src && !p.prep && !p.stats && p.Cin == p.Cout && p.Cin >= 32 && p.H % 2 == 0 && p.W % 2 == 0)) return false;
    if (p.fr_out && !(p.in_coef && p.dot_src && p.fr_img4 && !p.prep && !p.stats && p.Cin == 16 && p.Cout == 16 && p.H % 2 == 0 && p.W % 2 == 0)) return false;
    if (p.prep && !(p.Cin == p.Cout && p.Cin >= 32)) return false;
    if (p.Cin == 64 && !p.dot_src && (p.stats || p.in_shift || (p.noise && p.noise_w_stride != 0))) return false;
    if (p.dot_src && (p.bias || p.noise || p.in_shift || p.act != DGE_ACT_NONE)) return false;
    if (p.noise && p.noise_w == nullptr) return false;
    if (dge_env().no_stream) return false;
    return true;
}

// The pooled epilogue (ConvParams::pool_out) exists in the streaming kernel only, for conv_2 of the first two encoder blocks
bool dge_conv_pool_ok(const ConvParams& p, int dtype, int ksize) {
    return ((p.Cin == 16 && p.Cout == 32) || (p.Cin == 32 && p.Cout == 64)) && !p.dot_src && !p.stats && !p.rgb_out && p.H % 2 == 0 && p.W % 2 == 0 &&
           dge_conv_stream_eligible(p, dtype, ksize);
}

// The fused toRGB epilogue (ConvParams::rgb_*) exists in the streaming kernel only, for the 32 -> 32 and 64 -> 64 generator flavours
bool dge_conv_rgb_ok(const ConvParams& p, int dtype, int ksize) {
    return ((p.Cin == 32 && p.Cout == 32) || (p.Cin == 64 && p.Cout == 64)) && !p.dot_src && !p.stats && !p.in_shift && !(p.noise && p.noise_w_stride != 0) &&
           dge_conv_stream_eligible(p, dtype, ksize);
}

int dge_conv_stream_launch(const ConvParams& p, hipStream_t s) {
#define GO(CI, CO) if (p.Cin == CI && p.Cout == CO) return launch_flavour<CI, CO>(p, s)
#ifdef DGE_SC_ONLY          // tuning builds: one channel configuration (compile time)
#define GO2(...) GO(__VA_ARGS__)
    GO2(DGE_SC_ONLY);
#undef GO2
#else
    GO(16, 16); GO(16, 32); GO(16, 64); GO(32, 16); GO(32, 32); GO(32, 64); GO(64, 16); GO(64, 32); GO(64, 64);
#endif
#undef GO
    dge_set_error("conv_stream: unsupported channel configuration %d -> %d", p.Cin, p.Cout);
    return -1;
}
