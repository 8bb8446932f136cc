// 3x3 convolution of the LOW-RESOLUTION layers (4^2 .. 16^2 / 24^2 at 256 - 512 channels) for gfx950, bf16, NHWC.
//
// These layers (StyleGAN2 layers 0-4, stylegan2_generator.py:488-490; the 512-channel encoder blocks at <= 16^2, E.py:97-117;
// LPIPS conv5_x on the cropped images) have a few hundred pixels per sample and 4.7 MB of weights: every workgroup of an
// implicit GEMM has to pull its 64-row weight slice (590 KB) through one compute unit.  The general kernel (conv_igemm.hip)
// does that through a 2-deep LDS ring with a barrier per stage and reaches ~22 GB/s per CU (28 us per launch whatever the
// resolution); tools/probes/probe_l2bw.hip shows that one CU can pull 85-110 GB/s out of L2 when enough loads are in flight.
// This kernel is built around that stream:
//
//   * one workgroup = 8x8 output pixels x 64 output channels, 8 waves; the halo tile of ALL input channels (10 x 10 pixels x
//     Cin, <= 104 KB) is staged into LDS ONCE, with the fused prologue affine (style modulation / instance-norm apply);
//   * the weights never touch LDS: they are packed in MFMA-fragment order (DGE_PACK_FRAG, s2_kernels.hip: pack_out_index),
//     every wave streams ITS share - a 32-channel half of the N tile x a quarter of the input channels - straight into
//     registers with fully coalesced 1 KiB loads, one tap (8 loads) ahead of the MFMAs that consume them; waves never wait for
//     each other in the main loop (no barrier, no LDS ring): 8 waves x 8-16 KB in flight per CU;
//   * the four K-quarter partial sums of a 32 x 32 output tile are added through LDS (the dead halo tile) and waves 0-3 run the
//     family's shared epilogue (conv_epilogue.h): demodulation / noise / bias / activation / residual addend / statistics /
//     data-gradient dot products / depth-to-space store of the folded up layer.
//
// Reference math: model/stylegan2_generator.py:855-922, model/E/E.py:50-85 (same call sites as conv_igemm.hip).
#include <stdlib.h>
#include "common.h"
#include "conv_params.h"
#include "conv_epilogue.h"

namespace {

struct SmallCfg {                                   // what conv_epilogue.h needs: one 32x32 tile per carrier wave, 2 x 2 carriers
    static constexpr int MT = 1, NT = 1, WTM = 32, WTN = 32, BM = 64;
    static constexpr int ESTR = 32 * 4 + 16;
};
constexpr int TH = 8, TW = 8, BN = 64, HH = 10, HW = 10, CIN_MAX = 512;
// pixel pitch 2*Cin + 16 and row pitch + 224 bytes: the A-fragment ds_read_b128 of 32 pixels (4 rows of 8) x 2 K halves is
// conflict free in every lane group of the instruction (searched offline over both paddings)
__host__ __device__ constexpr int pstr_of(int cin) { return 2 * cin + 16; }
__host__ __device__ constexpr int rpitch_of(int cin) { return HW * pstr_of(cin) + 224; }
constexpr int A_MAX = HH * rpitch_of(CIN_MAX);                 // 106,240 B
constexpr int RED_BYTES = 8 * 2 * 16 * 64 * 4;                 // K-quarter partial sums of all waves: 64 KB (over the dead halo tile)
constexpr int EST_BYTES = 4 * 32 * SmallCfg::ESTR;             // epilogue staging of the 4 carrier waves
constexpr int N_BYTES = 4 * SmallCfg::BM * 4;                  // noise tile (x4 phases in up mode)
constexpr int PF_BYTES = 8 * 256;                              // landing zone of the next launch's weight prefetch (never read)
constexpr int LDS_BYTES = A_MAX + N_BYTES + PF_BYTES;
static_assert(RED_BYTES + EST_BYTES <= A_MAX, "reduction + staging must fit the halo region");

__device__ __forceinline__ unsigned lds_base_u32(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
__device__ __forceinline__ f32x16_t mma(const uint4& a, const uint4& b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&a, *(const bf16x8_t*)&b, c, 0, 0, 0);
}

// KQ = K steps (16 channels) per wave and tap = Cin / 64
template <int KQ, int MODE>
__global__ __launch_bounds__(512, 2) void conv_small_kernel(ConvParams p) {
    constexpr int CIN = KQ * 64, CH = CIN / 8;                 // CH = 16-byte chunks per pixel
    constexpr int PSTR = pstr_of(CIN), RPITCH = rpitch_of(CIN);
    constexpr int KST = CIN / 16;                              // K steps per tap
    __shared__ __attribute__((aligned(256))) unsigned char lds[LDS_BYTES];
    float* ldsN = (float*)(lds + A_MAX);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile order (as conv_igemm): workgroup i runs on XCD i % 8; every XCD gets a contiguous range of tiles, so the
    // workgroups that share a weight slice (same N tile, different samples / pixel tiles) share that XCD's L2
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, per = nb >> 3;
        if (per > 0 && bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
    }
    const int vbid = bid;
    const int tx_i = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty_i = bid % p.tiles_y; bid /= p.tiles_y;
    const int b = bid % p.B;
    const int ntile = bid / p.B;
    const int x0 = tx_i * TW, y0 = ty_i * TH, bn0 = ntile * BN;
    const int nt = wave & 1, kq = wave >> 1;                   // this wave's half of the N tile and quarter of the input channels

    // ---- weight stream: fragment blocks ((tap * N/32 + n32) * KST + kstep) of 1 KiB, lane-linear
    const unsigned char* wbase = (const unsigned char*)p.w + ((size_t)(bn0 / 32 + nt) * KST + kq * KQ) * 1024 + lane * 16;
    const size_t tap_stride = (size_t)(p.Ntot / 32) * KST * 1024;
    // PF taps (PF * KQ KiB per wave) are requested ahead of the MFMAs.  In a training step the weights of a layer are L2-cold (they
    // come from HBM at ~2.5 us): 8 waves x 24 KB in flight per CU; measured in-step 25 us with PF = 2 against 14 us on hot weights
    constexpr int PF = KQ >= 8 ? 3 : 4;
    uint4 bw[PF + 1][KQ];
#pragma unroll
    for (int t = 0; t < PF; t++)
#pragma unroll
        for (int j = 0; j < KQ; j++) bw[t][j] = *(const uint4*)(wbase + t * tap_stride + j * 1024);   // fly under the halo staging

    // ---- L2 warm-up for the next low-resolution launch of the stream (ConvParams::pf_w): this workgroup runs on XCD blockIdx.x % 8,
    // and so will the workgroups of that launch that read N tiles [x*nt/8, (x+1)*nt/8) (same XCD-aware order); the workgroups of
    // an XCD share the lines of those slices, one 128-byte line per lane, as LDS-DMA into a dummy zone (no register is written,
    // nothing waits.  A one-round grid issues them behind its main loop (no wait of its own weight stream includes them:
    // 23.8 -> 17.6 us on HBM-cold weights at 16^2, batch 8; hot 17.2), a multi-round grid up front (38.6 -> 33.8 us at batch 16); s_waitcnt vmcnt(0) at the end of the kernel).
    auto prefetch_next = [&]() {
        const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3, R = max(1, (int)gridDim.x >> 3);
        const int nt_n = p.pf_ntot / BN, kst_n = p.pf_cin / 16;
        const int n_lo = xcd * nt_n / 8, n_hi = max(n_lo + 1, (xcd + 1) * nt_n / 8);
        const unsigned lines_tap = (unsigned)(n_hi - n_lo) * 2u * kst_n * 8u;            // 128-byte lines of the slice, per tap
        const unsigned nlines = 9u * lines_tap;
        const size_t tap_stride_n = (size_t)(p.pf_ntot / 32) * kst_n * 1024;
        const unsigned m0v = lds_base_u32(lds) + A_MAX + N_BYTES + wave * 256;
        for (unsigned l = (unsigned)r * 512u + tid; l - lane < nlines; l += (unsigned)R * 512u) {   // (wave-uniform trip count)
            const unsigned tp = l / lines_tap, within = l - tp * lines_tap;
            const unsigned char* a = (const unsigned char*)p.pf_w + tp * tap_stride_n + (size_t)n_lo * 2 * kst_n * 1024 + (size_t)within * 128;
            if (l >= nlines) a = (const unsigned char*)p.pf_w;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(a), "s"(m0v) : "memory");
        }
    };
    const bool pf_late = gridDim.x <= 256;
    if (p.pf_w && !pf_late) prefetch_next();

    // ---- halo tile of all input channels: global -> registers -> (affine) -> LDS, zero outside the image
    {
        const bf16_t* __restrict__ Xb = (const bf16_t*)p.x + (size_t)b * p.H * p.W * CIN;
        const bool affine = p.in_scale || p.in_shift;
        const int chunk = tid % CH;                            // CH divides 512: every item of a thread has the same chunk
        float asc[8], ash[8];
        if (affine) {
            const int ci = b * CIN + chunk * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) { asc[e] = p.in_scale ? p.in_scale[ci + e] : 1.f; ash[e] = p.in_shift ? p.in_shift[ci + e] : 0.f; }
        }
        constexpr int NITEM = HH * HW * CH, NPER = (NITEM + 511) / 512;
        uint4 v[NPER];
        unsigned inmask = 0;
#pragma unroll
        for (int i = 0; i < NPER; i++) {
            const int idx = tid + i * 512;
            const int pix = idx / CH, hx = pix % HW, hy = pix / HW;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool inside = ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W) & (idx < NITEM);
            v[i] = make_uint4(0, 0, 0, 0);
            if (inside) v[i] = *(const uint4*)(Xb + (gy * p.W + gx) * CIN + chunk * 8);
            inmask |= (inside ? 1u : 0u) << i;
        }
        const float* __restrict__ nz_src = p.prep ? p.prep_noise : p.noise;      // (prep: the plane of the layer below)
        if (nz_src) {
            const int nz_bs = p.prep ? p.prep_noise_bstride : p.noise_bstride;
            const int OWn = p.up ? 2 * p.W : p.W;
            const int nph = p.up ? 4 : 1;
            for (int idx = tid; idx < nph * SmallCfg::BM; idx += 512) {
                const int m = idx % SmallCfg::BM, ph = idx / SmallCfg::BM;
                const int gy = y0 + m / TW, gx = x0 + m % TW;
                const int oy = p.up ? 2 * gy + (ph >> 1) : gy, ox = p.up ? 2 * gx + (ph & 1) : gx;
                ldsN[idx] = (gy < p.H && gx < p.W) ? nz_src[(size_t)b * nz_bs + (size_t)oy * OWn + ox] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < NPER; i++) {
            const int idx = tid + i * 512;
            if (idx < NITEM) {
                const int pix = idx / CH, hx = pix % HW, hy = pix / HW;
                uint4 q = v[i];
                if (affine && ((inmask >> i) & 1u)) {          // padding stays zero (it follows the affine in the reference)
                    float f[8];
                    unpack16(q, f, (bf16_t*)nullptr);
#pragma unroll
                    for (int e = 0; e < 8; e++) f[e] = f[e] * asc[e] + ash[e];
                    if (p.in_relu) {
#pragma unroll
                        for (int e = 0; e < 8; e++) f[e] = fmaxf(f[e], 0.f);
                    }
                    q = pack16(f, (bf16_t*)nullptr);
                }
                *(uint4*)(lds + hy * RPITCH + hx * PSTR + chunk * 16) = q;
            }
        }
    }
    __syncthreads();

    // ---- main loop: 9 taps x KQ K steps x 2 pixel groups, no synchronisation between waves
    f32x16_t acc[2];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[g][r] = 0.f;
    int aoff[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const int m = g * 32 + (lane & 31);
        aoff[g] = (m / TW) * RPITCH + (m % TW) * PSTR + kq * KQ * 32 + (lane >> 5) * 16;
    }
    StaticFor<9>::run([&](auto tc) {
        constexpr int tap = decltype(tc)::value;
        constexpr int dy = tap / 3, dx = tap % 3;
        if (tap + PF < 9) {
            const unsigned char* wn = wbase + (size_t)(tap + PF) * tap_stride;
#pragma unroll
            for (int j = 0; j < KQ; j++) bw[(tap + PF) % (PF + 1)][j] = *(const uint4*)(wn + j * 1024);
        }
        const unsigned char* at = lds + dy * RPITCH + dx * PSTR;
#pragma unroll
        for (int j = 0; j < KQ; j++) {
            const uint4 a0 = *(const uint4*)(at + aoff[0] + j * 32);
            const uint4 a1 = *(const uint4*)(at + aoff[1] + j * 32);
            acc[0] = mma(a0, bw[tap % (PF + 1)][j], acc[0]);
            acc[1] = mma(a1, bw[tap % (PF + 1)][j], acc[1]);
        }
    });

    if (p.pf_w && pf_late) prefetch_next();
    // ---- add the four K quarters of every 32 x 32 tile (through the dead halo region), hand the tiles to waves 0-3
    __syncthreads();
    float* red = (float*)lds;
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int r = 0; r < 16; r++) red[((wave * 2 + g) * 16 + r) * 64 + lane] = acc[g][r];
    __syncthreads();
    const bool carrier = wave < 4;
    f32x16_t tile[1][1];
    if (carrier) {
        const int wm = wave >> 1, wn = wave & 1;               // conv_epilogue's wave grid: pixel group, N half
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++) s += red[(((q * 2 + wn) * 2 + wm) * 16 + r) * 64 + lane];
            tile[0][0][r] = s;
        }
    }
    conv_epilogue<bf16_t, SmallCfg, TH, TW, BN, 2, 2, 512, MODE>(p, tile, lds + RED_BYTES, ldsN, b, x0, y0, bn0, ntile, vbid, tx_i, ty_i,
                                                            wave, lane, tid, carrier);
    if (p.pf_w) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the prefetch requests target this workgroup's LDS
}

int max_hw() {
    static const int v = [] { const char* e = getenv("DGE_SMALL_MAXHW"); const int n = e ? atoi(e) : 24; return n < 0 ? 0 : n; }();
    return v;
}

}  // namespace

bool dge_conv_small_shape_ok(int H, int W, int cin, int ntot, int ksize, int in_s2d, int in_up2, int dtype) {
    return dtype == DGE_BF16 && ksize == 3 && !in_s2d && !in_up2 && (cin == 512 || cin == 256) && ntot % BN == 0 && ntot >= BN &&
           H <= max_hw() && W <= max_hw();
}

extern "C" int dge_conv_small_supported(int H, int W, int cin, int ntot, int ksize, int in_s2d, int in_up2, int dtype) {
    return dge_conv_small_shape_ok(H, W, cin, ntot, ksize, in_s2d, in_up2, dtype) ? 1 : 0;
}

int dge_conv_small_launch(const ConvParams& p0, hipStream_t s) {
    ConvParams p = p0;
    p.dbg = 0;
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    const long grid = (long)p.tiles_x * p.tiles_y * p.B * (p.Ntot / BN);
    DGE_CHECK(grid > 0 && grid < (1L << 31), "conv_small: bad grid");
    if (p.stats && !p.up)
        DGE_CHECK(dge_det_fits((long long)p.B * (p.Ntot / BN), (long long)p.tiles_x * p.tiles_y * 2, BN * 2),
                  "conv_small: the deterministic-mode workspace is too small for this launch");
    dge_note_kernel("conv_small<bf16,8,8,64,%d>%s", p.Cin, p.prep ? "+prep" : "");
    const int mode = p.prep ? 2 : ((p.addend || p.dot_src) ? 1 : 0);      // epilogue mode (conv_epilogue.h)
#define DGE_GO(KQ, MODE) hipLaunchKernelGGL((conv_small_kernel<KQ, MODE>), dim3((unsigned)grid), dim3(512), 0, s, p)
    if (p.Cin == 512) { if (mode == 2) DGE_GO(8, 2); else if (mode == 1) DGE_GO(8, 1); else DGE_GO(8, 0); }
    else { if (mode == 2) DGE_GO(4, 2); else if (mode == 1) DGE_GO(4, 1); else DGE_GO(4, 0); }
#undef DGE_GO
    DGE_LAUNCH_CHECK("conv_small");
    return 0;
}
