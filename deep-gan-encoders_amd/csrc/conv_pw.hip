// Pointwise (1x1) convolution for the narrow high-resolution layers, bf16, gfx950: the residual branch conv_3 of the encoder's
// BEBlock (reference model/E/E.py:44-45,80-84: x = 0.111*x + 0.889*conv_3(downscale2d(residual))) and its data gradient.
//
// With Cin <= 64 a 1x1 convolution is a per-pixel mat-vec: 32 .. 256 flop per byte, far under the ridge.  conv_igemm ran it
// through its tile machinery (halo-less 16x16 tiles, LDS staging, LDS-transposed epilogue) at 2.5-3.5x the byte floor
// (16 -> 32 @512^2 with addend + statistics: 180 us against 335 MB = 67 us).  Here there is no LDS at all:
//   * the weights of one 32-channel M tile sit in registers as MFMA A operands (Cin/16 fragments);
//   * the B operand is read STRAIGHT from global memory: lane (pixel n, half h) of a 32-pixel group loads the 16 bytes
//     x[pixel][16*ks + 8*h ..] - a wave instruction reads 32 pixels x Cin contiguous bytes;
//   * the rows of the M tile are permuted as in conv_stream (A row 8j + 4h + i holds channel 16(j>>1) + 8h + 4(j&1) + i) so that
//     the 16 accumulators of a lane are two runs of 8 consecutive channels of its pixel and the two K-half lanes of a pixel
//     write 32 contiguous bytes per instruction: addend in, result out as 16-byte vectors, no transpose;
//   * bias / gain / residual addend / per-(b, channel) statistics of the result in registers; one atomic per channel and wave.
// A workgroup's four waves own the M tiles of one pixel range (Cout = 128) or four pixel ranges (Cout = 32).
#include "common.h"
#include "conv_params.h"
#include "../../include/dge_hip.h"

namespace {

template <int CIN>
__global__ __launch_bounds__(256) void conv_pw_kernel(ConvParams p, int nmt, int groups_per_sample) {
    constexpr int KS = CIN / 16;
    __shared__ float red[256];                         // per-channel (sum, sum of squares) of the workgroup
    if (p.stats) { red[threadIdx.x] = 0.f; __syncthreads(); }
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int mt = wave % nmt;                         // this wave's 32-channel tile
    const int sub = wave / nmt, nsub = 4 / nmt;        // waves of a workgroup that share a tile take different pixel groups
    const int HW = p.H * p.W;
    const bf16_t* __restrict__ X = (const bf16_t*)p.x + (size_t)b * HW * CIN;
    const bf16_t* __restrict__ Wp = (const bf16_t*)p.w;
    bf16_t* __restrict__ Y = (bf16_t*)p.y + (size_t)b * HW * p.Cout;
    const bf16_t* __restrict__ ADD = p.addend ? (const bf16_t*)p.addend + (size_t)b * HW * p.Cout : nullptr;
    // A operand: physical row m = l31 = 8j + 4h + i of the tile is channel 16(j>>1) + 8h + 4(j&1) + i
    const int crow = mt * 32 + 16 * (l31 >> 4) + 8 * ((l31 >> 2) & 1) + 4 * ((l31 >> 3) & 1) + (l31 & 3);
    uint4 a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) a[ks] = *(const uint4*)(Wp + (size_t)crow * CIN + ks * 16 + 8 * lh);     // (rows beyond Cout are zero-padded)
    const int c0 = mt * 32 + 8 * lh, c1 = c0 + 16;     // this lane's two runs of 8 consecutive output channels (registers 0-7, 8-15)
    const bool cv0 = c0 < p.Cout, cv1 = c1 < p.Cout;
    float bia[16], s0[16], s1[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const bool cv = r < 8 ? cv0 : cv1;
        bia[r] = (p.bias && cv) ? p.bias[(r < 8 ? c0 : c1) + (r & 7)] * p.bias_scale * p.gain : 0.f;
        s0[r] = 0.f; s1[r] = 0.f;
    }
    const int ngroups = (HW + 31) / 32;
    const int gstride = gridDim.x * nsub;
    auto load_x = [&](int g, uint4 (&xv)[KS]) {
        const int pix = g * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
            xv[ks] = pix < HW ? *(const uint4*)(X + (size_t)pix * CIN + ks * 16 + 8 * lh) : make_uint4(0, 0, 0, 0);
    };
    auto load_add = [&](int g, uint4 (&av)[2]) {
        const int pix = g * 32 + l31;
        av[0] = make_uint4(0, 0, 0, 0); av[1] = av[0];
        if (ADD && pix < HW) {
            if (cv0) av[0] = *(const uint4*)(ADD + (size_t)pix * p.Cout + c0);
            if (cv1) av[1] = *(const uint4*)(ADD + (size_t)pix * p.Cout + c1);
        }
    };
    int g = blockIdx.x * nsub + sub;
    uint4 xv[2][KS], av[2][2];
    if (g < ngroups) { load_x(g, xv[0]); load_add(g, av[0]); }
    int par = 0;
    for (; g < ngroups; g += gstride, par ^= 1) {
        const int gn = g + gstride;
        // the next group's operands are requested before this group's arithmetic (two register sets, compile-time indexed below)
        if (par == 0) { if (gn < ngroups) { load_x(gn, xv[1]); load_add(gn, av[1]); } }
        else          { if (gn < ngroups) { load_x(gn, xv[0]); load_add(gn, av[0]); } }
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        uint4 ad0, ad1;
        if (par == 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&a[ks], *(const bf16x8_t*)&xv[0][ks], acc, 0, 0, 0);
            ad0 = av[0][0]; ad1 = av[0][1];
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&a[ks], *(const bf16x8_t*)&xv[1][ks], acc, 0, 0, 0);
            ad0 = av[1][0]; ad1 = av[1][1];
        }
        const int pix = g * 32 + l31;
        const bool pv = pix < HW;
        float f[16], ad[16];
        unpack16(ad0, *(float(*)[8])&ad[0], (bf16_t*)nullptr);
        unpack16(ad1, *(float(*)[8])&ad[8], (bf16_t*)nullptr);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            f[r] = fmaf(acc[r], p.gain, bia[r]);
            f[r] = fmaf(p.add_scale, ad[r], f[r]);
            const float fs = (pv && (r < 8 ? cv0 : cv1)) ? f[r] : 0.f;
            s0[r] += fs; s1[r] = fmaf(fs, fs, s1[r]);
        }
        if (pv) {
            if (cv0) *(uint4*)(Y + (size_t)pix * p.Cout + c0) = pack16(*(float(*)[8])&f[0], (bf16_t*)nullptr);
            if (cv1) *(uint4*)(Y + (size_t)pix * p.Cout + c1) = pack16(*(float(*)[8])&f[8], (bf16_t*)nullptr);
        }
    }
    if (p.stats) {
        // the (sum, sum of squares) of a channel: 32 pixel lanes by shuffles, the waves of the workgroup that share the channel
        // through LDS, then ONE atomic per channel and workgroup (f32 atomics reach L2 at ~10 per ns on the whole device: with one
        // per channel and wave the statistics cost more than the convolution)
        float* __restrict__ ST = p.stats + (size_t)(blockIdx.x % p.stats_slots) * p.B * p.Cout * 2;
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int msk = 1; msk < 32; msk <<= 1) { s0[r] += __shfl_xor(s0[r], msk, 64); s1[r] += __shfl_xor(s1[r], msk, 64); }
            if (l31 == 0) {
                const int c = (r < 8 ? c0 : c1) + (r & 7);
                atomicAdd(&red[c * 2], s0[r]);
                atomicAdd(&red[c * 2 + 1], s1[r]);
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 2 * p.Cout; idx += 256) atomicAdd(ST + (size_t)b * p.Cout * 2 + idx, red[idx]);
    }
}

}  // namespace

bool dge_conv_pw_eligible(const ConvParams& p, int dtype, int ksize) {
    if (dge_env().no_pw || dtype != DGE_BF16 || ksize != 1) return false;
    if (p.up || p.in_s2d || p.in_up2 || p.in_t2d || p.in_relu || p.in_scale || p.in_shift || p.out_scale || p.noise || p.dot_src || p.prep ||
        p.mask_relu || p.w_frag || p.act != DGE_ACT_NONE)
        return false;
    if (!(p.Cin == 16 || p.Cin == 32 || p.Cin == 64) || p.Cout % 16 || p.Cout > 128 || p.Ntot % 32 || p.Ntot > 128 || p.Ntot == 96) return false;
    if (p.stats && dge_get_deterministic()) return false;              // (statistics through plain atomics)
    // the narrow layers at >= 256^2 (batch 8; 64 -> 128 @128^2 measured 42 us here, 43 us there); the rest stay on conv_igemm (DGE_FORCE_STREAM: the small ragged shapes of the tests)
    return dge_env().force_stream || (long)p.B * p.H * p.W >= (1L << 18);
}

template <int CIN>
static int launch_pw(const ConvParams& p, hipStream_t s) {
    const int nmt = p.Ntot / 32;                                        // 1, 2 or 4 M tiles = waves per pixel range
    const int nsub = 4 / nmt;
    const int HW = p.H * p.W, ngroups = (HW + 31) / 32;
    auto kern = conv_pw_kernel<CIN>;
    static int caps[16] = {0};                                          // resident workgroups of this instantiation, per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& cap = caps[dev >= 0 && dev < 16 ? dev : 0];
    if (!cap) {
        int occ = 0, ncu = 256;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, 256, 0) != hipSuccess || occ < 1) occ = 1;
        cap = occ * ncu;
    }
    // persistent workgroups, one generation: every workgroup pays the weight fetch and 2 * Cout statistics atomics once
    int gx = (ngroups + nsub - 1) / nsub;
    const int per_sample = (cap + p.B - 1) / p.B;
    if (gx > per_sample) gx = per_sample;
    if (gx < 1) gx = 1;
    dge_note_kernel("conv_pw<bf16,%d,%d>", p.Cin, p.Ntot);
    hipLaunchKernelGGL(kern, dim3(gx, p.B), dim3(256), 0, s, p, nmt, 0);
    DGE_LAUNCH_CHECK("conv_pw");
    return 0;
}

int dge_conv_pw_launch(const ConvParams& p, hipStream_t s) {
    if (p.Cin == 16) return launch_pw<16>(p, s);
    if (p.Cin == 32) return launch_pw<32>(p, s);
    return launch_pw<64>(p, s);
}
