// Backward kernels of the encoder E.BE (reference model/E/E.py:50-85 differentiated):
// conv weight gradient (MFMA, split over pixel tiles, f32 atomics), instance-norm / statistics
// backward, activation backward with bias / noise-weight reductions, FromRGB and dense-layer
// weight gradients.  The conv data gradients reuse conv_igemm (DGE_PACK_DGRAD).
#include <stdlib.h>
#include "common.h"
#include "../../include/dge_hip.h"

// ------------------------------------------------------------------ conv weight gradient
// dW[o][i][tap] += sum_{b,p} g[b,p,o] * Xn[b,p+tap,i],  Xn = X*sc[b,i] + sh[b,i] inside the image, 0 outside.
// GEMM view: M = o (32 per block), N = i (32 per block), K = pixels of an 8x16 tile.
// MFMA wants K-contiguous fragments but NHWC has the channels contiguous, so both tiles are
// TRANSPOSED on their way into LDS ([channel][row][col], 16-bit scatter writes) and fragments are
// read with aligned 16-byte LDS reads.  The three x-shifts of a kernel row are produced in
// registers from one aligned window (v_alignbit for dx=1, register renaming for dx=2), so one
// window read feeds 3 MFMAs.  Each wave owns two tile rows and ALL taps (9 accumulator tiles), the
// g fragment of a row is reused by its 9 taps; waves never exchange data and flush with f32 atomics.
// Flush of the per-wave 32x32 accumulator tiles of all taps: the 4 waves' partial sums are combined in
// LDS and staged as [o][i][tap] -- the layout of dW -- so that consecutive lanes add to consecutive
// addresses (one f32 atomic per output element per workgroup, 64 contiguous floats per wave instruction;
// lane-strided atomics on the 36-byte tap pitch were 10x slower than the whole MFMA phase).
// `smem` must hold (4*1024 + 1024*NTAP) floats.
template <int NTAP>
__device__ __forceinline__ void wgrad_flush(const f32x16_t (&acc)[NTAP], float* smem, float* __restrict__ dW, int o0, int i0,
                                            int Co, int Ci, int tid, int lane, int wave) {
    float* red = smem;
    float* stage = smem + 4096;
#pragma unroll
    for (int t = 0; t < NTAP; t++) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave * 1024 + r * 64 + lane] = acc[t][r];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int e = tid + q * 256;                  // e = r*64 + lane'
            const int r = e >> 6, ln = e & 63;
            const int ol = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), il = ln & 31;
            stage[(ol * 32 + il) * NTAP + t] = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
        }
    }
    __syncthreads();
    const int ni = (Ci - i0 < 32 ? Ci - i0 : 32) * NTAP;  // valid floats of one o row of this tile (contiguous in dW)
    if (det_on()) {
        // deterministic mode: domain = the (o, i) tile (blockIdx.x), slot = the pixel-tile group (blockIdx.y); the last group to
        // arrive adds the ordered sum of all groups to dW (single writer: dW may already hold the other phase's gradient)
        const int L = 1024 * NTAP, nslots = gridDim.y;
        float* slot = det_slot(blockIdx.x, gridDim.x, blockIdx.y, nslots, L);
        for (int idx = tid; idx < L; idx += 256) slot[idx] = stage[idx];
        if (det_arrive_wg(blockIdx.x, nslots)) {
            for (int idx = tid; idx < L; idx += 256) {
                const int ol = idx / (32 * NTAP), j = idx - ol * (32 * NTAP);
                if (o0 + ol < Co && j < ni) dW[((size_t)(o0 + ol) * Ci + i0) * NTAP + j] += det_sum(blockIdx.x, nslots, L, idx);
            }
        }
        return;
    }
    for (int idx = tid; idx < 1024 * NTAP; idx += 256) {
        const int ol = idx / (32 * NTAP), j = idx - ol * (32 * NTAP);
        if (o0 + ol < Co && j < ni) atomicAdd(dW + ((size_t)(o0 + ol) * Ci + i0) * NTAP + j, stage[idx]);
    }
}

template <typename T, int KS>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const T* __restrict__ g, const T* __restrict__ X,
                                                          const float* __restrict__ sc, const float* __restrict__ sh,
                                                          float* __restrict__ dW, int B, int H, int W, int Co, int Ci,
                                                          int tiles_x, int tiles_y, int ntile_groups) {
    constexpr int TH = 8, TW = 16, HALO = KS / 2, HH = TH + 2 * HALO, HW = TW + 2 * HALO;
    constexpr int EP = Elem<T>::PER16;
    constexpr int ES = (int)sizeof(T);
    constexpr int NTAP = KS * KS;
    constexpr int GPIT = TH * TW + 16 / ES;              // g^T: elements per channel (padded, 16-B multiple)
    constexpr int XROW = 24 * 2 / ES >= 24 ? 24 : 24;     // x^T: elements per halo row (>= HW + 2, 16-B multiple)
    constexpr int XPIT = HH * XROW + 16 / ES;            // x^T: elements per channel
    constexpr int LG_BYTES = 32 * GPIT * ES, LX_BYTES = 32 * XPIT * ES;
    constexpr int FL_BYTES = (4096 + 1024 * NTAP) * 4;      // flush staging (wgrad_flush)
    constexpr int SM_BYTES = (LG_BYTES + LX_BYTES) > FL_BYTES ? (LG_BYTES + LX_BYTES) : FL_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SM_BYTES];
    T* lg = (T*)smem;
    T* lx = (T*)(smem + LG_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_it = (Ci + 31) / 32;
    const int o0 = (blockIdx.x / n_it) * 32, i0 = (blockIdx.x % n_it) * 32;
    f32x16_t acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    const int ntiles = tiles_x * tiles_y * B;
    constexpr int CPR = 32 / EP;                        // 16-byte channel chunks per 32-channel slab
    constexpr int NG = (TH * TW * CPR + 255) / 256, NX = (HH * HW * CPR + 255) / 256;
    __shared__ float laff[64];                          // instance-norm affine of the current sample: sc[32] | sh[32]
    const int m = lane & 31, kg = lane >> 5;
    uint4 greg[NG], xreg[NX];                           // next tile, in flight while the current one is in the MFMAs
    auto load_tile = [&](int tile) {
        const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx_i * TW, y0 = ty_i * TH;
#pragma unroll
        for (int k = 0; k < NG; k++) {
            const int idx = tid + k * 256;
            const int pix = idx % (TH * TW), c = idx / (TH * TW), gy = y0 + pix / TW, gx = x0 + pix % TW;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < TH * TW * CPR && gy < H && gx < W && o0 + c * EP < Co)
                v = *(const uint4*)(g + ((size_t)(b * H + gy) * W + gx) * Co + o0 + c * EP);
            greg[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NX; k++) {
            const int idx = tid + k * 256;
            const int pix = idx % (HH * HW), c = idx / (HH * HW), hy = pix / HW, hx = pix % HW;
            const int gy = y0 + hy - HALO, gx = x0 + hx - HALO;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < HH * HW * CPR && gy >= 0 && gy < H && gx >= 0 && gx < W && i0 + c * EP < Ci)
                v = *(const uint4*)(X + ((size_t)(b * H + gy) * W + gx) * Ci + i0 + c * EP);
            xreg[k] = v;
        }
    };
    int b_aff = -1;
    if ((int)blockIdx.y < ntiles) load_tile(blockIdx.y);
    for (int tile = blockIdx.y; tile < ntiles; tile += ntile_groups) {
        const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx_i * TW, y0 = ty_i * TH;
        __syncthreads();                                 // previous tile's fragments are consumed
        if (sc && b != b_aff) {                          // (block-uniform) refresh the affine table
            if (tid < 64) {
                const int ch = i0 + (tid & 31);
                laff[tid] = ch < Ci ? (tid < 32 ? sc[b * Ci + ch] : sh[b * Ci + ch]) : 0.f;
            }
            b_aff = b;
            __syncthreads();
        }
        // ---- transposing scatter: consecutive lanes hold consecutive pixels of one channel chunk
#pragma unroll
        for (int k = 0; k < NG; k++) {
            const int idx = tid + k * 256;
            if (idx < TH * TW * CPR) {
                const int pix = idx % (TH * TW), c = idx / (TH * TW);
                const T* e = (const T*)&greg[k];
#pragma unroll
                for (int q = 0; q < EP; q++) lg[(c * EP + q) * GPIT + pix] = e[q];
            }
        }
#pragma unroll
        for (int k = 0; k < NX; k++) {
            const int idx = tid + k * 256;
            if (idx < HH * HW * CPR) {
                const int pix = idx % (HH * HW), c = idx / (HH * HW), hy = pix / HW, hx = pix % HW;
                const int gy = y0 + hy - HALO, gx = x0 + hx - HALO;
                uint4 v = xreg[k];
                if (sc && gy >= 0 && gy < H && gx >= 0 && gx < W) {      // padding stays zero
                    float f[EP];
                    unpack16(v, f, (T*)nullptr);
#pragma unroll
                    for (int q = 0; q < EP; q++) f[q] = f[q] * laff[c * EP + q] + laff[32 + c * EP + q];
                    v = pack16(f, (T*)nullptr);
                }
                const T* e = (const T*)&v;
#pragma unroll
                for (int q = 0; q < EP; q++) lx[(c * EP + q) * XPIT + hy * XROW + hx] = e[q];
            }
        }
        __syncthreads();
        if (tile + ntile_groups < ntiles) load_tile(tile + ntile_groups);   // flies under the MFMAs
        // ---- MFMA: wave w owns tile rows {RW*w .. RW*w + RW-1}
        constexpr int RW = TH / 4;
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            const int row = RW * wave + rr;
            if constexpr (sizeof(T) == 2) {
                const uint4 a = *(const uint4*)(lg + m * GPIT + row * TW + kg * 8);
#pragma unroll
                for (int dy = 0; dy < KS; dy++) {
                    const T* xr = lx + m * XPIT + (row + dy) * XROW + kg * 8;
                    const uint4 v0 = *(const uint4*)xr;
                    if constexpr (KS == 3) {
                        const uint32_t v1 = *(const uint32_t*)(xr + 8);
                        const uint4 b1 = make_uint4(__builtin_amdgcn_alignbit(v0.y, v0.x, 16), __builtin_amdgcn_alignbit(v0.z, v0.y, 16),
                                                    __builtin_amdgcn_alignbit(v0.w, v0.z, 16), __builtin_amdgcn_alignbit(v1, v0.w, 16));
                        const uint4 b2 = make_uint4(v0.y, v0.z, v0.w, v1);
                        acc[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&a, *(const bf16x8_t*)&v0, acc[dy * 3 + 0], 0, 0, 0);
                        acc[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&a, *(const bf16x8_t*)&b1, acc[dy * 3 + 1], 0, 0, 0);
                        acc[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&a, *(const bf16x8_t*)&b2, acc[dy * 3 + 2], 0, 0, 0);
                    } else {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&a, *(const bf16x8_t*)&v0, acc[0], 0, 0, 0);
                    }
                }
            } else {
                // f32: MFMA step j contracts pixels (j, j+8) of the row: lane half kg reads 8 consecutive floats
                const float* ar = (const float*)lg + m * GPIT + row * TW + kg * 8;
                float af[8];
                *(uint4*)&af[0] = *(const uint4*)ar; *(uint4*)&af[4] = *(const uint4*)(ar + 4);
#pragma unroll
                for (int dy = 0; dy < KS; dy++) {
                    const float* xr = (const float*)lx + m * XPIT + (row + dy) * XROW + kg * 8;
                    float xf[12];
                    *(uint4*)&xf[0] = *(const uint4*)xr; *(uint4*)&xf[4] = *(const uint4*)(xr + 4);
                    if constexpr (KS == 3) *(uint4*)&xf[8] = *(const uint4*)(xr + 8);
#pragma unroll
                    for (int dx = 0; dx < KS; dx++)
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            acc[dy * KS + dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], xf[j + dx], acc[dy * KS + dx], 0, 0, 0);
                }
            }
        }
    }
    wgrad_flush<NTAP>(acc, (float*)smem, dW, o0, i0, Co, Ci, tid, lane, wave);
}

// bf16 variant built on the gfx950 transposing LDS read.  Both tiles sit in LDS in their natural NHWC
// order (one 64-byte row of 32 channels per pixel, written with plain 16-byte stores) and
// ds_read_b64_tr_b16 delivers the K(=pixel)-contiguous MFMA fragments: in each 16-lane group lane i
// points at pixel (i>>2), channels 4*(i&3)..+3 of a [4 pixel][16 channel] block and receives channel
// (i) of the 4 pixels.  Two reads (pixels +0..3, +4..7) make the 8-deep fragment of one lane half.
// The x window of a kernel row is read once as 12 pixels; the dx = 1, 2 fragments are register
// shifts of it (v_alignbit / renaming), so 3 + 2 LDS reads feed 3 MFMAs.
typedef short v4s_t __attribute__((ext_vector_type(4)));
template <int KS, int TH>
__global__ __launch_bounds__(256) void conv_wgrad_tr_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ X,
                                                             const float* __restrict__ sc, const float* __restrict__ sh,
                                                             float* __restrict__ dW, int B, int H, int W, int Co, int Ci,
                                                             int tiles_x, int tiles_y, int ntile_groups) {
    // TH x 16 pixel tiles.  TH = 16 wherever the image has 16 rows: a wave then issues 36 instead of 18 MFMAs between
    // the two barriers of a tile, which is what the per-tile overhead (LDS fill, barriers, fragment latency) is paid against.
    constexpr int TW = 16, HALO = KS / 2, HH = TH + 2 * HALO, HW = TW + 2 * HALO;
    constexpr int NTAP = KS * KS;
    constexpr int LG_BYTES = TH * TW * 64, LX_BYTES = (HH * HW + 4) * 64;     // +4 pixels: the 12-pixel window of the last row over-reads
    constexpr int FL_BYTES = (4096 + 1024 * NTAP) * 4;      // flush staging (wgrad_flush)
    constexpr int SM_BYTES = (LG_BYTES + LX_BYTES) > FL_BYTES ? (LG_BYTES + LX_BYTES) : FL_BYTES;
    __shared__ __attribute__((aligned(256))) unsigned char smem[SM_BYTES];
    __shared__ float laff[64];                          // instance-norm affine of the current sample: sc[32] | sh[32]
    unsigned char* lg = smem;
    unsigned char* lx = smem + LG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_it = (Ci + 31) / 32;
    const int o0 = (blockIdx.x / n_it) * 32, i0 = (blockIdx.x % n_it) * 32;
    f32x16_t acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    const int ntiles = tiles_x * tiles_y * B;
    constexpr int NG = TH * TW * 4 / 256, NX = (HH * HW * 4 + 255) / 256;
    const int cch = tid & 3;                             // 16-byte channel chunk of every item of this thread
    const bool gc_ok = o0 + cch * 8 < Co, xc_ok = i0 + cch * 8 < Ci;
    uint4 greg[NG], xreg[NX];                            // next tile, in flight while the current one is in the MFMAs
    unsigned xin = 0;                                    // bit k: x item k lies inside the image
    auto load_tile = [&](int tile) {
        const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx_i * TW, y0 = ty_i * TH;
        const bf16_t* gb = g + (size_t)b * H * W * Co + o0 + cch * 8;
        const bf16_t* xb = X + (size_t)b * H * W * Ci + i0 + cch * 8;
#pragma unroll
        for (int k = 0; k < NG; k++) {
            const int pix = (tid >> 2) + k * 64, gy = y0 + pix / TW, gx = x0 + pix % TW;
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((gy < H) & (gx < W) & gc_ok) v = *(const uint4*)(gb + (gy * W + gx) * Co);
            greg[k] = v;
        }
        xin = 0;
#pragma unroll
        for (int k = 0; k < NX; k++) {
            const int pix = (tid >> 2) + k * 64, hy = pix / HW, hx = pix % HW;
            const int gy = y0 + hy - HALO, gx = x0 + hx - HALO;
            const bool in = (pix < HH * HW) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W) & xc_ok;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (in) v = *(const uint4*)(xb + (gy * W + gx) * Ci);
            xin |= (in ? 1u : 0u) << k;
            xreg[k] = v;
        }
    };
    // per-lane fragment addressing of the transposing reads
    const int grp = lane >> 4, li = lane & 15;
    const int frag_off = ((grp >> 1) * 8 + (li >> 2)) * 64 + (grp & 1) * 32 + (li & 3) * 8;
    typedef __attribute__((address_space(3))) v4s_t* lds_v4s_ptr;
    auto trd = [&](const unsigned char* base, int byte_off) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(base + byte_off));
    };
    int b_aff = -1;
    if ((int)blockIdx.y < ntiles) load_tile(blockIdx.y);
    for (int tile = blockIdx.y; tile < ntiles; tile += ntile_groups) {
        const int b = tile / (tiles_x * tiles_y);
        __syncthreads();                                 // previous tile's fragments are consumed
        if (sc && b != b_aff) {                          // (block-uniform) refresh the affine table
            if (tid < 64) {
                const int ch = i0 + (tid & 31);
                laff[tid] = ch < Ci ? (tid < 32 ? sc[b * Ci + ch] : sh[b * Ci + ch]) : 0.f;
            }
            b_aff = b;
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < NG; k++) *(uint4*)(lg + (tid + k * 256) * 16) = greg[k];
#pragma unroll
        for (int k = 0; k < NX; k++) {
            if (NX * 256 == HH * HW * 4 || tid + k * 256 < HH * HW * 4) {
                uint4 v = xreg[k];
                if (sc && ((xin >> k) & 1u)) {           // padding stays zero
                    float f[8];
                    unpack16(v, f, (bf16_t*)nullptr);
#pragma unroll
                    for (int q = 0; q < 8; q++) f[q] = f[q] * laff[cch * 8 + q] + laff[32 + cch * 8 + q];
                    v = pack16(f, (bf16_t*)nullptr);
                }
                *(uint4*)(lx + (tid + k * 256) * 16) = v;
            }
        }
        __syncthreads();
        if (tile + ntile_groups < ntiles) load_tile(tile + ntile_groups);   // flies under the MFMAs
        // ---- MFMA: wave w owns tile rows {RW*w .. RW*w + RW-1}
        constexpr int RW = TH / 4;
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            const int row = RW * wave + rr;
            const unsigned char* ga = lg + row * TW * 64 + frag_off;
            const v4s_t a0 = trd(ga, 0), a1 = trd(ga, 4 * 64);
            const uint2 a01 = *(const uint2*)&a0, a23 = *(const uint2*)&a1;
            const uint4 av = make_uint4(a01.x, a01.y, a23.x, a23.y);
            const bf16x8_t a = *(const bf16x8_t*)&av;
#pragma unroll
            for (int dy = 0; dy < KS; dy++) {
                const unsigned char* xa = lx + (row + dy) * HW * 64 + frag_off;
                const v4s_t q0 = trd(xa, 0), q1 = trd(xa, 4 * 64);
                const uint2 w01 = *(const uint2*)&q0, w23 = *(const uint2*)&q1;
                const uint4 v0 = make_uint4(w01.x, w01.y, w23.x, w23.y);
                if constexpr (KS == 3) {
                    const v4s_t q2 = trd(xa, 8 * 64);
                    const uint32_t w4 = (*(const uint2*)&q2).x;
                    const uint4 b1 = make_uint4(__builtin_amdgcn_alignbit(v0.y, v0.x, 16), __builtin_amdgcn_alignbit(v0.z, v0.y, 16),
                                                __builtin_amdgcn_alignbit(v0.w, v0.z, 16), __builtin_amdgcn_alignbit(w4, v0.w, 16));
                    const uint4 b2 = make_uint4(v0.y, v0.z, v0.w, w4);
                    acc[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, *(const bf16x8_t*)&v0, acc[dy * 3 + 0], 0, 0, 0);
                    acc[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, *(const bf16x8_t*)&b1, acc[dy * 3 + 1], 0, 0, 0);
                    acc[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, *(const bf16x8_t*)&b2, acc[dy * 3 + 2], 0, 0, 0);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, *(const bf16x8_t*)&v0, acc[0], 0, 0, 0);
                }
            }
        }
    }
    wgrad_flush<NTAP>(acc, (float*)smem, dW, o0, i0, Co, Ci, tid, lane, wave);
}

// ------------------------------------------------------------------ activation backward (+pool adjoint)
// a = lrelu(pre) saved.  g_pre[b,p,c] = scale * g_up[b,q(p),c] * lrelu'(a)   (q = p/2 per axis when pool)
// red[c,:] (pre-zeroed, summed over the whole batch) += { sum g_pre, sum g_pre*noise[b,p] }
template <typename T, int NS>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ gup, const T* __restrict__ a,
                                                       const float* __restrict__ noise, T* __restrict__ gpre,
                                                       float* __restrict__ red_out, int H, int W, int C, int pool, float scale, float slope) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * NS * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    const int HW = H * W, UW = pool ? W / 2 : W, UHW = pool ? HW / 4 : HW;
    const float raw_w = pool ? 0.25f : 1.f;              // every gup element is visited by its 4 children
    float s[NS][EP];
#pragma unroll
    for (int k = 0; k < NS; k++)
#pragma unroll
        for (int e = 0; e < EP; e++) s[k][e] = 0.f;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            const int q = pool ? (p / W / 2) * UW + (p % W) / 2 : p;
            float g[EP], av[EP];
            unpack16(*(const uint4*)(gup + ((size_t)b * UHW + q) * C + chunk * EP), g, (T*)nullptr);
            unpack16(*(const uint4*)(a + ((size_t)b * HW + p) * C + chunk * EP), av, (T*)nullptr);
            const float nz = noise ? noise[(size_t)b * HW + p] : 0.f;
#pragma unroll
            for (int e = 0; e < EP; e++) {
                if constexpr (NS == 3) s[2][e] += raw_w * g[e];
                g[e] = scale * g[e] * (av[e] > 0.f ? 1.f : slope);
                s[0][e] += g[e]; s[1][e] += g[e] * nz;
            }
            *(uint4*)(gpre + ((size_t)b * HW + p) * C + chunk * EP) = pack16(g, (T*)nullptr);
        }
    }
    if (red_out) block_chan_flush<EP, NS>(s, cpt, ppi, red_out + (size_t)b * C * NS, C, red);      // per-sample partial sums
}

// The pooled case from the sign mask of dge_blend_pool_mask: one thread = one 16-byte chunk of one POOLED pixel q = its four children.
// g_pre[b, child, c] = scale * g_up[b,q,c] * (bit ? 1 : slope);  red as act_bwd_kernel (third column: sum of g_up).
template <typename T, int NS>
__global__ __launch_bounds__(256) void act_bwd_mask_kernel(const T* __restrict__ gup, const unsigned* __restrict__ mask,
                                                            const float* __restrict__ noise, T* __restrict__ gpre,
                                                            float* __restrict__ red_out, int H, int W, int C, float scale, float slope) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * NS * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    const int UW = W / 2, UHW = (H / 2) * UW, HW = H * W;
    float s[NS][EP];
#pragma unroll
    for (int k = 0; k < NS; k++)
#pragma unroll
        for (int e = 0; e < EP; e++) s[k][e] = 0.f;
    // two pooled pixels per thread and iteration, every load of both issued before the arithmetic (more bytes in flight per wave)
    const int stride = gridDim.x * ppi;
    for (int q0 = blockIdx.x * ppi; q0 < UHW; q0 += 2 * stride) {
        const int qq[2] = {q0 + slot, q0 + slot + stride};
        bool ok[2]; uint4 gv[2]; unsigned mv[2]; int p00[2]; float nzv[2][4];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            ok[h] = slot < ppi && qq[h] < UHW;
            gv[h] = make_uint4(0, 0, 0, 0); mv[h] = 0; p00[h] = 0;
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++) nzv[h][c4] = 0.f;
            if (ok[h]) {
                gv[h] = *(const uint4*)(gup + ((size_t)b * UHW + qq[h]) * C + chunk * EP);
                mv[h] = mask[((size_t)b * UHW + qq[h]) * cpt + chunk];
                const int oy = qq[h] / UW, ox = qq[h] - oy * UW;
                p00[h] = (2 * oy) * W + 2 * ox;
                if (noise) {
#pragma unroll
                    for (int c4 = 0; c4 < 4; c4++) nzv[h][c4] = noise[(size_t)b * HW + p00[h] + (c4 >> 1) * W + (c4 & 1)];
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (!ok[h]) continue;
            float g[EP];
            unpack16(gv[h], g, (T*)nullptr);
            const unsigned m = mv[h];
#pragma unroll
            for (int e = 0; e < EP; e++) { if constexpr (NS == 3) s[2][e] += g[e]; g[e] *= scale; }
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++) {
                const int p = p00[h] + (c4 >> 1) * W + (c4 & 1);
                const float nz = nzv[h][c4];
                float o[EP];
#pragma unroll
                for (int e = 0; e < EP; e++) {
                    o[e] = g[e] * (((m >> (c4 * EP + e)) & 1u) ? 1.f : slope);
                    s[0][e] += o[e]; s[1][e] += o[e] * nz;
                }
                *(uint4*)(gpre + ((size_t)b * HW + p) * C + chunk * EP) = pack16(o, (T*)nullptr);
            }
        }
    }
    if (red_out) block_chan_flush<EP, NS>(s, cpt, ppi, red_out + (size_t)b * C * NS, C, red);
}

// ------------------------------------------------------------------ instance-norm + statistics backward
// Coefficients of g_X = A*g_y + Bc*X + Cc for y = (X-mu)*r, with the extra gradients g_mu, g_sigma of
// the (mean, std) outputs (E.py:51-53): see DESIGN.md.  dots[b,c] = (sum g_y*X, sum g_y) from the dgrad conv
// epilogue (may be null when y has no consumer), gms [B,2C] = [g_mu | g_sigma].
__global__ void in_bwd_coef_kernel(const float* __restrict__ dots, const float* __restrict__ gms, const float* __restrict__ musig,
                                   const float* __restrict__ sc, const float* __restrict__ sh, float* __restrict__ coef,
                                   int B, int C, float inv_n, int nslot) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx % C;
    const float r = sc[idx], s = sh[idx];
    float S2 = 0.f, S1 = 0.f;                 // nslot copies of the two sums (dge_conv2d statistics slots) are added here
    if (dots) {
        const float2 ss = sum_slot_pairs(dots + (size_t)idx * 2, (size_t)B * C * 2, nslot);
        S2 = ss.x; S1 = ss.y;
    }
    const float m1 = S1 * inv_n, m2 = (r * S2 + s * S1) * inv_n;
    const float mu = musig[(size_t)b * 2 * C + c], sg = musig[(size_t)b * 2 * C + C + c];
    const float gmu = gms ? gms[(size_t)b * 2 * C + c] : 0.f, gsg = gms ? gms[(size_t)b * 2 * C + C + c] : 0.f;
    const float k = sg > 0.f ? gsg * inv_n / sg : 0.f;
    coef[(size_t)idx * 3 + 0] = r;
    coef[(size_t)idx * 3 + 1] = -r * r * m2 + k;
    coef[(size_t)idx * 3 + 2] = -r * m1 - r * m2 * s + gmu * inv_n - k * mu;
}

// g_X = A*g_y + Bc*X + Cc + extra_scale * extra[q(p)]   then, when act != 0 (X = lrelu(pre)):
// g_pre = g_X * lrelu'(X) with the bias / noise-weight reductions as in act_bwd_kernel.
// Sources of the coefficients when in_bwd computes them itself (dge_in_bwd_fused: no in_bwd_coef launch in front of it)
struct InCoefSrc { const float* dots; const float* gms; const float* musig; const float* sc; const float* sh; int nslot, B; float inv_n; };
// FR (the last launch of the encoder backward: X is the FromRGB output x0 = lrelu(W img + b), net.py:231-240): the result g_x0 is
// not stored - its only reader was dge_fromrgb_bwd, which read x0 a second time - but reduced in place: g_pre = g_x0*lrelu'(x0),
// fr_out[b][c][0..2] += sum g_pre*img[k], fr_out[b][c][3] += sum g_pre.
template <typename T, bool FR>
__global__ __launch_bounds__(256) void in_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ X, const float* __restrict__ coef,
                                                      const T* __restrict__ extra, const float* __restrict__ noise,
                                                      T* __restrict__ gout, float* __restrict__ red_out, int H, int W, int C,
                                                      int extra_pool, float extra_scale, int act, InCoefSrc cs,
                                                      const float* __restrict__ img, float* __restrict__ fr_out) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * (FR ? 4 : 2) * EP];
    __shared__ float lcoef[3 * 512];                    // fused form (C <= 512): this sample's coefficients, computed by the workgroup
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    const int HW = H * W, UW = extra_pool ? W / 2 : W, UHW = extra_pool ? HW / 4 : HW;
    float s[FR ? 4 : 2][EP], A[EP], Bc[EP], Cc[EP];
    if (!coef) {
        // the math of in_bwd_coef_kernel for the C channels of sample b (every workgroup of the sample repeats it: C <= 512 channels,
        // one thread each, slot copies summed with 8 loads in flight - cheaper than a launch of its own in front of this one)
        for (int c = threadIdx.x; c < C; c += 256) {
            const int idx = b * C + c;
            const float r = cs.sc[idx], sft = cs.sh[idx];
            float S2 = 0.f, S1 = 0.f;
            if (cs.dots) { const float2 ss = sum_slot_pairs(cs.dots + (size_t)idx * 2, (size_t)cs.B * C * 2, cs.nslot); S2 = ss.x; S1 = ss.y; }
            const float m1 = S1 * cs.inv_n, m2 = (r * S2 + sft * S1) * cs.inv_n;
            const float mu = cs.musig[(size_t)b * 2 * C + c], sg = cs.musig[(size_t)b * 2 * C + C + c];
            const float gmu = cs.gms ? cs.gms[(size_t)b * 2 * C + c] : 0.f, gsg = cs.gms ? cs.gms[(size_t)b * 2 * C + C + c] : 0.f;
            const float k = sg > 0.f ? gsg * cs.inv_n / sg : 0.f;
            lcoef[c * 3 + 0] = r;
            lcoef[c * 3 + 1] = -r * r * m2 + k;
            lcoef[c * 3 + 2] = -r * m1 - r * m2 * sft + gmu * cs.inv_n - k * mu;
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < EP; e++) {
        s[0][e] = s[1][e] = 0.f;
        if constexpr (FR) s[2][e] = s[3][e] = 0.f;
        if (coef) {
            const size_t ci = ((size_t)b * C + chunk * EP + e) * 3;
            A[e] = coef[ci]; Bc[e] = coef[ci + 1]; Cc[e] = coef[ci + 2];
        } else {
            const int ci = (chunk * EP + e) * 3;
            A[e] = lcoef[ci]; Bc[e] = lcoef[ci + 1]; Cc[e] = lcoef[ci + 2];
        }
    }
    // two pixels per thread and iteration, all loads issued before the arithmetic (more bytes in flight per wave)
    const int stride = gridDim.x * ppi;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += 2 * stride) {
        const int pA = p0 + slot, pB = pA + stride;
        const bool okA = slot < ppi && pA < HW, okB = slot < ppi && pB < HW;
        uint4 gA = make_uint4(0, 0, 0, 0), gB = gA, xA = gA, xB = gA, eA = gA, eB = gA;
        const size_t oA = ((size_t)b * HW + pA) * C + chunk * EP, oB = ((size_t)b * HW + pB) * C + chunk * EP;
        if (okA) { if (gy) gA = *(const uint4*)(gy + oA); xA = *(const uint4*)(X + oA); }
        if (okB) { if (gy) gB = *(const uint4*)(gy + oB); xB = *(const uint4*)(X + oB); }
        if (extra) {
            if (okA) { const int q = extra_pool ? (pA / W / 2) * UW + (pA % W) / 2 : pA; eA = *(const uint4*)(extra + ((size_t)b * UHW + q) * C + chunk * EP); }
            if (okB) { const int q = extra_pool ? (pB / W / 2) * UW + (pB % W) / 2 : pB; eB = *(const uint4*)(extra + ((size_t)b * UHW + q) * C + chunk * EP); }
        }
        float nzA = 0.f, nzB = 0.f;
        if (act && noise) { if (okA) nzA = noise[(size_t)b * HW + pA]; if (okB) nzB = noise[(size_t)b * HW + pB]; }
        float imA[3] = {0.f, 0.f, 0.f}, imB[3] = {0.f, 0.f, 0.f};
        if constexpr (FR) {
            const float* ib = img + (size_t)b * 3 * HW;
#pragma unroll
            for (int k = 0; k < 3; k++) { if (okA) imA[k] = ib[(size_t)k * HW + pA]; if (okB) imB[k] = ib[(size_t)k * HW + pB]; }
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
            if (!(h2 ? okB : okA)) continue;
            float g[EP], xv[EP];
            unpack16(h2 ? gB : gA, g, (T*)nullptr);
            unpack16(h2 ? xB : xA, xv, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < EP; e++) g[e] = A[e] * g[e] + Bc[e] * xv[e] + Cc[e];
            if (extra) {
                float ex[EP];
                unpack16(h2 ? eB : eA, ex, (T*)nullptr);
#pragma unroll
                for (int e = 0; e < EP; e++) g[e] += extra_scale * ex[e];
            }
            if (act) {
                const float nz = h2 ? nzB : nzA;
#pragma unroll
                for (int e = 0; e < EP; e++) {
                    g[e] *= (xv[e] > 0.f ? 1.f : 0.2f);
                    s[0][e] += g[e]; s[1][e] += g[e] * nz;
                }
            }
            if constexpr (FR) {
                const float* im = h2 ? imB : imA;
#pragma unroll
                for (int e = 0; e < EP; e++) {
                    const float gp = g[e] * (xv[e] > 0.f ? 1.f : 0.2f);
                    s[0][e] += gp * im[0]; s[1][e] += gp * im[1]; s[2][e] += gp * im[2]; s[3][e] += gp;
                }
            } else {
                *(uint4*)(gout + (h2 ? oB : oA)) = pack16(g, (T*)nullptr);
            }
        }
    }
    if constexpr (FR) block_chan_flush<EP, 4>(s, cpt, ppi, fr_out + (size_t)b * C * 4, C, red);
    else if (red_out) block_chan_flush<EP, 2>(s, cpt, ppi, red_out + (size_t)b * C * 2, C, red);       // per-sample partial sums
}

// per-channel sum over batch and pixels of an NHWC tensor: out[c] += scale * sum x[b,p,c]
template <typename T>
__global__ __launch_bounds__(256) void chan_sum_kernel(const T* __restrict__ x, float* __restrict__ out, int HW, int C, float scale) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    float s[1][EP];
#pragma unroll
    for (int e = 0; e < EP; e++) s[0][e] = 0.f;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            float f[EP];
            unpack16(*(const uint4*)(x + ((size_t)b * HW + p) * C + chunk * EP), f, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < EP; e++) s[0][e] += scale * f[e];
        }
    }
    block_chan_flush<EP, 1>(s, cpt, ppi, out + (size_t)blockIdx.y * C, C, red);
}

// FromRGB backward: x0 = lrelu(W img + b).  g_pre = g_x0*lrelu'(x0);  out[o][0..2] += sum g_pre*img[c], out[o][3] += sum g_pre
template <typename T>
__global__ __launch_bounds__(256) void fromrgb_bwd_kernel(const T* __restrict__ gx, const T* __restrict__ x0,
                                                           const float* __restrict__ img, float* __restrict__ out, int HW, int C) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 4 * EP];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    float s[4][EP];
#pragma unroll
    for (int e = 0; e < EP; e++) s[0][e] = s[1][e] = s[2][e] = s[3][e] = 0.f;
    const float* ib = img + (size_t)b * 3 * HW;
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {
        const int p = p0 + slot;
        if (slot < ppi && p < HW) {
            const size_t o = ((size_t)b * HW + p) * C + chunk * EP;
            float g[EP], xv[EP];
            unpack16(*(const uint4*)(gx + o), g, (T*)nullptr);
            unpack16(*(const uint4*)(x0 + o), xv, (T*)nullptr);
            const float r = ib[p], gr = ib[HW + p], bl = ib[2 * HW + p];
#pragma unroll
            for (int e = 0; e < EP; e++) {
                const float gp = g[e] * (xv[e] > 0.f ? 1.f : 0.2f);
                s[0][e] += gp * r; s[1][e] += gp * gr; s[2][e] += gp * bl; s[3][e] += gp;
            }
        }
    }
    block_chan_flush<EP, 4>(s, cpt, ppi, out + (size_t)b * C * 4, C, red);
}

// FromRGB data gradient (needed when the encoder input itself carries a gradient: embedding_img.py:88 E(imgs2)):
// gimg[b,k,p] = sum_c W[c][k] * g_x0[b,p,c] * lrelu'(x0[b,p,c]),  k = 0..2.  One wave per 64 pixels x channel chunks:
// thread (slot, chunk) reduces its chunk, the chunks of a pixel are combined in LDS.
template <typename T>
__global__ __launch_bounds__(256) void fromrgb_dgrad_kernel(const T* __restrict__ gx, const T* __restrict__ x0, const float* __restrict__ w,
                                                             float* __restrict__ gimg, int HW, int C) {
    constexpr int EP = Elem<T>::PER16;
    __shared__ float red[256 * 3];
    const int b = blockIdx.y;
    const int cpt = C / EP, ppi = 256 / cpt;
    const int chunk = threadIdx.x % cpt, slot = threadIdx.x / cpt;
    float wr[3][EP];
#pragma unroll
    for (int e = 0; e < EP; e++)
#pragma unroll
        for (int k = 0; k < 3; k++) wr[k][e] = w[(size_t)(chunk * EP + e) * 3 + k];
    for (int p0 = blockIdx.x * ppi; p0 < HW; p0 += gridDim.x * ppi) {      // block-uniform trip count
        const int p = p0 + slot;
        float a[3] = {0.f, 0.f, 0.f};
        if (slot < ppi && p < HW) {
            const size_t o = ((size_t)b * HW + p) * C + chunk * EP;
            float g[EP], xv[EP];
            unpack16(*(const uint4*)(gx + o), g, (T*)nullptr);
            unpack16(*(const uint4*)(x0 + o), xv, (T*)nullptr);
#pragma unroll
            for (int e = 0; e < EP; e++) {
                const float gp = g[e] * (xv[e] > 0.f ? 1.f : 0.2f);
                a[0] += gp * wr[0][e]; a[1] += gp * wr[1][e]; a[2] += gp * wr[2][e];
            }
        }
        red[threadIdx.x * 3 + 0] = a[0]; red[threadIdx.x * 3 + 1] = a[1]; red[threadIdx.x * 3 + 2] = a[2];
        __syncthreads();
        if (chunk == 0 && slot < ppi && p < HW) {
            float t[3] = {0.f, 0.f, 0.f};
            for (int c = 0; c < cpt; c++)
#pragma unroll
                for (int k = 0; k < 3; k++) t[k] += red[(slot * cpt + c) * 3 + k];
#pragma unroll
            for (int k = 0; k < 3; k++) gimg[((size_t)b * 3 + k) * HW + p] = t[k];
        }
        __syncthreads();
    }
}

// upscale2d (nearest x2) materialised: y[b,2y+dy,2x+dx,c] = scale * x[b,y,x,c]
template <typename T>
__global__ __launch_bounds__(256) void nearest_up2_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int C, float scale, long total16) {
    constexpr int EP = Elem<T>::PER16;
    const int cpt = C / EP;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total16; idx += (long)gridDim.x * 256) {
        const int chunk = idx % cpt;
        const long pix = idx / cpt;                       // output pixel index over [B,2H,2W]
        const int ox = pix % (2 * W), oy = (pix / (2 * W)) % (2 * H);
        const long b = pix / ((long)4 * H * W);
        float v[EP];
        unpack16(*(const uint4*)(x + ((b * H + (oy >> 1)) * W + (ox >> 1)) * C + chunk * EP), v, (T*)nullptr);
#pragma unroll
        for (int e = 0; e < EP; e++) v[e] *= scale;
        *(uint4*)(y + pix * C + chunk * EP) = pack16(v, (T*)nullptr);
    }
}

// dense layer parameter gradients: gW[o][i] (+)= sum_b gy[b][o]*x[b][i];  gb[o] (+)= sum_b gy[b][o]
__global__ void dense_wgrad_kernel(const float* __restrict__ gy, int ldgy, const float* __restrict__ x, int ldx,
                                   float* __restrict__ gW, float* __restrict__ gb, int B, int O, int I, int accumulate) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    if (idx >= (long)O * I) return;
    const int o = idx / I, i = idx % I;
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; b++) { const float g = gy[(size_t)b * ldgy + o]; s += g * x[(size_t)b * ldx + i]; sb += g; }
    gW[idx] = accumulate ? gW[idx] + s : s;
    if (gb && i == 0) gb[o] = accumulate ? gb[o] + sb : sb;
}

// ------------------------------------------------------------------ all inver_mod heads of the encoder backward at once
// The gradient of every head w_l = musig_l @ W_l^T + b_l is known when the backward starts (g_w [B, NL, O] comes from the loss),
// so the 2 x NL per-layer launches (transposed dense layer + parameter gradients) collapse into two:
//   gms_l[b,k]  = sum_o g_w[b, col_l + o] * W_l[o,k]                               (head_bwd_data_kernel)
//   gW_l[o,i]   = sum_b g_w[b, col_l + o] * musig_l[b,i],  gb_l[o] = sum_b g_w[..]  (head_bwd_param_kernel)
// musig_l / gms_l live at moff_l in flat [sum_l B*I_l] buffers, gW_l at woff_l, gb_l at boff_l of flat buffers.
struct HeadEntry { const float* W; long moff, woff; int I, gcol, boff, pad; const float* bias; };
__global__ __launch_bounds__(1024) void head_bwd_data_kernel(const HeadEntry* __restrict__ tab, const float* __restrict__ g, int ldg,
                                                             float* __restrict__ gms_all, int O) {
    __shared__ float part[16][64];
    const HeadEntry e = tab[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane, b = blockIdx.y;
    if (blockIdx.x * 64 >= e.I) return;                    // block-uniform
    const int per = (O + 15) / 16, o0 = wave * per, o1 = min(O, o0 + per);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    if (k < e.I) {
        const float* gr = g + (size_t)b * ldg + e.gcol;
        int o = o0;
        for (; o + 7 < o1; o += 8) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] += gr[o + j] * e.W[(size_t)(o + j) * e.I + k];
        }
        for (; o < o1; o++) acc[0] += gr[o] * e.W[(size_t)o * e.I + k];
    }
    part[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (wave == 0 && k < e.I) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) s += part[j][lane];
        gms_all[e.moff + (size_t)b * e.I + k] = s;
    }
}
__global__ void head_bwd_param_kernel(const HeadEntry* __restrict__ tab, const float* __restrict__ g, int ldg,
                                      const float* __restrict__ musig_all, float* __restrict__ gw_all, float* __restrict__ gb_all,
                                      int B, int O) {
    const HeadEntry e = tab[blockIdx.y];
    const int total = O * e.I;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int o = idx / e.I, i = idx - o * e.I;
        float s = 0.f, sb = 0.f;
        for (int b = 0; b < B; b++) {
            const float gv = g[(size_t)b * ldg + e.gcol + o];
            s += gv * musig_all[e.moff + (size_t)b * e.I + i]; sb += gv;
        }
        gw_all[e.woff + idx] = s;
        if (i == 0) gb_all[e.boff + o] = sb;
    }
}

// =================================================================== C ABI
#define CHAN_OK(C, ep) ((C) % (ep) == 0 && (C) / (ep) <= 256 && 256 % ((C) / (ep)) == 0)

int dge_wgrad_dma_try(const void* g, const void* x, const float* sc, const float* sh, float* dw, int B, int H, int W, int cout, int cin,
                      const float* wdot, float* dots, int dots_slots, hipStream_t s);       // wgrad_dma.hip: 0 done, 1 shape not covered, < 0 error
// dge_conv_wgrad + the two per-(sample, input channel) sums of the layer's data gradient (dge_conv_desc.dot_src statistics), taken
// from the weight-gradient correlations (wgrad_dma.hip).  Returns 1 (nothing launched) where the streaming kernel does not cover
// the shape: the caller then runs dge_conv_wgrad and lets the data gradient produce the sums.
extern "C" int dge_conv_wgrad_dots(const void* g, const void* x, const float* in_scale, const float* in_shift, float* dw, const float* w,
                                   float* dots, int dots_slots, int B, int H, int W, int cout, int cin, int dtype, hipStream_t s) {
    DGE_CHECK(g && x && dw && w && dots && in_scale && in_shift && dots_slots >= 1, "conv_wgrad_dots: null tensor");
    if (dtype != DGE_BF16) return 1;
    return dge_wgrad_dma_try(g, x, in_scale, in_shift, dw, B, H, W, cout, cin, w, dots, dots_slots, s);
}
extern "C" int dge_conv_wgrad(const void* g, const void* x, const float* in_scale, const float* in_shift, float* dw, int B, int H,
                              int W, int cout, int cin, int ksize, int dtype, hipStream_t s) {
    DGE_CHECK(ksize == 1 || ksize == 3, "conv_wgrad: ksize %d unsupported", ksize);
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(cout % ep == 0 && cin % ep == 0, "conv_wgrad: channels must be multiples of %d", ep);
    DGE_CHECK((in_scale == nullptr) == (in_shift == nullptr), "conv_wgrad: in_scale and in_shift go together");
    if (dtype == DGE_BF16 && ksize == 3) {       // the streaming kernel (LDS-DMA ring, affine on the accumulators)
        const int r = dge_wgrad_dma_try(g, x, in_scale, in_shift, dw, B, H, W, cout, cin, nullptr, nullptr, 1, s);
        if (r <= 0) return r;
    }
    const bool tall = dtype == DGE_BF16 && H >= 16 && !dge_env().wgrad_th8;
    const int tx = (W + 15) / 16, ty = tall ? (H + 15) / 16 : (H + 7) / 8;
    const int ntiles = tx * ty * B;
    const int noi = ((cout + 31) / 32) * ((cin + 31) / 32);
    int groups = 512 / noi; if (groups < 1) groups = 1;
    if (dge_env().wgrad_groups > 0) groups = dge_env().wgrad_groups;                          // tuning override (clamped below)
    if (groups > ntiles) groups = ntiles;   // few, long-running workgroups: one atomic flush each
    dim3 grid(noi, groups);
    if (dtype == DGE_BF16) dge_note_kernel("conv_wgrad_tr<%d,%d>", ksize, tall ? 16 : 8);
    else dge_note_kernel("conv_wgrad<f32,%d>", ksize);
#define WG(T, KS) hipLaunchKernelGGL((conv_wgrad_kernel<T, KS>), grid, dim3(256), 0, s, (const T*)g, (const T*)x, in_scale, in_shift, dw, B, H, W, cout, cin, tx, ty, groups)
    if (dtype == DGE_BF16) {
#define WT(KS, TH) hipLaunchKernelGGL((conv_wgrad_tr_kernel<KS, TH>), grid, dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)x, in_scale, in_shift, dw, B, H, W, cout, cin, tx, ty, groups)
        if (ksize == 3) { if (tall) WT(3, 16); else WT(3, 8); }
        else { if (tall) WT(1, 16); else WT(1, 8); }
#undef WT
    }
    else { if (ksize == 3) WG(float, 3); else WG(float, 1); }
#undef WG
    DGE_LAUNCH_CHECK("conv_wgrad");
    return 0;
}

// forward of all heads at once (E.py:51-53,64-66: w_l = musig_l @ W_l^T + b_l): one wave per (head, sample, output)
__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadEntry* __restrict__ tab, const float* __restrict__ musig_all,
                                                       float* __restrict__ w, int ldw, int B, int O) {
    const HeadEntry e = tab[blockIdx.y];
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B * O) return;
    const int b = wave / O, o = wave - b * O;
    const float* xr = musig_all + e.moff + (size_t)b * e.I;
    const float* wr = e.W + (size_t)o * e.I;
    float s = 0.f;
    for (int i = lane; i < e.I; i += 64) s = fmaf(xr[i], wr[i], s);
    s = wave_sum(s);
    if (lane == 0) w[(size_t)b * ldw + e.gcol + o] = s + (e.bias ? e.bias[o] : 0.f);
}
extern "C" int dge_heads_fwd(const void* dev_entries, int n, const float* musig_all, float* w, int ldw, int B, int O, hipStream_t s) {
    DGE_CHECK(n >= 1 && n <= 65535 && B >= 1 && O >= 1, "heads_fwd: bad sizes");
    hipLaunchKernelGGL(head_fwd_kernel, dim3((B * O * 64 + 255) / 256, n), dim3(256), 0, s, (const HeadEntry*)dev_entries, musig_all, w, ldw, B, O);
    DGE_LAUNCH_CHECK("heads_fwd");
    return 0;
}

extern "C" int dge_head_entry_size(void) { return (int)sizeof(HeadEntry); }
extern "C" int dge_heads_bwd(const void* dev_entries, int n, int max_I, const float* g, int ldg, const float* musig_all, float* gms_all,
                             float* gw_all, float* gb_all, int B, int O, hipStream_t s) {
    DGE_CHECK(n >= 1 && n <= 65535 && B >= 1 && B <= 65535 && O >= 1 && max_I >= 1, "heads_bwd: bad sizes");
    hipLaunchKernelGGL(head_bwd_data_kernel, dim3((max_I + 63) / 64, B, n), dim3(1024), 0, s, (const HeadEntry*)dev_entries, g, ldg, gms_all, O);
    hipLaunchKernelGGL(head_bwd_param_kernel, dim3(64, n), dim3(256), 0, s, (const HeadEntry*)dev_entries, g, ldg, musig_all, gw_all,
                       gb_all, B, O);
    DGE_LAUNCH_CHECK("heads_bwd");
    return 0;
}

extern "C" int dge_act_bwd(const void* gup, const void* a, const float* noise, void* gpre, float* red, int red_cols, int B, int H,
                           int W, int C, int pool, float scale, float slope, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(CHAN_OK(C, ep), "act_bwd: unsupported channel count %d", C);
    DGE_CHECK(red_cols == 2 || red_cols == 3, "act_bwd: red_cols must be 2 or 3 (got %d)", red_cols);
    dim3 grid(dge_stream_grid(H * W, 256 / (C / ep), B), B);
#define AB(T, NS) hipLaunchKernelGGL((act_bwd_kernel<T, NS>), grid, dim3(256), 0, s, (const T*)gup, (const T*)a, noise, (T*)gpre, red, H, W, C, pool, scale, slope)
    if (dtype == DGE_BF16) { if (red_cols == 3) AB(bf16_t, 3); else AB(bf16_t, 2); }
    else { if (red_cols == 3) AB(float, 3); else AB(float, 2); }
#undef AB
    DGE_LAUNCH_CHECK("act_bwd");
    return 0;
}

extern "C" int dge_act_bwd_mask(const void* gup, const unsigned* mask, const float* noise, void* gpre, float* red, int red_cols, int B, int H,
                                int W, int C, float scale, float slope, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(CHAN_OK(C, ep) && H % 2 == 0 && W % 2 == 0 && mask, "act_bwd_mask: unsupported shape (C %d, %d x %d)", C, H, W);
    DGE_CHECK(red_cols == 2 || red_cols == 3, "act_bwd_mask: red_cols must be 2 or 3 (got %d)", red_cols);
    dim3 grid(dge_stream_grid(H * W / 4, 256 / (C / ep), B), B);
#define AB(T, NS) hipLaunchKernelGGL((act_bwd_mask_kernel<T, NS>), grid, dim3(256), 0, s, (const T*)gup, mask, noise, (T*)gpre, red, H, W, C, scale, slope)
    if (dtype == DGE_BF16) { if (red_cols == 3) AB(bf16_t, 3); else AB(bf16_t, 2); }
    else { if (red_cols == 3) AB(float, 3); else AB(float, 2); }
#undef AB
    DGE_LAUNCH_CHECK("act_bwd_mask");
    return 0;
}

extern "C" int dge_in_bwd_coef_slots(const float* dots, int nslot, const float* gms, const float* musig, const float* sc,
                                     const float* sh, float* coef, int B, int C, int npix, hipStream_t s) {
    DGE_CHECK(nslot >= 1, "in_bwd_coef: nslot %d", nslot);
    hipLaunchKernelGGL(in_bwd_coef_kernel, dim3((B * C + 63) / 64), dim3(64), 0, s, dots, gms, musig, sc, sh, coef, B, C, 1.0f / (float)npix, nslot);
    DGE_LAUNCH_CHECK("in_bwd_coef");
    return 0;
}
extern "C" int dge_in_bwd_coef(const float* dots, const float* gms, const float* musig, const float* sc, const float* sh,
                               float* coef, int B, int C, int npix, hipStream_t s) {
    return dge_in_bwd_coef_slots(dots, 1, gms, musig, sc, sh, coef, B, C, npix, s);
}

static int in_bwd_launch(const void* gy, const void* x, const float* coef, const InCoefSrc& cs, const void* extra, const float* noise, void* gout,
                         float* red, int B, int H, int W, int C, int extra_pool, float extra_scale, int act, int dtype, hipStream_t s,
                         const float* img = nullptr, float* fr_out = nullptr) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(CHAN_OK(C, ep), "in_bwd: unsupported channel count %d", C);
    dim3 grid(dge_stream_grid(H * W, 256 / (C / ep), B), B);
#define DGE_IB(T, FR) hipLaunchKernelGGL((in_bwd_kernel<T, FR>), grid, dim3(256), 0, s, (const T*)gy, (const T*)x, coef, (const T*)extra, noise, (T*)gout, red, H, W, C, extra_pool, extra_scale, act, cs, img, fr_out)
    if (img) { if (dtype == DGE_BF16) DGE_IB(bf16_t, true); else DGE_IB(float, true); }
    else { if (dtype == DGE_BF16) DGE_IB(bf16_t, false); else DGE_IB(float, false); }
#undef DGE_IB
    DGE_LAUNCH_CHECK("in_bwd");
    return 0;
}
extern "C" int dge_in_bwd(const void* gy, const void* x, const float* coef, const void* extra, const float* noise, void* gout,
                          float* red, int B, int H, int W, int C, int extra_pool, float extra_scale, int act, int dtype, hipStream_t s) {
    DGE_CHECK(coef, "in_bwd: null coefficients");
    return in_bwd_launch(gy, x, coef, InCoefSrc{}, extra, noise, gout, red, B, H, W, C, extra_pool, extra_scale, act, dtype, s);
}
// dge_in_bwd_coef_slots + dge_in_bwd in one launch (C <= 512): every workgroup computes its sample's coefficients itself
extern "C" int dge_in_bwd_fused(const void* gy, const void* x, const float* dots, int nslot, const float* gms, const float* musig,
                                const float* sc, const float* sh, int npix, const void* extra, const float* noise, void* gout, float* red,
                                int B, int H, int W, int C, int extra_pool, float extra_scale, int act, int dtype, hipStream_t s) {
    DGE_CHECK(C <= 512 && nslot >= 1 && musig && sc && sh && npix > 0, "in_bwd_fused: needs C <= 512, musig, sc, sh");
    InCoefSrc cs{dots, gms, musig, sc, sh, nslot, B, 1.0f / (float)npix};
    return in_bwd_launch(gy, x, nullptr, cs, extra, noise, gout, red, B, H, W, C, extra_pool, extra_scale, act, dtype, s);
}

// dge_in_bwd_fused (act = 0) + dge_fromrgb_bwd in one launch: x is the FromRGB output x0, the gradient w.r.t. x0 is reduced, not stored
extern "C" int dge_in_bwd_fromrgb(const void* gy, const void* x0, const float* dots, int nslot, const float* gms, const float* musig,
                                  const float* sc, const float* sh, int npix, const void* extra, const float* img, float* out4,
                                  int B, int H, int W, int C, int extra_pool, float extra_scale, int dtype, hipStream_t s) {
    DGE_CHECK(C <= 512 && nslot >= 1 && musig && sc && sh && npix > 0 && img && out4, "in_bwd_fromrgb: needs C <= 512, musig, sc, sh, img, out4");
    InCoefSrc cs{dots, gms, musig, sc, sh, nslot, B, 1.0f / (float)npix};
    return in_bwd_launch(gy, x0, nullptr, cs, extra, nullptr, nullptr, nullptr, B, H, W, C, extra_pool, extra_scale, 0, dtype, s, img, out4);
}

extern "C" int dge_chan_sum(const void* x, float* out, int B, int HW, int C, float scale, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(CHAN_OK(C, ep), "chan_sum: unsupported channel count %d", C);
    dim3 grid(dge_stream_grid(HW, 256 / (C / ep), B), B);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(chan_sum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, out, HW, C, scale);
    else hipLaunchKernelGGL(chan_sum_kernel<float>, grid, dim3(256), 0, s, (const float*)x, out, HW, C, scale);
    DGE_LAUNCH_CHECK("chan_sum");
    return 0;
}

extern "C" int dge_fromrgb_bwd(const void* gx, const void* x0, const float* img, float* out4, int B, int HW, int C, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(CHAN_OK(C, ep), "fromrgb_bwd: unsupported channel count %d", C);
    dim3 grid(dge_stream_grid(HW, 256 / (C / ep), B), B);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(fromrgb_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)gx, (const bf16_t*)x0, img, out4, HW, C);
    else hipLaunchKernelGGL(fromrgb_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)gx, (const float*)x0, img, out4, HW, C);
    DGE_LAUNCH_CHECK("fromrgb_bwd");
    return 0;
}

extern "C" int dge_fromrgb_dgrad(const void* gx, const void* x0, const float* w, float* gimg, int B, int HW, int C, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(CHAN_OK(C, ep), "fromrgb_dgrad: unsupported channel count %d", C);
    dim3 grid(dge_stream_grid(HW, 256 / (C / ep), B), B);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(fromrgb_dgrad_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)gx, (const bf16_t*)x0, w, gimg, HW, C);
    else hipLaunchKernelGGL(fromrgb_dgrad_kernel<float>, grid, dim3(256), 0, s, (const float*)gx, (const float*)x0, w, gimg, HW, C);
    DGE_LAUNCH_CHECK("fromrgb_dgrad");
    return 0;
}

extern "C" int dge_nearest_up2(const void* x, void* y, int B, int H, int W, int C, float scale, int dtype, hipStream_t s) {
    const int ep = dtype == DGE_BF16 ? 8 : 4;
    DGE_CHECK(C % ep == 0, "nearest_up2: channels must be a multiple of %d", ep);
    const long total16 = (long)B * 4 * H * W * (C / ep);
    const unsigned grid = (unsigned)((total16 + 255) / 256 > 65535 ? 65535 : (total16 + 255) / 256);
    if (dtype == DGE_BF16) hipLaunchKernelGGL(nearest_up2_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, H, W, C, scale, total16);
    else hipLaunchKernelGGL(nearest_up2_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, H, W, C, scale, total16);
    DGE_LAUNCH_CHECK("nearest_up2");
    return 0;
}

extern "C" int dge_dense_wgrad(const float* gy, int ldgy, const float* x, int ldx, float* gw, float* gb, int B, int O, int I,
                               int accumulate, hipStream_t s) {
    const long n = (long)O * I;
    hipLaunchKernelGGL(dense_wgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gy, ldgy, x, ldx, gw, gb, B, O, I, accumulate);
    DGE_LAUNCH_CHECK("dense_wgrad");
    return 0;
}
