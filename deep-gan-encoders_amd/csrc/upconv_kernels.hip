// StyleGAN2 up layer (reference model/stylegan2_generator.py:879-896: conv_transpose2d(stride 2, flipped 3x3) followed by
// the 4x4 FIR [1,3,3,1]^2/64*4, then noise / bias / lrelu*sqrt2 :911-921) at its ALGORITHMIC cost.
//
// The folded form used by conv_igemm (one 3x3 conv per output phase, N = 4*Cout) executes 36 tap-MACs per input pixel.
// Here the two factors stay separate inside one kernel:
//   1. MFMA stage: the (2H+1)^2 transposed-conv result t, in phase form.  t[2m+u, 2n+v] for u,v in {0,1} needs
//      x[m-a, n-b] with a in {0,1} if u == 0 else {0} (same for b/v): 4 + 2 + 2 + 1 = 9 tap-MACs per input pixel,
//      the count of the transposed conv itself:  t[2m]   = x[m] W[2] + x[m-1] W[0],   t[2m+1] = x[m] W[1]   (per axis).
//   2. VALU stage: the separable 4-tap FIR over t, which never leaves LDS, fused with demodulation scale, noise, bias and
//      activation: y[Y] = sum_k k1[k] t[Y+k-1], k1 = [1,3,3,1]/4.
// A workgroup owns a 16x16 block of t-pixels (m,n) = 32x32 t values per channel, of which the FIR can finish 28x28
// outputs (it needs t[Y-1..Y+2]); tiles therefore advance by 14 input pixels and overlap by 2 (MAC efficiency 0.77:
// 11.8 effective tap-MACs per input pixel against 36).  N tile = 32 output channels x 4 phases; each wave owns 64
// t-pixels x all 4 phases (128 accumulator registers), so the 9 (phase, tap) products are balanced across waves by
// construction and the 4 shifted activation fragments are shared by the phases that use them.
#include <type_traits>
#include <stdlib.h>
#include "common.h"
#include "../../include/dge_hip.h"

template <typename T> struct MmaU;
template <> struct MmaU<bf16_t> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&a, *(const bf16x8_t*)&b, c, 0, 0, 0);
    }
};
template <> struct MmaU<float> {
    __device__ static __forceinline__ void run(const uint4& a, const uint4& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

template <int N, int I = 0> struct SFor {
    template <class F> __device__ static __forceinline__ void run(F&& f) { f(std::integral_constant<int, I>{}); SFor<N, I + 1>::run(f); }
};
template <int N> struct SFor<N, N> { template <class F> __device__ static __forceinline__ void run(F&&) {} };

template <int N> struct RegsU {
    uint4 v; RegsU<N - 1> rest;
    template <int I> __device__ __forceinline__ uint4& get() { if constexpr (I == 0) return v; else return rest.template get<I - 1>(); }
};
template <> struct RegsU<0> { template <int I> __device__ __forceinline__ uint4& get(); };

// 16-byte-per-lane global -> LDS DMA from inline asm (not tracked by the compiler's waitcnt insertion: awaited by hand with
// s_waitcnt vmcnt(0) before the barrier that publishes the buffer).  LDS destination = wave-uniform base (M0) + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}

// the same with a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset
__device__ __forceinline__ void glds16_saddr_u(unsigned voff, unsigned long long sbase, unsigned lds_dst_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst_uniform) : "memory");
}

struct UpParams {
    const void* x; const void* w; void* y;
    const float* in_scale; const float* out_scale; const float* noise; const float* noise_w; const float* bias;
    int B, H, W, Cin, Cout;
    int noise_bstride, noise_w_stride, act;
    float bias_scale, gain;
    int tiles_x, tiles_y;
    int dbg;                  // ablation switches for tuning (DGE_UP_DBG), 0 in production
};

// the 9 (phase, tap) units: phase = u*2+v, shift (a,b), source tap (wy,wx) of the 3x3 kernel
__host__ __device__ constexpr int unit_phase(int q) { return q < 4 ? 0 : (q < 6 ? 1 : (q < 8 ? 2 : 3)); }
__host__ __device__ constexpr int unit_a(int q) { return (q == 2 || q == 3 || q == 5) ? 1 : 0; }
__host__ __device__ constexpr int unit_b(int q) { return (q == 1 || q == 3 || q == 7) ? 1 : 0; }
__host__ __device__ constexpr int unit_wy(int q) { return unit_phase(q) < 2 ? (unit_a(q) ? 0 : 2) : 1; }
__host__ __device__ constexpr int unit_wx(int q) { return (unit_phase(q) & 1) == 0 ? (unit_b(q) ? 0 : 2) : 1; }

template <typename T>
struct UpCfg {
    static constexpr int ESZ = (int)sizeof(T);
    static constexpr int KC = 64 / ESZ;                 // channels per K chunk (64 bytes)
    static constexpr int APS = 80;                      // halo pixel stride (64 B + 16 B pad: conflict-free ds_read_b128)
    static constexpr int ARP = 17 * APS;                // halo row pitch
    static constexpr int A_BYTES = ((17 * ARP + 255) / 256) * 256;
    static constexpr int BRS = 64;                      // weight rows unpadded (LDS-DMA image), 16-byte chunks XOR-swizzled
    static constexpr int B_BYTES = 9 * 32 * BRS;        // one K chunk of all 9 units = 18 x 1 KiB DMA pieces
    // t tile.  f32: [32 rows][32 cols] pixels of 32 channels (+16 B pad).  bf16: a 32-bit word holds the t values of two
    // horizontally adjacent columns (2j, 2j+1) of one channel -- the two v phases of one accumulator lane -- so that the
    // horizontal FIR is v_dot2_f32_bf16 on whole words (no unpacking): [32 rows][16 column pairs] x (32 + 4 pad) words.
    static constexpr int TPS = ESZ == 2 ? 36 * 4 : 32 * 4 + 16;   // stride of one pixel (f32) / one column pair (bf16)
    static constexpr int T_BYTES = (ESZ == 2 ? 32 * 16 : 32 * 32) * TPS;
    static constexpr int AB = A_BYTES + 2 * B_BYTES;    // weights double-buffered
    static constexpr int LDS_BYTES = AB > T_BYTES ? AB : T_BYTES;
    static constexpr int NA = 17 * 17 * 4, NA_PER = (NA + 255) / 256;     // 16-byte staging items
    static constexpr int MINW = LDS_BYTES <= 80 * 1024 ? 2 : 1;
};

template <typename T>
__global__ __launch_bounds__(256, (UpCfg<T>::MINW)) void upconv_fir_kernel(UpParams p) {
    using C = UpCfg<T>;
    constexpr int EP16 = Elem<T>::PER16;
    __shared__ __attribute__((aligned(256))) unsigned char lds[C::LDS_BYTES];
    unsigned char* ldsA = lds;
    unsigned char* ldsB = lds + C::A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    {   // XCD-aware order (workgroup i runs on XCD i % 8): a contiguous tile range per XCD
        const int nb = gridDim.x, per = nb >> 3;
        if (per > 0 && bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
    }
    // channel block fastest: the Cout/32 workgroups that share one activation tile run back to back on the same XCD, so the tile
    // crosses HBM once and is served from that XCD's L2 afterwards (the packed weights, <= 4.7 MB, stay L2 resident anyway)
    const int ncb = p.Cout >> 5;
    const int bn0 = (bid % ncb) * 32; bid /= ncb;
    const int tx_i = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty_i = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int my0 = 14 * ty_i - 1, mx0 = 14 * tx_i - 1;        // first t-pixel of the tile
    const T* __restrict__ Xb = (const T*)p.x + (size_t)b * p.H * p.W * p.Cin;
    const T* __restrict__ Wp = (const T*)p.w;

    f32x16_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // fragment base offsets: t-pixel m of M block i -> halo pixel (my + 1 - a, mx + 1 - b)
    int aoff[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int m = (2 * wave + i) * 32 + l31;
        aoff[i] = ((m >> 4) + 1) * C::ARP + ((m & 15) + 1) * C::APS + lh * 16;
    }
    const int brow = l31 * C::BRS, bsw = (l31 >> 2) & 3;       // weight row and its chunk swizzle

    const int nchunks = p.Cin / C::KC;
    const int achunk = tid & 3;
    RegsU<C::NA_PER> areg;
    float asc[EP16];
    // Staging items of this thread, fixed for the whole K loop: 32-bit byte offsets from the (wave-uniform) sample / weight
    // base, so that no 64-bit pointers stay live across the MFMA section (they were spilled, and every scratch reload put a
    // vmcnt(0) between the global loads of a chunk: one full memory latency per load instead of one per chunk).
    // Out-of-image halo pixels load from a clamped address and are zeroed when written to LDS: no divergent branches.
    unsigned gaoff[C::NA_PER];
    unsigned inmask = 0;
    SFor<C::NA_PER>::run([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int idx = tid + i * 256;
        const int pix = idx >> 2;
        const int hy = pix / 17, hx = pix - hy * 17;
        const int gy = my0 - 1 + hy, gx = mx0 - 1 + hx;
        const bool inside = ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W) & (idx < C::NA);
        const int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
        gaoff[i] = (unsigned)(((cy * p.W + cx) * p.Cin + achunk * EP16) * C::ESZ);
        inmask |= (inside ? 1u : 0u) << i;
    });
    // weight chunk kc of all 9 units -> LDS buffer `buf`: 18 pieces of 1 KiB (16 rows x 64 B), piece pc by wave pc % 4;
    // lane -> (row r = 16 pc + lane/4, source chunk c = lane%4 ^ swizzle(r)), destination lane-linear
    const unsigned ldsB_off = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)ldsB;
    // the per-lane part of the source address does not depend on the chunk: one 32-bit byte offset per piece, computed once;
    // a chunk adds a scalar base, so a piece is one M0 write + one instruction (as in conv_igemm)
    unsigned boffu[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int pc = wave + 4 * k;
        const int r = pc * 16 + (lane >> 2);                     // q*32 + n
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        boffu[k] = (unsigned)((((size_t)(r >> 5) * p.Cout + bn0 + (r & 31)) * p.Cin + c * EP16) * sizeof(T));
    }
    auto dma_b = [&](int kc, int buf) {
        const unsigned long long sb = (unsigned long long)Wp + (size_t)kc * C::KC * sizeof(T);
        SFor<5>::run([&](auto kcst) {
            constexpr int k = decltype(kcst)::value;
            const int pc = wave + 4 * k;
            if (pc < 18) glds16_saddr_u(boffu[k], sb, __builtin_amdgcn_readfirstlane(ldsB_off + buf * C::B_BYTES + pc * 1024));
        });
    };
    const unsigned char* __restrict__ Xbc = (const unsigned char*)Xb;

    auto load_ab = [&](int kc) {
        const unsigned koff = (unsigned)(kc * 64);
        if (p.in_scale) {
            const float* sp = p.in_scale + (size_t)b * p.Cin + kc * C::KC + achunk * EP16;
#pragma unroll
            for (int e4 = 0; e4 < EP16 / 4; e4++) *(float4*)&asc[e4 * 4] = *(const float4*)(sp + e4 * 4);
        } else {
#pragma unroll
            for (int e = 0; e < EP16; e++) asc[e] = 1.f;
        }
        SFor<C::NA_PER>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            areg.template get<i>() = *(const uint4*)(Xbc + (gaoff[i] + koff));
        });
    };
    auto store_ab = [&]() {
        SFor<C::NA_PER>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int idx = tid + i * 256;
            if (idx < C::NA) {
                const int pix = idx >> 2;
                const int hy = pix / 17, hx = pix - hy * 17;
                float f[EP16];
                unpack16(areg.template get<i>(), f, (T*)nullptr);
                const float keep = ((inmask >> i) & 1u) ? 1.f : 0.f;       // padding is zero
#pragma unroll
                for (int e = 0; e < EP16; e++) f[e] *= asc[e] * keep;
                *(uint4*)(ldsA + hy * C::ARP + hx * C::APS + achunk * 16) = pack16(f, (T*)nullptr);
            }
        });
    };

    dma_b(0, 0);
    load_ab(0);
    store_ab();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the untracked weight DMA
    __syncthreads();
    for (int kc = 0; kc < ((p.dbg & 1) ? 0 : nchunks); kc++) {
        if (kc + 1 < nchunks) { if (!(p.dbg & 8)) dma_b(kc + 1, (kc + 1) & 1); if (!(p.dbg & 16)) load_ab(kc + 1); }     // in flight behind the MFMAs of this chunk
        const unsigned char* bcur = ldsB + (kc & 1) * C::B_BYTES;
        {
            // 18 steps = 2 K slices x 9 (phase, tap) units, software-pipelined by hand: the weight fragment of step t+3 is
            // requested before the two MFMAs of step t (an LDS read takes ~3 MFMA pairs to return), and the units are ordered
            // so that each shifted activation fragment is re-loaded for the second K slice right after its last use in the first.
            constexpr int ORD[9] = {3, 2, 5, 1, 7, 0, 4, 6, 8};      // shifts (1,1) | (1,0) x2 | (0,1) x2 | (0,0) x4
            uint4 af[2][4], bf[4];
            auto lda = [&](int ks, int sft) {
#pragma unroll
                for (int i = 0; i < 2; i++)
                    af[i][sft] = *(const uint4*)(ldsA + aoff[i] - (sft >> 1) * C::ARP - (sft & 1) * C::APS + ks * 32);
            };
            auto ldb = [&](int t) -> uint4 {
                return *(const uint4*)(bcur + ORD[t % 9] * 32 * C::BRS + brow + ((((t / 9) * 2 + lh) ^ bsw) << 4));
            };
            lda(0, 3); bf[0] = ldb(0); lda(0, 2); bf[1] = ldb(1); bf[2] = ldb(2); lda(0, 1); lda(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            SFor<18>::run([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int q = ORD[t % 9];
                constexpr int sft = unit_a(q) * 2 + unit_b(q);
                if constexpr (t + 3 < 18) bf[(t + 3) & 3] = ldb(t + 3);
                __builtin_amdgcn_sched_barrier(0);       // keep the prefetch ABOVE the MFMAs (the scheduler sinks it to its use otherwise)
                MmaU<T>::run(af[0][sft], bf[t & 3], acc[0][unit_phase(q)]);
                MmaU<T>::run(af[1][sft], bf[t & 3], acc[1][unit_phase(q)]);
                if constexpr (t == 0) lda(1, 3);
                if constexpr (t == 2) lda(1, 2);
                if constexpr (t == 4) lda(1, 1);
                if constexpr (t == 8) lda(1, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // next weight chunk landed (and the halo registers)
        __syncthreads();                                          // every wave has read chunk kc
        if (kc + 1 < nchunks) {
            store_ab();
            __syncthreads();
        }
    }

    // ---------------------------------------------------------------- epilogue
    // thread = (output column xc, channel quad cq).  The noise column of the tile is requested now, so that its latency
    // hides behind the t store and the barrier (a dependent global read per FIR row serialised the whole loop).
    const int cq = tid & 7, xc = tid >> 3;
    const int OH = 2 * p.H, OW = 2 * p.W;
    const int ox = 28 * tx_i + xc;
    const bool col_ok = (xc < 28) & (ox < OW);
    float nzr[28];
    {
        const float* __restrict__ nzp = p.noise ? p.noise + (size_t)b * p.noise_bstride : nullptr;
#pragma unroll
        for (int yl = 0; yl < 28; yl++) {
            const int oy = 28 * ty_i + yl;
            nzr[yl] = (nzp && col_ok && oy < OH) ? nzp[(size_t)oy * OW + ox] : 0.f;
        }
    }
    unsigned char* ldsT = lds;
    const int o0 = bn0 + cq * 4;
    const float slope = p.act == DGE_ACT_LRELU ? 0.2f : (p.act == DGE_ACT_RELU ? 0.f : 1.f);
    T* __restrict__ Yb = (T*)p.y + (size_t)b * OH * OW * p.Cout;
    if constexpr (C::ESZ == 2) {
        // ---- t -> LDS as packed column pairs: lane (channel l31) holds both v phases of a t-pixel
        typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
        typedef __attribute__((ext_vector_type(2))) float f2_t;
        if (!(p.dbg & 2)) {
            unsigned char* tl = ldsT + ((2 * 4 * wave) * 16 + 4 * lh) * C::TPS + l31 * 4;      // row 2*(4 wave), pair 4 lh
            SFor<2>::run([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                SFor<2>::run([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        // m = 64 wave + 32 i + 8 (r>>2) + (r&3) + 4 lh  ->  t row 2 (m>>4) + u, column pair m & 15
                        const int trow = 2 * (2 * i + (r >> 3)) + u, pair = 8 * ((r >> 2) & 1) + (r & 3);
                        *(unsigned*)(tl + (trow * 16 + pair) * C::TPS) = pack2bf(acc[i][2 * u][r], acc[i][2 * u + 1][r]);
                    }
                });
            });
        }
        __syncthreads();
        if (col_ok && !(p.dbg & 4)) {
            // horizontal taps of output column xc: t columns xc+1 .. xc+4.  xc odd: two aligned words (k0,k1),(k1,k0);
            // xc even: three words (0,k0),(k1,k1),(k0,0).  Same code for both: coefficient words + first pair index.
            const unsigned K0 = 0x3e80u, K1 = 0x3f40u;            // bf16 0.25, 0.75
            const bool odd = xc & 1;
            const unsigned cA = odd ? (K0 | (K1 << 16)) : (K0 << 16), cB = odd ? (K1 | (K0 << 16)) : (K1 | (K1 << 16)), cC = odd ? 0u : K0;
            const int j0 = odd ? (xc + 1) >> 1 : xc >> 1;
            const int j2 = min(j0 + 2, 15);                       // (coefficient 0 when clamped)
            f2_t osc[2], bia[2], nwv[2];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                osc[e >> 1][e & 1] = (p.out_scale ? p.out_scale[(size_t)b * p.Cout + o0 + e] : 1.f) * p.gain;
                bia[e >> 1][e & 1] = p.bias ? p.bias[o0 + e] * p.bias_scale * p.gain : 0.f;
                nwv[e >> 1][e & 1] = p.noise ? p.noise_w[(o0 + e) * p.noise_w_stride] * p.gain : 0.f;
            }
            const unsigned char* tb = ldsT + cq * 16;
            const size_t rstride = (size_t)OW * p.Cout;
            T* dst = Yb + ((size_t)(28 * ty_i) * OW + ox) * p.Cout + o0;
            f2_t h[4][2];
#pragma unroll
            for (int a = 0; a < 4; a++) { h[a][0] = f2_t{0.f, 0.f}; h[a][1] = f2_t{0.f, 0.f}; }
            SFor<31>::run([&](auto rc) {
                constexpr int r = decltype(rc)::value + 1;
                const uint4 wa = *(const uint4*)(tb + (r * 16 + j0) * C::TPS);
                const uint4 wb = *(const uint4*)(tb + (r * 16 + j0 + 1) * C::TPS);
                const uint4 wc = *(const uint4*)(tb + (r * 16 + j2) * C::TPS);
                const unsigned ua[4] = {wa.x, wa.y, wa.z, wa.w}, ub[4] = {wb.x, wb.y, wb.z, wb.w}, uc[4] = {wc.x, wc.y, wc.z, wc.w};
                float hn[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float v = __builtin_amdgcn_fdot2_f32_bf16(*(const bf2_t*)&ua[e], *(const bf2_t*)&cA, 0.f, false);
                    v = __builtin_amdgcn_fdot2_f32_bf16(*(const bf2_t*)&ub[e], *(const bf2_t*)&cB, v, false);
                    hn[e] = __builtin_amdgcn_fdot2_f32_bf16(*(const bf2_t*)&uc[e], *(const bf2_t*)&cC, v, false);
                }
                h[0][0] = h[1][0]; h[0][1] = h[1][1]; h[1][0] = h[2][0]; h[1][1] = h[2][1]; h[2][0] = h[3][0]; h[2][1] = h[3][1];
                h[3][0] = f2_t{hn[0], hn[1]}; h[3][1] = f2_t{hn[2], hn[3]};
                constexpr int yl = r - 4;
                if constexpr (yl >= 0) {
                    if (28 * ty_i + yl < OH) {
                        unsigned outw[2];
#pragma unroll
                        for (int g = 0; g < 2; g++) {
                            const f2_t f = (h[0][g] + h[3][g]) * 0.25f + (h[1][g] + h[2][g]) * 0.75f;
                            const f2_t uu = f * osc[g] + (nwv[g] * nzr[yl] + bia[g]);
                            const f2_t lo = uu * slope;
                            outw[g] = pack2bf(fmaxf(uu[0], lo[0]), fmaxf(uu[1], lo[1]));
                        }
                        *(uint2*)(dst + yl * rstride) = make_uint2(outw[0], outw[1]);
                    }
                }
            });
        }
    } else {
        // ---- f32 (parity path): plain layout, plain arithmetic
        if (!(p.dbg & 2))
        SFor<2>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            SFor<4>::run([&](auto pc) {
                constexpr int ph = decltype(pc)::value;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = (2 * wave + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int trow = 2 * (m >> 4) + (ph >> 1), tcol = 2 * (m & 15) + (ph & 1);
                    *((float*)(ldsT + (trow * 32 + tcol) * C::TPS) + l31) = acc[i][ph][r];
                }
            });
        });
        __syncthreads();
        if (col_ok && !(p.dbg & 4)) {
            float osc[4], bia[4], nw[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                osc[e] = (p.out_scale ? p.out_scale[(size_t)b * p.Cout + o0 + e] : 1.f) * p.gain;
                bia[e] = p.bias ? p.bias[o0 + e] * p.bias_scale * p.gain : 0.f;
                nw[e] = p.noise ? p.noise_w[(o0 + e) * p.noise_w_stride] * p.gain : 0.f;
            }
            const float k1[4] = {0.25f, 0.75f, 0.75f, 0.25f};
            float h[4][4];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int e = 0; e < 4; e++) h[a][e] = 0.f;
            const unsigned char* tbase = ldsT + (xc + 1) * C::TPS + cq * 16;
            SFor<31>::run([&](auto rc) {
                constexpr int r = decltype(rc)::value + 1;
                float hn[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int bq = 0; bq < 4; bq++) {
                    const float4 tv = *(const float4*)(tbase + (r * 32 + bq) * C::TPS);
                    hn[0] = fmaf(k1[bq], tv.x, hn[0]); hn[1] = fmaf(k1[bq], tv.y, hn[1]);
                    hn[2] = fmaf(k1[bq], tv.z, hn[2]); hn[3] = fmaf(k1[bq], tv.w, hn[3]);
                }
#pragma unroll
                for (int e = 0; e < 4; e++) { h[0][e] = h[1][e]; h[1][e] = h[2][e]; h[2][e] = h[3][e]; h[3][e] = hn[e]; }
                constexpr int yl = r - 4;
                if constexpr (yl >= 0) {
                    const int oy = 28 * ty_i + yl;
                    if (oy < OH) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float f = k1[0] * h[0][e] + k1[1] * h[1][e] + k1[2] * h[2][e] + k1[3] * h[3][e];
                            const float u = fmaf(f, osc[e], fmaf(nw[e], nzr[yl], bia[e]));
                            v[e] = fmaxf(u, u * slope);
                        }
                        *(float4*)(Yb + ((size_t)oy * OW + ox) * p.Cout + o0) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            });
        }
    }
}

// out[q][o][i] = scale * w[o][i][wy(q)][wx(q)]   (w: [Cout,Cin,3,3] f32, the reference's parameter layout)
template <typename T>
__global__ void upconv_pack_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, float scale) {
    const long total = 9L * Cout * Cin;
    const int wy[9] = {unit_wy(0), unit_wy(1), unit_wy(2), unit_wy(3), unit_wy(4), unit_wy(5), unit_wy(6), unit_wy(7), unit_wy(8)};
    const int wx[9] = {unit_wx(0), unit_wx(1), unit_wx(2), unit_wx(3), unit_wx(4), unit_wx(5), unit_wx(6), unit_wx(7), unit_wx(8)};
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
        const int i = idx % Cin; const long r = idx / Cin; const int o = r % Cout; const int q = r / Cout;
        Elem<T>::st(out + idx, scale * w[((size_t)o * Cin + i) * 9 + wy[q] * 3 + wx[q]]);
    }
}

// upconv_stream.hip: the streaming form for the 64 -> 32 layer at the top of the generator
bool dge_upconv_stream_ok(int B, int H, int W, int Cin, int Cout, int dtype);
int dge_upconv_stream_launch(const void* x, const void* w_packed, void* y, const float* in_scale, const float* out_scale, const float* noise,
                             int noise_bstride, const float* noise_w, const float* bias, float bias_scale, float gain, int act,
                             int B, int H, int W, hipStream_t s);

extern "C" int dge_upconv_supported(int Cin, int Cout, int dtype) {
    const int kc = dtype == DGE_BF16 ? 32 : 16;
    return (Cin % kc == 0 && Cout % 32 == 0) ? 1 : 0;
}

extern "C" int dge_pack_upconv_weight(const float* w, void* out, int Cout, int Cin, float scale, int dtype, hipStream_t s) {
    const long total = 9L * Cout * Cin;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    if (dtype == DGE_BF16) hipLaunchKernelGGL(upconv_pack_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, w, (bf16_t*)out, Cout, Cin, scale);
    else hipLaunchKernelGGL(upconv_pack_kernel<float>, dim3(grid), dim3(256), 0, s, w, (float*)out, Cout, Cin, scale);
    DGE_LAUNCH_CHECK("pack_upconv_weight");
    return 0;
}

extern "C" int dge_upconv_fir(const void* x, const void* w_packed, void* y, const float* in_scale, const float* out_scale,
                              const float* noise, int noise_bstride, const float* noise_w, int noise_w_stride, const float* bias,
                              float bias_scale, float gain, int act, int B, int H, int W, int Cin, int Cout, int dtype,
                              hipStream_t s) {
    DGE_CHECK(dge_upconv_supported(Cin, Cout, dtype), "upconv_fir: Cin=%d must be a multiple of the 64-byte K chunk and Cout=%d of 32", Cin, Cout);
    DGE_CHECK(gain > 0.f, "upconv_fir: the gain is folded into scale / noise / bias and must be positive");
    DGE_CHECK(!noise || noise_w, "upconv_fir: noise needs its weight");
    if ((!noise || noise_w_stride == 0) && dge_upconv_stream_ok(B, H, W, Cin, Cout, dtype))
        return dge_upconv_stream_launch(x, w_packed, y, in_scale, out_scale, noise, noise_bstride, noise_w, bias, bias_scale, gain, act, B, H, W, s);
    UpParams p;
    p.x = x; p.w = w_packed; p.y = y; p.in_scale = in_scale; p.out_scale = out_scale; p.noise = noise; p.noise_w = noise_w; p.bias = bias;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.noise_bstride = noise_bstride; p.noise_w_stride = noise_w_stride; p.act = act; p.bias_scale = bias_scale; p.gain = gain;
    p.tiles_x = (W + 13) / 14; p.tiles_y = (H + 13) / 14;
    p.dbg = dge_env().up_dbg;
    const long nblk = (long)p.tiles_x * p.tiles_y * B * (Cout / 32);
    dge_note_kernel("upconv_fir<%s>", dtype == DGE_BF16 ? "bf16" : "f32");
    if (dtype == DGE_BF16) hipLaunchKernelGGL(upconv_fir_kernel<bf16_t>, dim3((unsigned)nblk), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(upconv_fir_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, s, p);
    DGE_LAUNCH_CHECK("upconv_fir");
    return 0;
}
