// Shared device helpers for the gfx950 kernels (NHWC activations, bf16 or f32 storage).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

enum { DGE_F32 = 0, DGE_BF16 = 1 };
enum { DGE_ACT_NONE = 0, DGE_ACT_LRELU = 1, DGE_ACT_RELU = 2 };

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {           // round-to-nearest-even
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);   // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// two f32 -> packed bf16 pair, round-to-nearest-even in hardware (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
#ifdef DGE_SOFT_BF16
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
#endif
    f32x2_t v = {lo, hi};
    bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    return *(uint32_t*)&h;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N)
#include <type_traits>
template <int N, int I = 0> struct StaticFor {
    template <class F> __device__ static __forceinline__ void run(F&& f) { f(std::integral_constant<int, I>{}); StaticFor<N, I + 1>::run(f); }
};
template <int N> struct StaticFor<N, N> { template <class F> __device__ static __forceinline__ void run(F&&) {} };

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int PER16 = 4;
    __device__ static float ld(const float* p) { return *p; }
    __device__ static void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int PER16 = 8;
    __device__ static float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// unpack a 16-byte chunk into floats / pack back
__device__ __forceinline__ void unpack16(const uint4& v, float (&f)[4], float*) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
__device__ __forceinline__ void unpack16(const uint4& v, float (&f)[8], bf16_t*) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack16(const float (&f)[4], float*) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
__device__ __forceinline__ uint4 pack16(const float (&f)[8], bf16_t*) {
    return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == DGE_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == DGE_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------ deterministic mode (dge_set_deterministic)
// Reductions that end in same-address f32 atomics are order dependent run to run (the reference pins
// torch.backends.cudnn.deterministic = True, training_utils.py:51).  In deterministic mode every such reduction goes through a
// library-owned workspace instead: each contributor stores its partial vector in ITS slot (plain stores), takes a ticket, and
// the LAST contributor of the domain sums the slots in slot order and hands the totals on.  No contributor ever waits for
// another one (no assumption on dispatch order); cross-CU visibility follows the release -> ticket -> acquire recipe.
// One copy of the state per translation unit (no relocatable device code); capi.hip updates all of them.
struct DgeDet {
    int enabled;
    float* ws;                 // [domain][slot][L] partial vectors of the running launch
    unsigned* counters;        // one arrival counter per domain, zero between launches (reset by the last arriver)
    long long ws_floats;
    int ncounters;
};
static __device__ DgeDet g_det = {0, nullptr, nullptr, 0, 0};
void dge_det_register(void (*setter)(const DgeDet*));
extern int dge_det_upload_failed;                 // capi.hip: set by a translation unit whose copy of the state could not be uploaded
// host-side capacity check of a deterministic-mode launch (ndomains x nslots vectors of L floats, one counter per domain):
// launchers refuse instead of letting det_slot() trap on the device.  True when the mode is off.
bool dge_det_fits(long long ndomains, long long nslots, long long L);
static void dge_det_set_this_tu(const DgeDet* v) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_det), v, sizeof(DgeDet), 0, hipMemcpyHostToDevice) != hipSuccess) dge_det_upload_failed = 1;
}
namespace { struct DgeDetReg { DgeDetReg() { dge_det_register(&dge_det_set_this_tu); } }; static DgeDetReg dge_det_reg_instance; }

__device__ __forceinline__ bool det_on() { return g_det.enabled != 0; }
__device__ __forceinline__ float* det_slot(int domain, int ndomains, int slot, int nslots, int L) {
    if ((long long)ndomains * nslots * L > g_det.ws_floats || ndomains > g_det.ncounters) __builtin_trap();   // loud: the launch aborts
    return g_det.ws + ((size_t)domain * nslots + slot) * L;
}
// Workgroup-level close of a domain: call after EVERY thread of the workgroup has stored its part of the slot.  Returns true
// (to all threads) in the last workgroup to arrive, after which it may read all `nslots` slots of the domain.
// `arrivals` = number of det_arrive_wg calls per domain.
__device__ __forceinline__ bool det_arrive_wg(int domain, int arrivals) {
    __shared__ int s_last;
    // every wave drains ITS slot stores to L2 before the barrier; thread 0's agent-scope release (buffer_wbl2) then writes the
    // XCD's dirty lines back, so the ticket cannot overtake another wave's stores (the fence alone waits only for wave 0's)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = atomicAdd(&g_det.counters[domain], 1u);
        const int last = (t == (unsigned)(arrivals - 1));
        if (last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); g_det.counters[domain] = 0u; }
        s_last = last;
    }
    __syncthreads();
    return s_last != 0;
}
// the same for a single wave (no workgroup barrier: the wave's own stores precede its fence in program order)
__device__ __forceinline__ bool det_arrive_wave(int domain, int arrivals) {
    int last = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) {
        const unsigned t = atomicAdd(&g_det.counters[domain], 1u);
        last = (t == (unsigned)(arrivals - 1));
        if (last) g_det.counters[domain] = 0u;
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return last != 0;
}
// ordered sum of element idx over the slots of a domain
__device__ __forceinline__ float det_sum(int domain, int nslots, int L, int idx) {
    const float* base = g_det.ws + (size_t)domain * nslots * L + idx;
    float s = 0.f;
    for (int k = 0; k < nslots; k++) s += base[(size_t)k * L];
    return s;
}

// block-parallel ordered total of element idx over the slots of a domain (fixed combination order: thread-strided partial
// sums, xor butterfly inside a wave, waves in index order): every thread of the workgroup calls; `red` holds >= 16 floats
__device__ __forceinline__ float det_total_wg(int domain, int nslots, int L, int idx, float* red) {
    const float* base = g_det.ws + (size_t)domain * nslots * L + idx;
    float s = 0.f;
    for (int k = threadIdx.x; k < nslots; k += blockDim.x) s += base[(size_t)k * L];
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += red[w];
    return t;
}

// Per-channel block reduction for NHWC streaming kernels.  Thread t owns channel chunk (t % cpt)
// (EP channels) of pixel slot (t / cpt); each thread holds NS partial sums per channel.  The
// sums of all pixel slots are combined in LDS and added atomically to out[c*NS + k].
// `red` must hold 256*NS*EP floats; all 256 threads must call.
template <int EP, int NS>
__device__ __forceinline__ void block_chan_flush(float (&s)[NS][EP], int cpt, int ppi, float* __restrict__ out_b, int C,
                                                 float* red) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NS; k++)
#pragma unroll
        for (int e = 0; e < EP; e++) red[tid * NS * EP + k * EP + e] = s[k][e];
    __syncthreads();
    if (det_on()) {
        // domain = the workgroups that share out_b (same blockIdx.y / z), slot = blockIdx.x
        const int L = cpt * NS * EP, nslots = gridDim.x, ndom = gridDim.y * gridDim.z, dom = blockIdx.y + gridDim.y * blockIdx.z;
        float* slot = det_slot(dom, ndom, blockIdx.x, nslots, L);
        for (int item = tid; item < L; item += 256) {
            const int ch = item / (NS * EP), r = item % (NS * EP);
            float a = 0.f;
            for (int j = 0; j < ppi; j++) a += red[(j * cpt + ch) * NS * EP + r];
            slot[item] = a;
        }
        if (det_arrive_wg(dom, nslots)) {
            for (int item = tid; item < L; item += 256) {
                const int ch = item / (NS * EP), r = item % (NS * EP);
                const int c = ch * EP + (r % EP);
                if (c < C) out_b[(size_t)c * NS + (r / EP)] += det_sum(dom, nslots, L, item);      // single writer
            }
        }
        __syncthreads();
        return;
    }
    for (int item = tid; item < cpt * NS * EP; item += 256) {
        const int ch = item / (NS * EP), r = item % (NS * EP);
        float a = 0.f;
        for (int j = 0; j < ppi; j++) a += red[(j * cpt + ch) * NS * EP + r];
        const int c = ch * EP + (r % EP);
        if (c < C) atomicAdd(out_b + (size_t)c * NS + (r / EP), a);
    }
    __syncthreads();
}

// Workgroups per sample of the streaming kernels that end in a per-(b,c) atomic flush (block_chan_flush).  Same-address
// f32 atomics retire at ~40 ns, so W workgroups cost W*40 ns on top of the data time; few workgroups underfeed HBM.
// Measured on the E_align step: 1024/sample -> 256/sample +9 % step throughput at B=8; 512 loses 2 % at B=2 against 256.
static inline int dge_stream_grid_cap1();
static inline int dge_stream_grid(int npix, int ppi, int B) {
    int cap = 1024 / (B < 1 ? 1 : B); cap = cap < 96 ? 96 : (cap > 256 ? 256 : cap);
    if (B == 1) cap = dge_stream_grid_cap1();
    const int g = (npix + ppi - 1) / ppi;
    return g > cap ? cap : (g < 1 ? 1 : g);
}

// sum over `nslot` copies of a (sum, sum-of-squares) pair, 8 independent loads in flight (the copies sit `stride` floats apart)
__device__ __forceinline__ float2 sum_slot_pairs(const float* __restrict__ base, size_t stride, int nslot) {
    float s0 = 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 8 <= nslot; k += 8) {
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = *(const float2*)(base + (size_t)(k + j) * stride);
#pragma unroll
        for (int j = 0; j < 8; j++) { s0 += v[j].x; s1 += v[j].y; }
    }
    for (; k < nslot; k++) { const float2 v = *(const float2*)(base + (size_t)k * stride); s0 += v.x; s1 += v.y; }
    return make_float2(s0, s1);
}

// Tuning / test overrides from the environment, read ONCE (at the first launch) instead of a getenv per launch on the production
// path (~10 per conv launch, ~1300 launches per step); dge_env_reload() re-reads them (tests that toggle a switch mid-process).
struct DgeEnv {
    int force_stream, no_stream, stream_nseg;            // DGE_FORCE_STREAM, DGE_NO_STREAM, DGE_STREAM_NSEG (0 = default)
    int no_pw;                                           // DGE_NO_PW: 1x1 launches stay on conv_igemm
    int conv_dbg, conv_bn, conv_kc, conv_small, conv_nok4, conv_nok2;   // DGE_CONV_* (bn / kc 0 = default, small -1 = default)
    int torgb_thread, wgrad_th8, wgrad_groups, up_dbg;   // DGE_TORGB_THREAD, DGE_WGRAD_TH8, DGE_WGRAD_GROUPS (0 = default), DGE_UP_DBG
    int up_variant;                                      // DGE_UP_VARIANT: 0 = by shape, 1 = "pp" (up_pp_kernel), 2 = "s4" (up_s4_kernel)
};
const DgeEnv& dge_env();
// batch 1 (the embedding_img inversion loop): workgroups per sample of the flushing stream kernels.  256 = one workgroup per CU fed HBM at
// 0.7 TB/s (blur_noise_act at 1024^2: 95 us for 67 MB); measured on the loop (hipGraph replay): 256 -> 13.2 ms per iteration, 384 -> 12.8,
// 512 -> 12.7, 768 -> 12.8, 1024 -> 13.1 (the same-address flush takes over).  DGE_STREAM_CAP1 overrides.
static inline int dge_stream_grid_cap1() { static const int v = getenv("DGE_STREAM_CAP1") ? atoi(getenv("DGE_STREAM_CAP1")) : 512; return v < 32 ? 32 : v; }

// error plumbing shared by the C ABI translation units
void dge_set_error(const char* fmt, ...);
// name of the kernel instantiation the calling thread's last conv-family entry point selected (dge_last_kernel)
void dge_note_kernel(const char* fmt, ...);
#define DGE_CHECK(cond, ...) do { if (!(cond)) { dge_set_error(__VA_ARGS__); return -1; } } while (0)
#define DGE_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) { dge_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); return -3; } } while (0)
