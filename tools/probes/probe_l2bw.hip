// Probe (dev tool): how fast can ONE compute unit pull L2-resident bytes?  Sizes the low-resolution convolution design
// (512->512 at 4^2..16^2: every workgroup re-streams its 64-row weight slice, 590 KB, from L2).
//   mode 0: global_load_dwordx4 to VGPRs, U loads in flight per lane
//   mode 1: global_load_lds_dwordx4 into an LDS ring, D pieces in flight per wave
// Each workgroup g reads region (g % NREG) of `bytes` bytes, `reps` times; grid = W workgroups of T threads.
// hipcc --offload-arch=gfx950 -O3 probe_l2bw.hip -o probe_l2bw && ./probe_l2bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

template <int U>
__global__ __launch_bounds__(1024) void k_vgpr(const uint4* __restrict__ src, uint32_t* out, int region_vec, int nreg) {
    const uint4* r = src + (size_t)(blockIdx.x % nreg) * region_vec;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const int T = blockDim.x;
    for (int i = threadIdx.x; i + (U - 1) * T < region_vec; i += U * T) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = r[i + u * T];
#pragma unroll
        for (int u = 0; u < U; u++) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) out[0] = 1;
}

// every wave streams its share of the region through a private LDS ring of D 1-KiB pieces
template <int D>
__global__ __launch_bounds__(1024) void k_lds(const unsigned char* __restrict__ src, uint32_t* out, int region_bytes, int nreg) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const unsigned char* r = src + (size_t)(blockIdx.x % nreg) * region_bytes;
    const unsigned base = lds_off(lds) + wave * D * 1024;
    const int per = region_bytes / nw / 1024;          // pieces per wave
    const unsigned char* g = r + (size_t)wave * per * 1024 + lane * 16;
    for (int i = 0; i < per; i++) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(base + (i % D) * 1024);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g + (size_t)i * 1024), "s"(dst) : "memory");
        if (i >= D - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(D - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (((uint32_t*)lds)[threadIdx.x] == 0x12345u) out[0] = 1;
}

template <typename F>
static float timeit(F launch, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    const int NREG = 8, REGION = 576 * 1024;          // 8 slices of ~590 KB = the packed 512x512x9 bf16 weight
    std::vector<uint32_t> h((size_t)NREG * REGION / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u);
    unsigned char* d; uint32_t* o;
    hipMalloc(&d, (size_t)NREG * REGION); hipMalloc(&o, 64);
    hipMemcpy(d, h.data(), (size_t)NREG * REGION, hipMemcpyHostToDevice);
    printf("region %d KB x %d regions; time per launch, GB/s per workgroup, aggregate TB/s\n", REGION / 1024, NREG);
    for (int W : {8, 64, 256, 512, 1024}) {
        for (int T : {256, 512, 1024}) {
            float t4 = timeit([&] { hipLaunchKernelGGL(k_vgpr<4>, dim3(W), dim3(T), 0, 0, (const uint4*)d, o, REGION / 16, NREG); }, 20);
            float t8 = timeit([&] { hipLaunchKernelGGL(k_vgpr<8>, dim3(W), dim3(T), 0, 0, (const uint4*)d, o, REGION / 16, NREG); }, 20);
            float t16 = timeit([&] { hipLaunchKernelGGL(k_vgpr<16>, dim3(W), dim3(T), 0, 0, (const uint4*)d, o, REGION / 16, NREG); }, 20);
            float l4 = timeit([&] { hipLaunchKernelGGL(k_lds<4>, dim3(W), dim3(T), (T / 64) * 4 * 1024, 0, d, o, REGION, NREG); }, 20);
            float l8 = timeit([&] { hipLaunchKernelGGL(k_lds<8>, dim3(W), dim3(T), (T / 64) * 8 * 1024, 0, d, o, REGION, NREG); }, 20);
            auto gb = [&](float us) { return REGION / us / 1e3; };
            printf("W=%4d T=%4d | vgpr U4 %6.1f us %6.1f GB/s  U8 %6.1f us %6.1f  U16 %6.1f us %6.1f | lds D4 %6.1f us %6.1f GB/s  D8 %6.1f us %6.1f | agg best %.2f TB/s\n",
                   W, T, t4, gb(t4), t8, gb(t8), t16, gb(t16), l4, gb(l4), l8, gb(l8),
                   (double)W * REGION / std::min(std::min(t8, t16), std::min(l4, l8)) / 1e6);
        }
    }
    // small regions: launch floor
    for (int KB : {18, 72, 144}) {
        float t = timeit([&] { hipLaunchKernelGGL(k_vgpr<8>, dim3(256), dim3(512), 0, 0, (const uint4*)d, o, KB * 1024 / 16, NREG); }, 50);
        printf("W=256 T=512 region %3d KB: %6.2f us per launch (back-to-back launches)\n", KB, t);
    }
    return 0;
}
