// Probe (dev tool): semantics of LDS-DMA on gfx950 that csrc/conv_stream.hip relies on.
//   1. global_load_lds_dwordx4 under a partial EXEC mask: inactive lanes must leave their LDS bytes untouched
//   2. buffer_load_dwordx4 ... offen lds with out-of-range offsets: do the lanes write zeros?
//   3. the same under a partial EXEC mask
// hipcc --offload-arch=gfx950 -O2 probe_ldsdma.hip -o probe_ldsdma && ./probe_ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ unsigned lds_off(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}

__global__ void probe(const uint32_t* src, uint32_t* out, int nbytes) {
    __shared__ __attribute__((aligned(1024))) uint32_t lds[3 * 256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 3 * 256; i += 64) lds[i] = 0xAAAAAAAAu;
    __syncthreads();
    const unsigned base = lds_off(lds);
    // 1: global_load_lds with exec = low 40 lanes
    {
        const uint32_t* g = src + lane * 4;
        unsigned long long mask = (1ull << 40) - 1;
        asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 exec, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_mov_b64 exec, -1"
                     :: "v"(g), "s"(base), "s"(mask) : "memory");
    }
    // 2: buffer_load lds, offsets: lane*16 - 64 (first 4 lanes negative), range = nbytes
    {
        __attribute__((ext_vector_type(4))) unsigned rs;
        const unsigned long long a = (unsigned long long)src;
        rs[0] = (unsigned)a; rs[1] = (unsigned)(a >> 32); rs[2] = (unsigned)nbytes; rs[3] = 0x00020000u;
        rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
        rs[2] = __builtin_amdgcn_readfirstlane(rs[2]); rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);
        int voff = lane * 16 - 64;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                     :: "v"(voff), "s"(base + 1024), "s"(rs) : "memory");
        unsigned long long mask = 0x00ffff00ffff00ffull;
        asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 exec, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen offset:16 lds\n\ts_mov_b64 exec, -1"
                     :: "v"(voff), "s"(base + 2048), "s"(rs), "s"(mask) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 3 * 256; i += 64) out[i] = lds[i];
}

int main() {
    const int N = 4096;
    std::vector<uint32_t> h(N);
    for (int i = 0; i < N; i++) h[i] = 0x1000000u + i;
    uint32_t *d, *o;
    hipMalloc(&d, N * 4); hipMalloc(&o, 3 * 256 * 4);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    const int nbytes = 800;     // lanes with offset >= 800 (lane*16-64 >= 800 -> lane >= 54) are out of range
    probe<<<1, 64>>>(d, o, nbytes);
    std::vector<uint32_t> r(3 * 256);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int k = 0; k < 4; k++) {
        uint32_t want = l < 40 ? 0x1000000u + l * 4 + k : 0xAAAAAAAAu;
        if (r[l * 4 + k] != want) { if (bad < 4) printf("test1 lane %d dword %d: %08x want %08x\n", l, k, r[l * 4 + k], want); bad++; }
    }
    printf("test1 (global_load_lds under EXEC mask): %s\n", bad ? "FAIL" : "ok");
    bad = 0;
    for (int l = 0; l < 64; l++) for (int k = 0; k < 4; k++) {
        int off = l * 16 - 64 + k * 4;
        uint32_t want = (off >= 0 && off + 4 <= nbytes) ? 0x1000000u + off / 4 : 0u;
        if (r[256 + l * 4 + k] != want) { if (bad < 8) printf("test2 lane %d dword %d: %08x want %08x\n", l, k, r[256 + l * 4 + k], want); bad++; }
    }
    printf("test2 (buffer_load lds, out-of-range lanes write zeros): %s\n", bad ? "FAIL" : "ok");
    bad = 0;
    const unsigned long long mask = 0x00ffff00ffff00ffull;
    for (int l = 0; l < 64; l++) for (int k = 0; k < 4; k++) {
        int off = l * 16 - 64 + 16 + k * 4;
        uint32_t want = !((mask >> l) & 1) ? 0xAAAAAAAAu : ((off >= 0 && off + 4 <= nbytes) ? 0x1000000u + off / 4 : 0u);
        if (r[512 + l * 4 + k] != want) { if (bad < 8) printf("test3 lane %d dword %d: %08x want %08x\n", l, k, r[512 + l * 4 + k], want); bad++; }
    }
    printf("test3 (buffer_load lds under EXEC mask, imm offset): %s\n", bad ? "FAIL" : "ok");
    return 0;
}
