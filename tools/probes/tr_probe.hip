// Probe of ds_read_b64_tr_b16 lane/element mapping (dev tool).  hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const int* addr_in, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) uint16_t*)lds + addr_in[threadIdx.x];
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
    int h_addr[64]; uint16_t h_out[256];
    // lane i of each 16-lane group points at row (i>>2) [pitch 64 elements = 128 B], column segment (i&3)*4; groups at +16 columns / +8 rows
    for (int l = 0; l < 64; l++) {
        int g = l >> 4, i = l & 15;
        int row = (g >> 1) * 8 + (i >> 2), col = (g & 1) * 16 + (i & 3) * 4;
        h_addr[l] = (row * 64 + col) * 2;
    }
    int* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) {
        printf("lane %2d addr(row %2d col %2d):", l, h_addr[l] / 2 / 64, h_addr[l] / 2 % 64);
        for (int j = 0; j < 4; j++) printf("  (r%d,c%d)", h_out[l * 4 + j] / 64, h_out[l * 4 + j] % 64);
        printf("\n");
    }
    return 0;
}
