// Counter-based normal noise for the data-parallel training step.
//
// The reference draws its encoder / generator noise with torch.randn (model/E/E.py:60,73; stylegan2_generator.py:911-913,
// :187 new_z; model/stylegan1/net.py noise layers).  Under data parallelism every rank must draw the rows of ITS samples out
// of the noise a single process would draw for the global batch, otherwise N ranks x B images do not reproduce one process at
// batch N*B.  A stateful generator cannot be sliced; a counter-based one can: element j of draw `subseq` under `seed` is a pure
// function of (seed, subseq, j), so a rank generates exactly the elements [goff, goff + count) of the global tensor.
//
// Generator: Philox4x32-10 (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11; the same
// algorithm curand / torch use), key = seed (lo, hi), counter = (quad index lo, hi, subseq, 0); the four 32-bit outputs of a
// counter give four normals by Box-Muller: u = (x + 0.5) * 2^-32, (n0, n1) = sqrt(-2 ln u0) * (cos, sin)(2 pi u1), likewise (n2, n3).
// CPU restatement: oracle/philox_ref.py (pinned on the Random123 known-answer vectors).
#include "common.h"
#include "../../include/dge_hip.h"

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
}
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u0 = ((float)a + 0.5f) * 2.3283064365386963e-10f;          // (0, 1]: (2^32 - 1 + .5) * 2^-32 rounds to 1, ln = 0
    const float u1 = ((float)b + 0.5f) * 2.3283064365386963e-10f;
    const float r = sqrtf(-2.f * logf(u0));
    float s, c;
    sincospif(2.f * u1, &s, &c);
    n0 = r * c; n1 = r * s;
}

struct RandSeg { long long start, count; unsigned long long goff; unsigned int subseq, pad; };
#define DGE_RAND_MAX_SEG 40
struct RandArgs { RandSeg seg[DGE_RAND_MAX_SEG]; };

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, RandArgs a, unsigned long long seed,
                                                    const unsigned long long* __restrict__ seed_dev) {
    const RandSeg sg = a.seg[blockIdx.y];
    if (seed_dev) seed = seed_dev[0];
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    // quads are aligned in the GLOBAL element index: the first / last quad of a segment may be partial
    const unsigned long long q0 = sg.goff >> 2, q1 = (sg.goff + (unsigned long long)sg.count + 3) >> 2;
    float* __restrict__ o = out + sg.start - (long long)sg.goff;             // o[global index] (only in-segment elements are touched)
    for (unsigned long long q = q0 + blockIdx.x * 256ull + threadIdx.x; q < q1; q += gridDim.x * 256ull) {
        uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), sg.subseq, 0u};
        philox4x32_10(c, k0, k1);
        float n[4];
        box_muller(c[0], c[1], n[0], n[1]);
        box_muller(c[2], c[3], n[2], n[3]);
        const unsigned long long e = q << 2;
        if (e >= sg.goff && e + 4 <= sg.goff + (unsigned long long)sg.count && ((((uintptr_t)(o + e)) & 15) == 0)) {
            *(float4*)(o + e) = make_float4(n[0], n[1], n[2], n[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (e + k >= sg.goff && e + k < sg.goff + (unsigned long long)sg.count) o[e + k] = n[k];
        }
    }
}

extern "C" int dge_randn(float* out, int nseg, const long long* start, const long long* count, const unsigned long long* goff,
                         const unsigned int* subseq, unsigned long long seed, const unsigned long long* seed_dev, hipStream_t s) {
    DGE_CHECK(out && nseg >= 1, "randn: bad arguments");
    for (int base = 0; base < nseg; base += DGE_RAND_MAX_SEG) {
        RandArgs a;
        const int cnt = nseg - base < DGE_RAND_MAX_SEG ? nseg - base : DGE_RAND_MAX_SEG;
        long long maxq = 1;
        for (int i = 0; i < cnt; i++) {
            DGE_CHECK(count[base + i] >= 0 && start[base + i] >= 0, "randn: negative segment");
            a.seg[i].start = start[base + i]; a.seg[i].count = count[base + i]; a.seg[i].goff = goff[base + i];
            a.seg[i].subseq = subseq[base + i]; a.seg[i].pad = 0;
            const long long nq = (count[base + i] + 3) / 4 + 1;
            if (nq > maxq) maxq = nq;
        }
        long long gx = (maxq + 255) / 256; if (gx > 2048) gx = 2048;
        hipLaunchKernelGGL(randn_kernel, dim3((unsigned)gx, cnt), dim3(256), 0, s, out, a, seed, seed_dev);
        DGE_LAUNCH_CHECK("randn");
    }
    return 0;
}
