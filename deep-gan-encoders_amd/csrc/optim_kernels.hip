// LREQAdam (reference model/utils/custom_adam.py:24-76) as one multi-tensor launch:
//   v = beta2*v + (1-beta2)*g^2 ; p -= step_size * g / (sqrt(v) + eps)
// with the per-tensor step_size = lr*sqrt(1-beta2^t)*lr_equalization_coef computed by the host; optionally the
// step-count factor comes from a DEVICE scalar (step_mult) so that a captured hipGraph of the step can be replayed.
#include "common.h"
#include "../../include/dge_hip.h"

#define ADAM_MAX_TENSORS 48
struct AdamTable {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    long n[ADAM_MAX_TENSORS];
    float step[ADAM_MAX_TENSORS];
};

__global__ __launch_bounds__(256) void lreq_adam_kernel(AdamTable t, float beta2, float eps, const float* __restrict__ gscale,
                                                         const float* __restrict__ step_mult) {
    const int ti = blockIdx.y;
    float* __restrict__ p = t.p[ti]; const float* __restrict__ g = t.g[ti]; float* __restrict__ v = t.v[ti];
    const long n = t.n[ti]; const float step = t.step[ti] * (step_mult ? step_mult[0] : 1.f);
    const float gs = gscale ? gscale[0] : 1.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const float gr = g[i] * gs;
        const float vv = v[i] * beta2 + (1.f - beta2) * gr * gr;
        v[i] = vv;
        p[i] -= step * gr / (sqrtf(vv) + eps);
    }
}

extern "C" int dge_lreq_adam_multi(int ntensors, float* const* host_p, const float* const* host_g, float* const* host_v,
                                   const long* host_n, const float* host_step, float beta2, float eps, const float* gscale,
                                   const float* step_mult, hipStream_t s) {
    DGE_CHECK(ntensors >= 0, "adam: bad tensor count");
    for (int base = 0; base < ntensors; base += ADAM_MAX_TENSORS) {
        AdamTable t;
        const int cnt = ntensors - base < ADAM_MAX_TENSORS ? ntensors - base : ADAM_MAX_TENSORS;
        long maxn = 0;
        for (int i = 0; i < cnt; i++) {
            t.p[i] = host_p[base + i]; t.g[i] = host_g[base + i]; t.v[i] = host_v[base + i];
            t.n[i] = host_n[base + i]; t.step[i] = host_step[base + i];
            if (t.n[i] > maxn) maxn = t.n[i];
        }
        int gx = (int)((maxn + 255) / 256); if (gx > 256) gx = 256; if (gx < 1) gx = 1;
        hipLaunchKernelGGL(lreq_adam_kernel, dim3(gx, cnt), dim3(256), 0, s, t, beta2, eps, gscale, step_mult);
        DGE_LAUNCH_CHECK("lreq_adam_multi");
    }
    return 0;
}
