// C ABI glue: error string, version, and the fused-convolution entry point.
#include <stdarg.h>
#include <stdio.h>
#include "common.h"
#include "conv_params.h"
#include "../../include/dge_hip.h"

static thread_local char g_err[512] = "";
void dge_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* dge_last_error(void) { return g_err; }
static thread_local char g_kernel[160] = "";
void dge_note_kernel(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap); va_end(ap);
}
extern "C" const char* dge_last_kernel(void) { return g_kernel; }
extern "C" int dge_version(void) { return 100; }

// ---- deterministic mode: the per-translation-unit copies of the device-side state are updated through their registered setters
#include <vector>
static std::vector<void (*)(const DgeDet*)>& det_setters() { static std::vector<void (*)(const DgeDet*)> v; return v; }
void dge_det_register(void (*setter)(const DgeDet*)) { det_setters().push_back(setter); }
static DgeDet g_det_host = {0, nullptr, nullptr, 0, 0};
int dge_det_upload_failed = 0;
bool dge_det_fits(long long ndomains, long long nslots, long long L) {
    return !g_det_host.enabled || (ndomains * nslots * L <= g_det_host.ws_floats && ndomains <= (long long)g_det_host.ncounters);
}
extern "C" int dge_set_deterministic(int on) {
    if (on && !g_det_host.ws) {
        const long long nf = 32LL << 20;                 // 32 Mi floats = 128 MiB (largest need measured: 4.7 Mi, the weight-gradient slabs)
        const int nc = 1 << 16;
        float* ws = nullptr; unsigned* ct = nullptr;
        DGE_CHECK(hipMalloc((void**)&ws, (size_t)nf * 4) == hipSuccess, "set_deterministic: workspace allocation failed");
        DGE_CHECK(hipMalloc((void**)&ct, (size_t)nc * 4) == hipSuccess, "set_deterministic: counter allocation failed");
        DGE_CHECK(hipMemset(ct, 0, (size_t)nc * 4) == hipSuccess, "set_deterministic: counter reset failed");
        g_det_host.ws = ws; g_det_host.counters = ct; g_det_host.ws_floats = nf; g_det_host.ncounters = nc;
    }
    g_det_host.enabled = on ? 1 : 0;
    DGE_CHECK(hipDeviceSynchronize() == hipSuccess, "set_deterministic: device synchronisation failed");   // no launch straddles the switch
    dge_det_upload_failed = 0;
    for (auto f : det_setters()) f(&g_det_host);
    DGE_CHECK(hipDeviceSynchronize() == hipSuccess && !dge_det_upload_failed, "set_deterministic: state upload failed");
    return 0;
}
extern "C" int dge_get_deterministic(void) { return g_det_host.enabled; }

static DgeEnv g_env;
static bool g_env_loaded = false;
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static void env_load() {
    g_env.force_stream = getenv("DGE_FORCE_STREAM") ? 1 : 0;
    g_env.no_stream = getenv("DGE_NO_STREAM") ? 1 : 0;
    g_env.no_pw = getenv("DGE_NO_PW") ? 1 : 0;
    g_env.stream_nseg = env_int("DGE_STREAM_NSEG", 0);
    g_env.conv_dbg = env_int("DGE_CONV_DBG", 0);
    g_env.conv_bn = env_int("DGE_CONV_BN", 0);
    g_env.conv_kc = env_int("DGE_CONV_KC", 0);
    g_env.conv_small = env_int("DGE_CONV_SMALL", -1);
    g_env.conv_nok4 = getenv("DGE_CONV_NOK4") ? 1 : 0;
    g_env.conv_nok2 = getenv("DGE_CONV_NOK2") ? 1 : 0;
    g_env.torgb_thread = getenv("DGE_TORGB_THREAD") ? 1 : 0;
    g_env.wgrad_th8 = getenv("DGE_WGRAD_TH8") ? 1 : 0;
    g_env.wgrad_groups = env_int("DGE_WGRAD_GROUPS", 0);
    if (g_env.wgrad_groups < 0) g_env.wgrad_groups = 0;
    g_env.up_dbg = env_int("DGE_UP_DBG", 0);
    { const char* v = getenv("DGE_UP_VARIANT"); g_env.up_variant = !v ? 0 : (v[0] == 's' ? 2 : 1); }
    g_env_loaded = true;
}
// (first use from several host threads - autograd's backward thread next to the caller's - loads the switches exactly once)
#include <mutex>
static std::once_flag g_env_once;
const DgeEnv& dge_env() { std::call_once(g_env_once, [] { if (!g_env_loaded) env_load(); }); return g_env; }
extern "C" void dge_env_reload(void) { env_load(); }

extern "C" int dge_conv2d(const dge_conv_desc* d, hipStream_t s) {
    DGE_CHECK(d && d->x && d->w_packed && d->y, "conv2d: null tensor");
    DGE_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv2d: bad shape");
    DGE_CHECK(d->dtype == DGE_F32 || d->dtype == DGE_BF16, "conv2d: bad dtype %d", d->dtype);
    DGE_CHECK(!(d->up && d->stats), "conv2d: stats are not available in up mode");
    DGE_CHECK(!(d->up && d->in_s2d), "conv2d: up and in_s2d are exclusive");
    DGE_CHECK(!d->in_up2 || (!d->in_s2d && d->H % 2 == 0 && d->W % 2 == 0), "conv2d: in_up2 needs even H, W and no in_s2d");
    DGE_CHECK(!d->in_s2d || d->Cin % 4 == 0, "conv2d: in_s2d needs Cin %% 4 == 0");
    DGE_CHECK(!d->dot_src || d->stats || d->mask_relu || d->in_bwd_coef, "conv2d: dot_src needs a stats buffer");
    DGE_CHECK(!d->mask_relu || (d->dot_src && !d->prep && !d->up), "conv2d: mask_relu needs dot_src and excludes prep / up");
    DGE_CHECK(!d->in_relu || d->in_scale || d->in_shift, "conv2d: in_relu is applied together with the prologue affine");
    DGE_CHECK(d->gain > 0.f, "conv2d: gain must be positive (it is folded through the activation)");
    DGE_CHECK((((uintptr_t)d->in_scale | (uintptr_t)d->in_shift) & 15) == 0, "conv2d: in_scale/in_shift must be 16-byte aligned");
    DGE_CHECK(!d->noise || d->noise_w, "conv2d: noise without noise_w");
    ConvParams p;
    p.x = d->x; p.w = d->w_packed; p.y = d->y; p.addend = d->addend; p.dot_src = d->dot_src;
    p.in_scale = d->in_scale; p.in_shift = d->in_shift; p.out_scale = d->out_scale;
    p.bias = d->bias; p.noise = d->noise; p.noise_w = d->noise_w; p.stats = d->stats;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
    p.Ntot_valid = d->up ? 4 * d->Cout : d->Cout;
    p.Ntot = dge_packed_n(p.Ntot_valid);
    p.up = d->up; p.in_s2d = d->in_s2d; p.in_up2 = d->in_up2; p.in_relu = d->in_relu;
    const int OH = d->up ? 2 * d->H : d->H, OW = d->up ? 2 * d->W : d->W;
    p.noise_bstride = d->noise_batch > 1 ? OH * OW : 0;
    p.noise_w_stride = d->noise_w_per_channel ? 1 : 0;
    p.act = d->act; p.bias_scale = d->bias_scale; p.gain = d->gain; p.add_scale = d->add_scale;
    p.tiles_x = p.tiles_y = 0;
    p.w_frag = d->w_layout == 1 ? 1 : 0;
    p.prep = d->prep ? 1 : 0;
    p.mask_relu = d->mask_relu ? 1 : 0;
    p.in_t2d = d->in_t2d ? 1 : 0;
    DGE_CHECK(!d->in_t2d || (d->ksize == 3 && !d->up && !d->in_s2d && !d->in_up2 && !d->in_scale && !d->in_shift && d->w_layout == 0),
              "conv2d: in_t2d is a plain 3x3 launch (no other read mode, no prologue affine, row-major weights)");
    p.prep_gain = d->prep_gain; p.prep_noise = d->prep_noise; p.prep_ns = d->prep_ns; p.prep_stats = d->prep_stats;
    p.prep_noise_bstride = d->prep_noise_batch > 1 ? OH * OW : 0;
    DGE_CHECK(!d->prep || (d->dot_src && d->prep_stats && d->prep_gain > 0.f && !d->up), "conv2d: prep needs dot_src, prep_stats, a positive prep_gain and no up mode");
    DGE_CHECK(!d->prep || !dge_get_deterministic(), "conv2d: the fused tail backward (prep) is not offered in deterministic mode; run dge_modconv_bwd_prep");
    DGE_CHECK(d->w_layout == 0 || d->w_layout == 1, "conv2d: bad w_layout %d", d->w_layout);
    p.stats_slots = d->stats_slots > 0 ? d->stats_slots : 1;
    p.rgb_w = d->rgb_out ? d->rgb_w : nullptr; p.rgb_style = d->rgb_style; p.rgb_bias = d->rgb_bias; p.rgb_out = d->rgb_out;
    p.rgb_wscale = d->rgb_wscale; p.rgb_skip_y = (d->rgb_out && d->rgb_skip_y) ? 1 : 0;
    p.pool_out = d->pool_out ? 1 : 0; p.pool_mask = d->pool_out ? (unsigned*)d->pool_mask : nullptr;
    const bool pf = d->w_layout == 1 && d->prefetch_w && d->prefetch_ntot >= 64 && d->prefetch_ntot % 64 == 0 &&
                    (d->prefetch_cin == 512 || d->prefetch_cin == 256);
    p.pf_w = pf ? d->prefetch_w : nullptr; p.pf_ntot = pf ? d->prefetch_ntot : 0; p.pf_cin = pf ? d->prefetch_cin : 0;
    p.in_coef = d->in_bwd_coef;
    p.in_extra = d->in_bwd_coef ? d->in_bwd_extra : nullptr; p.in_extra_scale = d->in_bwd_extra_scale;
    p.fr_img4 = d->in_bwd_coef ? d->fr_img4 : nullptr; p.fr_out = d->in_bwd_coef ? d->fr_out : nullptr;
    if (d->in_bwd_coef) {
        DGE_CHECK(!dge_get_deterministic(), "conv2d: in_bwd_coef is not offered in deterministic mode; run dge_in_bwd");
        DGE_CHECK(d->ksize == 3 && !d->up && dge_conv_stream_eligible(p, d->dtype, d->ksize), "conv2d: in_bwd_coef is offered where "
                  "dge_conv_in_bwd_supported() says so, with dot_src and prep_stats, without stats / prep");
    }
    if (d->pool_out)
        DGE_CHECK(!d->up && dge_conv_pool_ok(p, d->dtype, d->ksize), "conv2d: the pooled epilogue is offered where dge_conv_pool_supported() "
                  "says so (conv_2 of the first encoder blocks on the streaming kernel)");
    if (d->rgb_out) {
        DGE_CHECK(d->rgb_w && d->rgb_style && d->rgb_bias, "conv2d: rgb_out needs rgb_w, rgb_style, rgb_bias");
        DGE_CHECK(dge_conv_rgb_ok(p, d->dtype, d->ksize), "conv2d: the fused toRGB is offered where dge_conv_rgb_supported() says so "
                  "(bf16 3x3 32 -> 32 launches of the streaming kernel)");
    }
    return dge_conv_launch(p, d->dtype, d->ksize, s);
}

// 1 when an encoder-flavour launch (instance-norm affine, per-sample noise, bias, lrelu; no statistics) of this shape may store the
// 2x2 average pool of its result instead of the result (dge_conv_desc.pool_out)
extern "C" int dge_conv_pool_supported(int B, int H, int W, int Cin, int Cout, int ksize, int dtype) {
    ConvParams p = {};
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Ntot_valid = Cout; p.Ntot = dge_packed_n(Cout);
    p.in_shift = (const float*)16; p.noise_w_stride = 1;       // (the encoder flavour's work threshold)
    return dge_conv_pool_ok(p, dtype, ksize) ? 1 : 0;
}

// 1 when a data-gradient launch of this shape may apply the instance-norm backward of its output's layer input in its epilogue
// (dge_conv_desc.in_bwd_coef)
extern "C" int dge_conv_in_bwd_supported(int B, int H, int W, int Cin, int Cout, int ksize, int dtype) {
    ConvParams p = {};
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Ntot_valid = Cout; p.Ntot = dge_packed_n(Cout);
    p.dot_src = (const void*)16; p.prep_stats = (float*)16; p.in_coef = (const float*)16;
    return (ksize == 3 && !dge_get_deterministic() && dge_conv_stream_eligible(p, dtype, ksize)) ? 1 : 0;
}
// ... in its block-input form (no activation, pooled skip gradient added: dge_conv_desc.in_bwd_extra, no prep_stats)
extern "C" int dge_conv_in_bwd_x_supported(int B, int H, int W, int Cin, int Cout, int ksize, int dtype) {
    ConvParams p = {};
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Ntot_valid = Cout; p.Ntot = dge_packed_n(Cout);
    p.dot_src = (const void*)16; p.in_coef = (const float*)16;
    return (ksize == 3 && !dge_get_deterministic() && dge_conv_stream_eligible(p, dtype, ksize)) ? 1 : 0;
}
// ... and reduce the FromRGB parameter gradients from it instead of storing it (dge_conv_desc.fr_out)
extern "C" int dge_conv_in_bwd_fromrgb_supported(int B, int H, int W, int Cin, int Cout, int ksize, int dtype) {
    ConvParams p = {};
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Ntot_valid = Cout; p.Ntot = dge_packed_n(Cout);
    p.dot_src = (const void*)16; p.in_coef = (const float*)16; p.fr_img4 = (const float*)16; p.fr_out = (float*)16;
    return (ksize == 3 && !dge_get_deterministic() && dge_conv_stream_eligible(p, dtype, ksize)) ? 1 : 0;
}

// 1 when a plain generator-flavour launch (style scale, demodulation, shared noise plane, bias, activation) of this shape may carry
// the fused toRGB epilogue (dge_conv_desc.rgb_*)
extern "C" int dge_conv_rgb_supported(int B, int H, int W, int Cin, int Cout, int ksize, int dtype) {
    ConvParams p = {};
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.Ntot_valid = Cout; p.Ntot = dge_packed_n(Cout);
    return dge_conv_rgb_ok(p, dtype, ksize) ? 1 : 0;
}

// out[i] (+)= sum_s partial[s][i]   (combines the spread statistics copies of dge_conv2d)
__global__ void sum_slots_kernel(const float* __restrict__ partial, float* __restrict__ out, int nslot, int n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= nslot; k += 8) {                  // 8 independent loads in flight
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = partial[(size_t)(k + j) * n + i];
#pragma unroll
        for (int j = 0; j < 8; j++) s += v[j];
    }
    for (; k < nslot; k++) s += partial[(size_t)k * n + i];
    out[i] = accumulate ? out[i] + s : s;
}
// out[k*C + c] = sum_s partial[s][c*NS + k]: the slot sum with the result laid out planar, so that each of the NS reductions
// is a contiguous [C] vector (parameter gradients are handed on without a strided copy)
__global__ void sum_slots_planar_kernel(const float* __restrict__ partial, float* __restrict__ out, int nslot, int C, int NS) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * NS) return;
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= nslot; k += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = partial[(size_t)(k + j) * C * NS + i];
#pragma unroll
        for (int j = 0; j < 8; j++) s += v[j];
    }
    for (; k < nslot; k++) s += partial[(size_t)k * C * NS + i];
    out[(size_t)(i % NS) * C + i / NS] = s;
}
extern "C" int dge_sum_slots_planar(const float* partial, float* out, int nslot, int C, int NS, hipStream_t s) {
    DGE_CHECK(nslot >= 1 && C >= 1 && NS >= 1, "sum_slots_planar: bad sizes");
    hipLaunchKernelGGL(sum_slots_planar_kernel, dim3((C * NS + 255) / 256), dim3(256), 0, s, partial, out, nslot, C, NS);
    DGE_LAUNCH_CHECK("sum_slots_planar");
    return 0;
}
// several planar slot sums in one launch (the per-channel parameter-gradient reductions of an encoder backward: none of them is
// needed before the backward ends, so the per-layer launches are deferred and grouped); table passed by value
struct SumPlanarTable { dge_sum_planar_entry e[32]; };
__global__ void sum_slots_planar_multi_kernel(SumPlanarTable tab) {
    const dge_sum_planar_entry e = tab.e[blockIdx.y];
    const int n = e.C * e.NS;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= e.nslot; k += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = e.partial[(size_t)(k + j) * n + i];
#pragma unroll
            for (int j = 0; j < 8; j++) s += v[j];
        }
        for (; k < e.nslot; k++) s += e.partial[(size_t)k * n + i];
        e.out[(size_t)(i % e.NS) * e.C + i / e.NS] = s;
    }
}
extern "C" int dge_sum_slots_planar_multi(const dge_sum_planar_entry* entries, int n, hipStream_t s) {
    DGE_CHECK(entries && n >= 1 && n <= 32, "sum_slots_planar_multi: 1..32 entries");
    SumPlanarTable tab;
    int maxn = 1;
    for (int i = 0; i < n; i++) {
        const dge_sum_planar_entry& e = entries[i];
        DGE_CHECK(e.partial && e.out && e.nslot >= 1 && e.C >= 1 && e.NS >= 1, "sum_slots_planar_multi: bad entry %d", i);
        tab.e[i] = e;
        if (e.C * e.NS > maxn) maxn = e.C * e.NS;
    }
    hipLaunchKernelGGL(sum_slots_planar_multi_kernel, dim3((maxn + 255) / 256, n), dim3(256), 0, s, tab);
    DGE_LAUNCH_CHECK("sum_slots_planar_multi");
    return 0;
}
extern "C" int dge_sum_slots(const float* partial, float* out, int nslot, int n, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(sum_slots_kernel, dim3((n + 255) / 256), dim3(256), 0, s, partial, out, nslot, n, accumulate);
    DGE_LAUNCH_CHECK("sum_slots");
    return 0;
}
