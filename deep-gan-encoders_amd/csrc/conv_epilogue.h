// Epilogue of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_small.hip): the accumulator tiles of a workgroup
// (MFMA 32x32 layout: lane = output channel, registers = 16 pixels) leave through the fused per-element stages and an f32 LDS
// transpose as 16-byte channel vectors.  Shared so that every kernel of the family offers the same menu of fused stages.
//
// Reference math of the stages: model/stylegan2_generator.py:908-921 (demodulation, noise, bias, lrelu*sqrt(2)),
// model/E/E.py:60-62,73-74 (noise weight, bias, lrelu), :77-84 (residual join) and their adjoints.
#pragma once
#include "common.h"
#include "conv_params.h"

// C: tile constants (MT, NT, WTM, WTN, ESTR, BM); TH x TW pixel tile, BN channels per workgroup, WM x WN waves.
// `lds`: base of the workgroup's LDS (staging area of wave w at lds + w*32*C::ESTR; all main-loop reads are done);
// `ldsN`: the noise tile staged by the prologue.  Only waves 0 .. WM*WN-1 carry tiles (`carrier`); in deterministic mode
// every thread of the workgroup must call.
// MODE (compile time; the registers of a stage cost every instantiation that carries it a wave of occupancy, so launches
// that never use a stage get a kernel built without it):
//   0  no addend / dot_src (the forward convolutions)
//   1  addend and / or dot_src, requested one tile ahead of their use
//   2  mode 1 + the fused tail backward of ConvParams::prep (16 more per-lane sums)
template <typename T, class C, int TH, int TW, int BN, int WM, int WN, int NTHREADS, int MODE = 0>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16_t (&acc)[C::MT][C::NT], unsigned char* lds, const float* ldsN,
                                              int b, int x0, int y0, int bn0, int ntile, int vbid, int tx_i, int ty_i,
                                              int wave, int lane, int tid, bool carrier) {
    constexpr int EP16 = Elem<T>::PER16;
    constexpr bool PREP = MODE == 2;
    const int wm = wave / WN, wn = wave % WN;
    // NOTE: every accumulator index below is a compile-time constant (StaticFor): a run-time index
    // into acc[][] would push the whole accumulator file to scratch on every main-loop iteration.
    unsigned char* est = lds + wave * (32 * C::ESTR);     // (the noise tile at the end of `lds` stays valid)
    const int OH = p.up ? 2 * p.H : p.H, OW = p.up ? 2 * p.W : p.W;
    T* __restrict__ Y = (T*)p.y;
    const T* __restrict__ ADD = MODE >= 1 ? (const T*)p.addend : nullptr;      // (the launcher picks MODE from these pointers)
    const T* __restrict__ DOT = MODE >= 1 ? (const T*)p.dot_src : nullptr;
    constexpr int CPR = 32 / EP16;              // 16-byte chunks per 32-channel row
    constexpr int PPP = 64 / CPR;               // pixels per read-back pass
    // Two kinds of epilogue work:
    //   "pre"  (per accumulator element, this lane owns ONE channel): demodulation scale, noise,
    //          bias, activation, gain, plain statistics;
    //   "post" (after the LDS transpose, this lane owns 8/4 consecutive channels of one pixel, so
    //          every global access is a 16-byte vector): residual addend, the data-gradient dot
    //          products with dot_src, statistics of results that include the addend.
    // In DOT mode (data gradients) the per-channel scale is applied post, on the staged raw value.
    const bool post_stats = p.stats && (DOT || ADD);
    float* __restrict__ STATS = p.stats ? p.stats + (size_t)(vbid % p.stats_slots) * p.B * p.Cout * 2 : nullptr;
    // deterministic mode: domain = (sample, N tile), slot = (pixel tile, wave row wm), vector = (sum, sum2) per channel of the N tile;
    // the last workgroup of the domain sums the slots in order into copy 0 of the statistics buffer (the other copies stay zero)
    const bool det = p.stats && !p.up && det_on();
    const int det_ntn = (p.Ntot + BN - 1) / BN, det_dom = b * det_ntn + ntile, det_nslots = p.tiles_x * p.tiles_y * WM;
    float* det_vec = det ? det_slot(det_dom, p.B * det_ntn, (ty_i * p.tiles_x + tx_i) * WM + wm, det_nslots, BN * 2) : nullptr;
    // activation as max(v, slope*v) (slope in [0,1]); the gain (> 0, checked by the launcher) is folded
    // into scale / noise weight / bias because every supported activation is positively homogeneous
    const float slope = p.act == DGE_ACT_LRELU ? 0.2f : (p.act == DGE_ACT_RELU ? 0.f : 1.f);
    const bool tile_full = (y0 + TH <= p.H) && (x0 + TW <= p.W);
    const int lh = lane >> 5, l31 = lane & 31;
    T* __restrict__ Yb = Y + (size_t)b * OH * OW * p.Cout;           // in-image offsets fit 32 bits
    const T* __restrict__ ADDb = ADD ? ADD + (size_t)b * OH * OW * p.Cout : nullptr;
    const T* __restrict__ DOTb = DOT ? DOT + (size_t)b * OH * OW * p.Cout : nullptr;
    // The addend / dot_src vectors of a tile are requested ONE TILE AHEAD of their use (register double buffer): requested inside
    // the read-back loop, every pass exposed a full memory latency - the data-gradient launches of the 512^2 / 1024^2 layers
    // spent half their time there (tools/perf_s2d.py: 692 us with, 353 us without the dot / addend stages on layer 15).
    constexpr int NPASS = 32 / PPP;
    // (f32 tiles take 4 passes: 64 staging registers; the 8-tile wave of the 128-wide configuration has none to spare next to
    //  the prep sums: loads stay in place there)
    constexpr bool PF = MODE >= 1 && NPASS <= 2 && !(MODE == 2 && C::MT * C::NT >= 8);
    const bool pf = PF && !p.up && (DOT || ADD);
    uint4 pfd[2][PF ? NPASS : 1], pfa[2][PF ? NPASS : 1];
    auto pf_issue = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int jj = t / C::MT, ii = t % C::MT;
        const int oc_ = bn0 + wn * C::WTN + jj * 32 + (lane % CPR) * EP16;
#pragma unroll
        for (int q = 0; q < (PF ? NPASS : 0); q++) {
            const int m = wm * C::WTM + ii * 32 + q * PPP + lane / CPR;
            const int gy = y0 + m / TW, gx = x0 + m % TW;
            if ((gy < p.H) & (gx < p.W) & (oc_ < p.Cout)) {
                const int off = (gy * OW + gx) * p.Cout + oc_;
                if (DOT) pfd[t & 1][q] = *(const uint4*)(DOTb + off);
                if (ADD) pfa[t & 1][q] = *(const uint4*)(ADDb + off);
            }
        }
    };
    if (pf && carrier && !(p.dbg & 4)) pf_issue(std::integral_constant<int, 0>{});
    if (!(p.dbg & 4) && carrier)
    StaticFor<C::NT>::run([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int n0 = bn0 + wn * C::WTN + j * 32;          // first N of this 32-wide tile
        if (n0 >= p.Ntot_valid) return;                     // wave-uniform
        // up mode: N = (phase, channel).  A 32-wide tile lies inside one phase when Cout % 32 == 0 (tile-uniform
        // phase); with Cout == 16 (StyleGAN1 FFHQ-1024 top block) it spans two, so phase/channel are per lane.
        const bool split = p.up && (p.Cout & 31);
        int phase = p.up ? n0 / p.Cout : 0;                 // "pre" side: this lane's channel n0 + l31
        int o = (p.up ? n0 % p.Cout : n0) + l31;
        const int chq = lane % CPR;
        int phase_c = phase;                                // "post" side: this lane's chunk n0 + chq*EP16 ..
        int oc = (p.up ? n0 % p.Cout : n0) + chq * EP16;
        bool ovalid = o < p.Cout, cvalid = oc < p.Cout;
        if (split) {
            const int n = n0 + l31, nc = n0 + chq * EP16;
            phase = n / p.Cout; o = n - phase * p.Cout; ovalid = n < p.Ntot_valid;
            phase_c = nc / p.Cout; oc = nc - phase_c * p.Cout; cvalid = nc < p.Ntot_valid;
            if (!ovalid) phase = 0;
        }
        const int py = phase_c >> 1, px = phase_c & 1;
        const float osc = ((p.out_scale && ovalid && !DOT) ? p.out_scale[b * p.Cout + o] : 1.f) * p.gain;
        const float bia = (p.bias && ovalid) ? p.bias[o] * p.bias_scale * p.gain : 0.f;
        const float nw = (p.noise && ovalid) ? p.noise_w[o * p.noise_w_stride] * p.gain : 0.f;
        float ssum = 0.f, ssq = 0.f;
        float posc[EP16], ps0[EP16], ps1[EP16], pt0[PREP ? EP16 : 1], pt1[PREP ? EP16 : 1];
#pragma unroll
        for (int e = 0; e < EP16; e++) {
            posc[e] = (DOT && p.out_scale && cvalid) ? p.out_scale[b * p.Cout + oc + e] : 1.f;
            ps0[e] = 0.f; ps1[e] = 0.f;
            if constexpr (PREP) { pt0[e] = 0.f; pt1[e] = 0.f; }
        }
        // fused tail backward of the layer below (ConvParams::prep): g -> g_z and the two sums of its demodulation gradient
        const bool prep = PREP && p.prep && DOT;
        const float pg_pos = p.prep_gain, pg_neg = 0.2f * p.prep_gain, pz_pos = 1.f / p.prep_gain, pz_neg = 1.f / (0.2f * p.prep_gain);
        const float pns = (prep && p.prep_noise && p.prep_ns) ? p.prep_ns[0] : 0.f;
        const bool pnz = prep && p.prep_noise;                       // its plane sits in the noise tile (staged by the kernel's prologue)
        float* estw = (float*)(est + 4 * lh * C::ESTR) + l31;                     // + ((r&3) + 8(r>>2)) rows
        const float* nzb = ldsN + phase * C::BM + wm * C::WTM + 4 * lh;          // + i*32 + 8(r>>2) + (r&3)
        StaticFor<C::MT>::run([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int tcur = j * C::MT + i;
            if constexpr (PF && tcur + 1 < C::MT * C::NT) { if (pf) pf_issue(std::integral_constant<int, tcur + 1>{}); }
            const f32x16_t a = acc[i][j];
            float nz[16], val[16];
            if (p.noise) {
#pragma unroll
                for (int q = 0; q < 4; q++) *(float4*)&nz[q * 4] = *(const float4*)(nzb + i * 32 + q * 8);
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) nz[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float v = fmaf(a[r], osc, fmaf(nw, nz[r], bia));
                val[r] = fmaxf(v, v * slope);
            }
            if (p.stats && !post_stats) {
                if (tile_full) {
#pragma unroll
                    for (int r = 0; r < 16; r++) { ssum += val[r]; ssq += val[r] * val[r]; }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int m = wm * C::WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const bool pv = (y0 + m / TW < p.H) & (x0 + m % TW < p.W);
                        ssum += pv ? val[r] : 0.f; ssq += pv ? val[r] * val[r] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) estw[((r & 3) + 8 * (r >> 2)) * (C::ESTR / 4)] = val[r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 32 / PPP; q++) {
                const int ml = q * PPP + lane / CPR;
                const int m = wm * C::WTM + i * 32 + ml;
                const int gy = y0 + m / TW, gx = x0 + m % TW;
                const int oy = p.up ? 2 * gy + py : gy, ox = p.up ? 2 * gx + px : gx;
                if ((gy < p.H) & (gx < p.W) & cvalid) {
                    float f[EP16];
#pragma unroll
                    for (int e4 = 0; e4 < EP16 / 4; e4++)
                        *(uint4*)&f[e4 * 4] = *(const uint4*)(est + ml * C::ESTR + chq * EP16 * 4 + e4 * 16);
                    const int off = (oy * OW + ox) * p.Cout + oc;
                    if (DOT || ADD) {
                        float d[EP16];
                        if (DOT) {
                            uint4 dv;
                            if constexpr (PF) dv = pf ? pfd[tcur & 1][q] : *(const uint4*)(DOTb + off);
                            else dv = *(const uint4*)(DOTb + off);
                            unpack16(dv, d, (T*)nullptr);
#pragma unroll
                            for (int e = 0; e < EP16; e++) { ps0[e] += f[e] * d[e]; ps1[e] += f[e]; f[e] *= posc[e]; }
                        }
                        if (ADD) {
                            float ad[EP16];
                            uint4 av;
                            if constexpr (PF) av = pf ? pfa[tcur & 1][q] : *(const uint4*)(ADDb + off);
                            else av = *(const uint4*)(ADDb + off);
                            unpack16(av, ad, (T*)nullptr);
#pragma unroll
                            for (int e = 0; e < EP16; e++) f[e] += p.add_scale * ad[e];
                        }
                        if (DOT && p.mask_relu) {          // ReLU backward of the layer below (ConvParams::mask_relu)
#pragma unroll
                            for (int e = 0; e < EP16; e++) f[e] = d[e] > 0.f ? f[e] : 0.f;
                        }
                        if constexpr (PREP) if (prep) {
                            const float nzs = pnz ? pns * ldsN[m] : 0.f;
#pragma unroll
                            for (int e = 0; e < EP16; e++) {
                                const bool pos = d[e] > 0.f;
                                const float gz = f[e] * (pos ? pg_pos : pg_neg);
                                const float zt = d[e] * (pos ? pz_pos : pz_neg) - nzs;
                                pt0[e] = fmaf(gz, zt, pt0[e]); pt1[e] += gz;
                                f[e] = gz;
                            }
                        }
                        if (post_stats && !DOT) {
#pragma unroll
                            for (int e = 0; e < EP16; e++) { ps0[e] += f[e]; ps1[e] += f[e] * f[e]; }
                        }
                    }
                    *(uint4*)(Yb + off) = pack16(f, (T*)nullptr);
                }
            }
            __builtin_amdgcn_wave_barrier();
        });
        if (p.stats && !post_stats) {
            ssum += __shfl_xor(ssum, 32, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (lane < 32 && ovalid) {
                if (det) {                 // this wave's slot of the (sample, N tile) domain: one entry pair per channel of the tile
                    const int e = (wn * C::WTN + j * 32 + l31) * 2;
                    det_vec[e] = ssum; det_vec[e + 1] = ssq;
                } else {
                    atomicAdd(STATS + ((size_t)b * p.Cout + o) * 2, ssum);
                    atomicAdd(STATS + ((size_t)b * p.Cout + o) * 2 + 1, ssq);
                }
            }
        }
        // Per-channel sums of the "post" side: after the butterflies every lane with the same chunk index holds the totals of its
        // EP16 channels.  Lane (chq, e' = lane / CPR) keeps channel oc + e', so that ONE atomic instruction carries the 32
        // channels of the tile (an instruction per channel with CPR active lanes is an L2 request each; at ~10 requests per ns on
        // the whole device the 16 - 32 instructions per tile cost the data-gradient launches 25 - 100 us)
        const int esel = (lane / CPR) % EP16;
        auto lane_pick = [&](float (&v)[EP16]) {
            float out = 0.f;
#pragma unroll
            for (int e = 0; e < EP16; e++) {
#pragma unroll
                for (int msk = CPR; msk < 64; msk <<= 1) v[e] += __shfl_xor(v[e], msk, 64);
                if (esel == e) out = v[e];
            }
            return out;
        };
        if constexpr (PREP) if (prep && p.prep_stats) {
            float* __restrict__ PST = p.prep_stats + (size_t)(vbid % p.stats_slots) * p.B * p.Cout * 2;
            const float t0 = lane_pick(pt0), t1 = lane_pick(pt1);
            if (lane < CPR * EP16 && cvalid) {
                atomicAdd(PST + ((size_t)b * p.Cout + oc + esel) * 2, t0);
                atomicAdd(PST + ((size_t)b * p.Cout + oc + esel) * 2 + 1, t1);
            }
        }
        if (post_stats) {          // lanes with equal chq hold partial sums of the same channels
            if (det) {
#pragma unroll
                for (int e = 0; e < EP16; e++) {
#pragma unroll
                    for (int msk = CPR; msk < 64; msk <<= 1) { ps0[e] += __shfl_xor(ps0[e], msk, 64); ps1[e] += __shfl_xor(ps1[e], msk, 64); }
                    if (lane < CPR && cvalid) {
                        const int ee = (wn * C::WTN + j * 32 + chq * EP16 + e) * 2;
                        det_vec[ee] = ps0[e]; det_vec[ee + 1] = ps1[e];
                    }
                }
            } else {
                const float t0 = lane_pick(ps0), t1 = lane_pick(ps1);
                if (lane < CPR * EP16 && cvalid) {
                    atomicAdd(STATS + ((size_t)b * p.Cout + oc + esel) * 2, t0);
                    atomicAdd(STATS + ((size_t)b * p.Cout + oc + esel) * 2 + 1, t1);
                }
            }
        }
    });
    if (det) {
        if (det_arrive_wg(det_dom, p.tiles_x * p.tiles_y)) {
            for (int idx = tid; idx < BN * 2; idx += NTHREADS) {
                const int o = bn0 + (idx >> 1);
                if (o < p.Cout) p.stats[((size_t)b * p.Cout + o) * 2 + (idx & 1)] = det_sum(det_dom, det_nslots, BN * 2, idx);
            }
        }
    }
}
