/* ORACLE (test infrastructure only; never linked into the product) - plain C restatement of the
 * north-star kernel: StyleGAN2 ModulateConvBlock.forward, reference
 * model/stylegan2_generator.py:855-922, in the reference's own *fused* formulation (per-sample
 * modulated + demodulated weights :858-875, conv2d :898-904 or conv_transpose2d stride 2 with the
 * flipped kernel :879-895 followed by the 4x4 FIR filter :896 / :603-615), NCHW fp32.
 * Pinned against tests/golden/s2_blocks.npz (outputs of the reference itself) by
 * tests/test_oracle_golden.py::test_c_oracle_modconv.  Build: make -C oracle
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* style[b][i] = (w[b] . A[i]) / sqrt(wdim) + bias[i] + 1      (DenseBlock :990-996, additional_bias 1) */
void orc_style(const float* w, const float* A, const float* bias, float* style, int B, int cin, int wdim) {
    const float sc = 1.0f / sqrtf((float)wdim);
    for (int b = 0; b < B; b++)
        for (int i = 0; i < cin; i++) {
            double s = 0;
            for (int k = 0; k < wdim; k++) s += (double)w[b * wdim + k] * A[i * wdim + k];
            style[b * cin + i] = (float)s * sc + bias[i] + 1.0f;
        }
}

/* y [B,cout,R,R]; x [B,cin,Rin,Rin] with Rin = up ? R/2 : R; weight [cout,cin,k,k]; noise [R*R] or NULL */
void orc_modconv(const float* x, const float* weight, const float* style, const float* bias, const float* noise,
                 float noise_strength, float* y, int B, int cin, int cout, int R, int k, int up, int demodulate,
                 int lrelu) {
    const int Rin = up ? R / 2 : R;
    const float wscale = 1.0f / sqrtf((float)(cin * k * k));
    const int kk = k * k;
    const float fir1[4] = {1.f, 3.f, 3.f, 1.f};
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < B; b++)
        for (int o = 0; o < cout; o++) {
            /* per-sample weights  W'[i][t] = W*wscale*style, demodulated (:858-870) */
            float* wm = (float*)malloc(sizeof(float) * cin * kk);
            double nrm = 0;
            for (int i = 0; i < cin; i++)
                for (int t = 0; t < kk; t++) {
                    const float v = weight[((size_t)o * cin + i) * kk + t] * wscale * style[b * cin + i];
                    wm[i * kk + t] = v; nrm += (double)v * v;
                }
            if (demodulate) {
                const float inv = 1.0f / sqrtf((float)nrm + 1e-8f);
                for (int j = 0; j < cin * kk; j++) wm[j] *= inv;
            }
            float* out = y + ((size_t)b * cout + o) * R * R;
            if (!up) {
                const int pad = k / 2;
                for (int p = 0; p < R; p++)
                    for (int q = 0; q < R; q++) {
                        double s = 0;
                        for (int i = 0; i < cin; i++) {
                            const float* xi = x + ((size_t)b * cin + i) * R * R;
                            for (int u = 0; u < k; u++) {
                                const int yy = p + u - pad; if (yy < 0 || yy >= R) continue;
                                for (int v = 0; v < k; v++) {
                                    const int xx = q + v - pad; if (xx < 0 || xx >= R) continue;
                                    s += (double)wm[i * kk + u * k + v] * xi[yy * R + xx];
                                }
                            }
                        }
                        out[p * R + q] = (float)s;
                    }
            } else {
                /* conv_transpose2d(stride 2, pad 0) with the spatially flipped kernel -> (2Rin+1)^2 */
                const int Z = 2 * Rin + 1;
                double* z = (double*)calloc((size_t)Z * Z, sizeof(double));
                for (int i = 0; i < cin; i++) {
                    const float* xi = x + ((size_t)b * cin + i) * Rin * Rin;
                    for (int p = 0; p < Rin; p++)
                        for (int q = 0; q < Rin; q++) {
                            const float xv = xi[p * Rin + q];
                            for (int u = 0; u < 3; u++)
                                for (int v = 0; v < 3; v++)
                                    z[(2 * p + u) * Z + 2 * q + v] += (double)xv * wm[i * 9 + (2 - u) * 3 + (2 - v)];
                        }
                }
                /* filter: pad 1 on every side, 4x4 FIR outer([1,3,3,1])/64 * 4 */
                for (int p = 0; p < R; p++)
                    for (int q = 0; q < R; q++) {
                        double s = 0;
                        for (int u = 0; u < 4; u++) {
                            const int yy = p + u - 1; if (yy < 0 || yy >= Z) continue;
                            for (int v = 0; v < 4; v++) {
                                const int xx = q + v - 1; if (xx < 0 || xx >= Z) continue;
                                s += z[yy * Z + xx] * fir1[u] * fir1[v];
                            }
                        }
                        out[p * R + q] = (float)(s / 16.0);
                    }
                free(z);
            }
            for (int j = 0; j < R * R; j++) {
                float v = out[j];
                if (noise) v += noise[j] * noise_strength;
                v += bias[o];
                if (lrelu) v = (v > 0 ? v : 0.2f * v) * 1.41421356237f;
                out[j] = v;
            }
            free(wm);
        }
}
