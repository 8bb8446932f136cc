/* ORACLE (test infrastructure only) - driver of the sanitizer build (make -C oracle san-check): runs the C restatement over
 * exactly-sized heap buffers on the shapes the parity tests use plus the edge cases (1x1 and 3x3 kernels, up-sampling layers,
 * R = 2 / 4, a single channel, no noise, no demodulation), so that AddressSanitizer / UBSan see every index expression at its
 * extremes.  Exit code 0 = clean; the sanitizers abort otherwise. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

void orc_style(const float* w, const float* A, const float* bias, float* style, int B, int cin, int wdim);
void orc_modconv(const float* x, const float* weight, const float* style, const float* bias, const float* noise,
                 float noise_strength, float* y, int B, int cin, int cout, int R, int k, int up, int demodulate, int lrelu);

static float* fill(size_t n, unsigned* s) {
    float* p = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++) { *s = *s * 1664525u + 1013904223u; p[i] = ((*s >> 8) & 0xffff) / 32768.0f - 1.0f; }
    return p;
}

int main(void) {
    const int cases[][7] = {      /* B, cin, cout, R, k, up, noise */
        {2, 8, 8, 8, 3, 0, 1}, {2, 8, 4, 8, 3, 1, 1}, {1, 4, 3, 8, 1, 0, 0}, {2, 1, 1, 2, 3, 0, 1}, {1, 3, 5, 4, 3, 1, 0},
        {3, 5, 7, 6, 3, 0, 1}, {1, 2, 2, 2, 3, 1, 1}, {1, 6, 2, 16, 3, 1, 1},
    };
    unsigned seed = 12345u;
    double checksum = 0;
    for (unsigned c = 0; c < sizeof(cases) / sizeof(cases[0]); c++) {
        const int B = cases[c][0], cin = cases[c][1], cout = cases[c][2], R = cases[c][3], k = cases[c][4], up = cases[c][5];
        const int Rin = up ? R / 2 : R, wdim = 16;
        float* w = fill((size_t)B * wdim, &seed);
        float* A = fill((size_t)cin * wdim, &seed);
        float* sb = fill(cin, &seed);
        float* style = (float*)malloc(sizeof(float) * B * cin);
        orc_style(w, A, sb, style, B, cin, wdim);
        float* x = fill((size_t)B * cin * Rin * Rin, &seed);
        float* wt = fill((size_t)cout * cin * k * k, &seed);
        float* bias = fill(cout, &seed);
        float* noise = cases[c][6] ? fill((size_t)R * R, &seed) : NULL;
        float* y = (float*)malloc(sizeof(float) * B * cout * R * R);
        for (int demod = 0; demod < 2; demod++)
            for (int lrelu = 0; lrelu < 2; lrelu++) {
                orc_modconv(x, wt, style, bias, noise, 0.3f, y, B, cin, cout, R, k, up, demod, lrelu);
                for (size_t i = 0; i < (size_t)B * cout * R * R; i++) {
                    if (!isfinite(y[i])) { fprintf(stderr, "case %u: non-finite output\n", c); return 2; }
                    checksum += y[i];
                }
            }
        free(w); free(A); free(sb); free(style); free(x); free(wt); free(bias); free(noise); free(y);
    }
    printf("san_driver: %u cases clean, checksum %.6f\n", (unsigned)(sizeof(cases) / sizeof(cases[0])), checksum);
    return 0;
}
