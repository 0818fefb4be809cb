/* capi_demo.c - the C ABI of include/adsp.h from plain C, no Python: the reference's CreateLowCutFilter(800) at 44.1 kHz on
 * 512-sample chunks (EffectFFTFilter.py:91-151), 3 channels, 8 chunks through adsp_apply_host, checked against the float64
 * direct convolution  out[tau] = sum_t h[t] s[tau - N + d - t]  (SURVEY.md section 0).
 *
 *   gcc -O2 -std=c11 -Iinclude examples/capi_demo.c -Lpyaudiodsptools_amd -ladsp -lm \
 *       -Wl,-rpath,$PWD/pyaudiodsptools_amd -Wl,-rpath-link,/opt/rocm/lib -o /tmp/capi_demo && /tmp/capi_demo
 *
 * Exit code 0 and "max |error| ... OK" on a machine with a GPU; exit code 2 with libadsp's message without one (there is
 * no CPU fallback). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "adsp.h"

#define N 512
#define CHANNELS 3
#define STEPS 8
static const double PI = 3.14159265358979323846;

static double sinc(double x) { return x == 0.0 ? 1.0 : sin(PI * x) / (PI * x); }

int main(void) {
    /* the reference's design: Blackman windowed sinc, L = N/2 - 1 taps, unity DC gain, spectral inversion */
    enum { L = N / 2 - 1, D = (L - 1) / 2, F = 2 * N };
    static double h[L];
    double sum = 0.0;
    for (int n = 0; n < L; ++n) {
        const double w = 0.42 - 0.5 * cos(2 * PI * n / (L - 1)) + 0.08 * cos(4 * PI * n / (L - 1));
        h[n] = sinc(2.0 * 800.0 / 44100.0 * (n - (L - 1) / 2.0)) * w;
        sum += h[n];
    }
    for (int n = 0; n < L; ++n) h[n] = -h[n] / sum;
    h[D] += 1.0;

    /* geometry of include/adsp.h for a cut filter: F = 2N, window starts 1.25 N before the block, kept slice at N/4;
     * the symmetric kernel is centred on circular index 0, so its spectrum is real (exact zeros select the cheaper stage) */
    adsp_config cfg = {0, N, CHANNELS, F, 2, N + N / 4, N / 4, 0, ADSP_FORMAT_F32};
    adsp_engine* eng = NULL;
    if (adsp_create(&cfg, &eng) != ADSP_OK) {
        fprintf(stderr, "adsp_create: %s\n", adsp_last_error());
        return 2;
    }
    static float spec[2 * (F / 2 + 1)];
    for (int k = 0; k <= F / 2; ++k) { /* H[k] = sum_t h[t] cos(2 pi k (t - D) / F): the kernel sits at circular indices t - D */
        double re = 0.0;
        for (int t = 0; t < L; ++t) re += h[t] * cos(2 * PI * k * (double)(t - D) / F);
        spec[2 * k] = (float)re;
        spec[2 * k + 1] = 0.0f;
    }
    if (adsp_set_spectrum(eng, spec, F / 2 + 1) != ADSP_OK || adsp_set_kernel_reach(eng, D) != ADSP_OK) {
        fprintf(stderr, "adsp_set_spectrum: %s\n", adsp_last_error());
        return 1;
    }
    /* the multi-GPU job's one collective, here over a world of one engine: libadsp opens librccl itself (ncclCommInitAll, no
     * rendezvous) and every engine rebuilds its tables from what the broadcast left in its memory.  No RCCL on the machine is
     * not an error of this demo: the filter set above simply stays in place */
    int rccl = 0;
    adsp_engine* world[1] = {eng};
    if (adsp_rccl_version(&rccl) == ADSP_OK) {
        if (adsp_bcast_spectrum(world, 1, 0) != ADSP_OK) {
            fprintf(stderr, "adsp_bcast_spectrum: %s\n", adsp_last_error());
            return 1;
        }
    }
    int is_real = 0;
    adsp_spectrum_is_real(eng, &is_real);

    static float x[STEPS][CHANNELS][N], y[STEPS][CHANNELS][N];
    unsigned s = 12345u;
    for (int k = 0; k < STEPS; ++k)
        for (int c = 0; c < CHANNELS; ++c)
            for (int i = 0; i < N; ++i) {
                s = s * 1664525u + 1013904223u;
                x[k][c][i] = (float)((s >> 8) / 8388608.0 - 1.0);
            }
    /* chunk by chunk, as the reference's device loop does (Example1.py:16-18); the last two chunks in one call */
    for (int k = 0; k < STEPS - 2; ++k)
        if (adsp_apply_host(eng, x[k], y[k], 1) != ADSP_OK) {
            fprintf(stderr, "adsp_apply_host: %s\n", adsp_last_error());
            return 1;
        }
    if (adsp_apply_host(eng, x[STEPS - 2], y[STEPS - 2], 2) != ADSP_OK) return 1;

    double worst = 0.0, scale = 0.0;
    for (int c = 0; c < CHANNELS; ++c)
        for (int tau = 0; tau < STEPS * N; ++tau) {
            double acc = 0.0;
            for (int t = 0; t < L; ++t) {
                const int src = tau - N + D - t;
                if (src >= 0) acc += h[t] * x[src / N][c][src % N];
            }
            const double got = y[tau / N][c][tau % N];
            if (fabs(got - acc) > worst) worst = fabs(got - acc);
            if (fabs(acc) > scale) scale = fabs(acc);
        }
    adsp_destroy(eng);
    printf("libadsp ABI %d, RCCL %d, real-spectrum stage %d, max |error| %.3e of %.3f: %s\n", adsp_version(), rccl, is_real, worst, scale,
           worst <= 1e-5 * scale ? "OK" : "FAIL");
    return worst <= 1e-5 * scale ? 0 : 1;
}
