/* oracle/c/lfilter.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of the two CPU loops of the reference's LUFS meter:
 *
 *  - lfilter_df1_f32: torchaudio.functional.lfilter as called at
 *    /root/reference/audiotools/core/loudness.py:122-124 (float32 Direct-Form-I,
 *    zero initial state; mirrors torchaudio's cpu_lfilter_core_loop: FIR part
 *    first, then "o0 -= a_flipped[k] * out[t+k]" for the recursive part,
 *    parallel over rows like at::parallel_for over batch*channel).
 *  - block_energy_f32: julius.core.unfold + .square().sum(2) as at
 *    loudness.py:164-174,214 (K-sample blocks, stride S, zero-padded tail).
 *
 * Built by oracle/c/Makefile into oracle/_build/liboracle_c.so.
 */
#include <stdint.h>
#include <stddef.h>

void lfilter_df1_f32(const float *x, int64_t rows, int64_t T,
                     const double *b64, const double *a64, float *y)
{
    const float a0 = (float)a64[0];
    const float b0 = (float)b64[0] / a0, b1 = (float)b64[1] / a0, b2 = (float)b64[2] / a0;
    const float a1 = (float)a64[1] / a0, a2 = (float)a64[2] / a0;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const float *xr = x + r * T;
        float *yr = y + r * T;
        float x1 = 0.f, x2 = 0.f, y1 = 0.f, y2 = 0.f;
        for (int64_t n = 0; n < T; ++n) {
            const float xn = xr[n];
            float o = b2 * x2;
            o += b1 * x1;
            o += b0 * xn;
            o -= a2 * y2;
            o -= a1 * y1;
            yr[n] = o;
            x2 = x1; x1 = xn; y2 = y1; y1 = o;
        }
    }
}

void block_energy_f32(const float *y, int64_t rows, int64_t T, int64_t K,
                      int64_t S, int64_t nblk, float *z)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const float *yr = y + r * T;
        for (int64_t j = 0; j < nblk; ++j) {
            int64_t lo = j * S, hi = lo + K;
            if (hi > T) hi = T;
            double acc = 0.0;
            for (int64_t n = lo; n < hi; ++n) acc += (double)yr[n] * (double)yr[n];
            z[r * nblk + j] = (float)acc;
        }
    }
}
