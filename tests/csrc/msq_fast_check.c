/* msq_fast_check.c -- test infrastructure (CPU): the division-free form of the MSQ quantizer that the HIP kernels use on
 * their critical path (quantized_neural_nets_amd/csrc/gpfq_device.h quant_msq_from_dot) against the reference's form
 * (step_algorithm.py:145-146 and :56:  q = sign(s) * step * min(|floor(s / step + 0.5)|, K),  s = <u, x> / ||x||^2),
 * restated here in plain C with every operation individually rounded (-ffp-contract=off, no fast-math).
 * Whenever the fast form says "ok" its index and value must equal the division form's bit for bit.
 *   usage: msq_fast_check <samples per thread> <seed>     prints: samples, ok fraction, mismatches; exit 1 on a mismatch
 * Arguments are drawn both at random and hugging the rounding boundaries (k - 0.5)(1 + eps), |eps| <= 2^-14, where
 * the two forms could part. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef TOL_UNIT
#define TOL_UNIT 0x1p-18f            /* the device's constant; the test suite also runs a pass with 2^-24 that must FAIL */
#endif

static float sgnf(float x) { return (float)((0.0f < x) - (x < 0.0f)); }

static float exact_form(float v, float n2, float step, float Kf, int* idx)
{
    float s = (n2 > 0.0f) ? v / n2 : 0.0f;
    float z = s / step;
    z = z + 0.5f;
    float r = fminf(fabsf(floorf(z)), Kf);
    float sg = sgnf(s);
    *idx = (int)(sg * r);
    return (sg * step) * r;
}

static int fast_form(float v, float in2, float inv_step, float step, float Kf, float thr, float* q, int* idx)
{
    const float r = (v * in2) * inv_step;
    const float y = r + 0.5f;
    const float fl = floorf(y);
    const float d = (y - fl) - 0.5f;
    float rm;
    rm = fminf(fabsf(fl), Kf);       /* the device spells this line as one v_min_f32 with an |.| modifier */
    *q = __builtin_copysignf(step * rm, v);
    *idx = (int)__builtin_copysignf(rm, v);
    return (__builtin_fabsf(d) < thr) & (__builtin_fabsf(r) >= 0x1p-60f);
}

static uint64_t rng_next(uint64_t* s)
{
    uint64_t x = *s;
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return *s = x;
}
static double uni(uint64_t* s) { return (double)(rng_next(s) >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char** argv)
{
    const long per = argc > 1 ? atol(argv[1]) : 1000000;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], 0, 10) : 1;
    long total = 0, okc = 0, bad = 0;
#pragma omp parallel reduction(+ : total, okc, bad)
    {
        int tid = 0;
#ifdef _OPENMP
        extern int omp_get_thread_num(void);
        tid = omp_get_thread_num();
#endif
        uint64_t st = seed * 0x9E3779B97F4A7C15ull + (uint64_t)(tid + 1) * 0xD1B54A32D192ED03ull;
        for (long i = 0; i < per; ++i) {
            /* column norm over 2^-20 .. 2^36, step over 2^-12 .. 2^4, K in {1, 2, 4, 8, 128} */
            const float n2 = (float)exp2(uni(&st) * 56.0 - 20.0);
            const float step = (float)exp2(uni(&st) * 16.0 - 12.0);
            static const float Ks[6] = {1.0f, 2.0f, 4.0f, 8.0f, 128.0f, 1024.0f};
            const float Kf = Ks[rng_next(&st) % 6];
            float v;
            const unsigned kind = (unsigned)(rng_next(&st) % 8);
            if (kind < 5) {            /* hugging a boundary: z ~ (k - 0.5)(1 + eps) */
                const int k = (int)(rng_next(&st) % (2 * ((int)Kf + 3))) - ((int)Kf + 3);
                const double eps = (uni(&st) * 2.0 - 1.0) * exp2(-14.0 - (double)(rng_next(&st) % 12));
                v = (float)(((double)k - 0.5) * (1.0 + eps) * (double)n2 * (double)step);
            } else if (kind < 7) {     /* anywhere in the alphabet's range and a little beyond */
                v = (float)((uni(&st) * 2.0 - 1.0) * ((double)Kf + 2.0) * (double)n2 * (double)step);
            } else {                   /* tiny, huge, zero */
                const unsigned w = (unsigned)(rng_next(&st) % 4);
                v = w == 0 ? 0.0f : w == 1 ? (float)((uni(&st) - 0.5) * 1e-38) : w == 2 ? (float)((uni(&st) - 0.5) * 1e30)
                                                                                           : -0.0f;
            }
            const float in2 = (n2 > 0.0f) ? 1.0f / n2 : 0.0f;
            const float inv_step = 1.0f / step;
            int ie, iq;
            float qf;
            const float qe = exact_form(v, n2, step, Kf, &ie);
            const int ok = fast_form(v, in2, inv_step, step, Kf, 0.5f - (Kf + 4.0f) * TOL_UNIT, &qf, &iq);
            ++total;
            if (ok) {
                ++okc;
                if (ie != iq || memcmp(&qe, &qf, 4) != 0) {
                    if (bad < 5) fprintf(stderr, "MISMATCH v=%a n2=%a step=%a K=%g exact (%d, %a) fast (%d, %a)\n", v, n2, step, Kf, ie, qe, iq, qf);
                    ++bad;
                }
            }
        }
    }
    /* zero column: the kernels pass c = 0 and v = +0 */
    {
        int ie, iq; float qf;
        const float qe = exact_form(0.0f, 0.0f, 0.1f, 8.0f, &ie);
        const int ok = fast_form(0.0f, 0.0f, 10.0f, 0.1f, 8.0f, 0.5f - 12.0f * TOL_UNIT, &qf, &iq);
        if (ok) ++bad;                 /* a zero dot product always takes the divisions */
        (void)qe; (void)ie;
    }
    printf("%ld samples, %.6f on the division-free path, %ld mismatches\n", total, (double)okc / (double)total, bad);
    return bad ? 1 : 0;
}
