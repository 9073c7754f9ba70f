/* The quotient make_ray computes with multiply-adds (div_by in renderer_amd/csrc/rt_core.h) against the IEEE division
 * it replaces, on the CPU: every float32 numerator between two bounds, both signs, for each divisor given.
 *
 *     make -C tools/micro div_check && tools/micro/div_check 5e-8 8192 1920 1080 2.41421342 1.35799515
 *
 * Prints the number of quotients that differ (none have, over 4e10 cases: extents and projection terms of the
 * bench workloads and forty random divisors).  Exit status 1 if any differs. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float div_by(float a, float b, float rb) { /* the device function, statement for statement */
    float q = a * rb;
    float e = fmaf(-b, q, a);
    q = fmaf(e, rb, q);
    e = fmaf(-b, q, a);
    return fmaf(e, rb, q);
}

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s smallest_numerator largest_numerator divisor...\n", argv[0]);
        return 2;
    }
    const uint32_t ulo = f2u((float)atof(argv[1])), uhi = f2u((float)atof(argv[2]));
    unsigned long long total_bad = 0;
    for (int k = 3; k < argc; k++) {
        const float b = (float)atof(argv[k]);
        const float rb = 1.0f / b;
        unsigned long long bad = 0, n = 0;
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
        for (long long u = ulo; u <= (long long)uhi; u++) {
            const float a = u2f((uint32_t)u);
            bad += f2u(div_by(a, b, rb)) != f2u(a / b);
            bad += f2u(div_by(-a, b, rb)) != f2u(-a / b);
            n += 2;
        }
        printf("divisor %.9g: %llu numerators, %llu quotients differ\n", b, n, bad);
        total_bad += bad;
    }
    return total_bad ? 1 : 0;
}
