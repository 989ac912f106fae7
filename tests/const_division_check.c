/* const_division_check.c -- is  q + fma(-q, C, x) * R  with q = x * R, R = RN(1 / C)  equal to  x / C ?
 * (rgbd-recon_amd/csrc/kernels_pre.hip: divc, the constant divisions of the Lab conversion)
 *
 *   gcc -O2 -fopenmp -ffp-contract=off tests/const_division_check.c -o /tmp/cdc -lm
 *   /tmp/cdc            every binary32 x (4.3e9 per constant; ~25 s per constant on 8 cores)
 *   /tmp/cdc <stride>   every stride-th bit pattern
 * Prints, per constant, the number of x whose results differ (NaN == NaN) and how many of those lie in
 * 1e-30 < |x| < 1e30; exit status 1 if any does.  Full run, 2026-10: mid = 0 for every constant; the
 * differences are -0, +-inf and |x| < 1e-30 (quotients near the denormal range).  */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline float bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t ubits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main(int argc, char** argv)
{
  const long long stride = argc > 1 ? atoll(argv[1]) : 1;
  const float Cs[] = {255.0f, 12.92f, 95.047f, 100.0f, 108.883f, 116.0f};
  int fail = 0;
  for (unsigned ci = 0; ci < sizeof(Cs) / sizeof(Cs[0]); ++ci) {
    const volatile float Cv = Cs[ci];
    const float C = Cv, R = 1.0f / C;
    unsigned long long bad = 0, mid = 0;
#pragma omp parallel for reduction(+ : bad, mid) schedule(static)
    for (long long i = 0; i < (1LL << 32); i += stride) {
      const float x = bits((uint32_t)i);
      const float ref = x / C;
      const float q = x * R;
      const float got = fmaf(fmaf(-q, C, x), R, q);
      if (!(ubits(ref) == ubits(got) || (ref != ref && got != got))) {
        ++bad;
        const float a = fabsf(x);
        if (a > 1e-30f && a < 1e30f) ++mid;
      }
    }
    printf("C=%g R=%a differ=%llu of which 1e-30<|x|<1e30: %llu\n", (double)C, (double)R, bad, mid);
    if (mid) fail = 1;
  }
  return fail;
}
