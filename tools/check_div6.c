/* Exhaustive CPU proof of the identity the HIP epilogues rely on (cpp-paddle-ocr_amd/csrc/ocr_common.h):
 * for EVERY f32 bit pattern y, the division-free hard-swish
 *     t = clamp(y + 3, 0, 6); u = y * t; q0 = u * r; e = fma(-6, q0, u); q = copysign(fma(e, r, q0), u)
 * (r = RN(1/6)) equals the contract's  u / 6.0f  bit for bit whenever the range guard
 * 2^-119 <= |y| < 2^125 lets it run; outside the guard the kernels take the division itself.
 * Also reports how many inputs the unguarded formula would get wrong (all of them have a denormal,
 * infinite or NaN q0).   gcc -O2 -fopenmp -ffp-contract=off tools/check_div6.c -lm && ./a.out
 * Test infrastructure only. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float asf(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t asu(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
int main(void) {
  const float r = asf(0x3e2aaaabu);
  long bad = 0, unguarded_bad = 0, guarded = 0;
#pragma omp parallel for reduction(+ : bad, unguarded_bad, guarded) schedule(static)
  for (long i = 0; i < (1L << 32); ++i) {
    const float y = asf((uint32_t)i);
    const float t = fminf(fmaxf(y + 3.0f, 0.0f), 6.0f);
    const float u = y * t;
    const float want = u / 6.0f;
    const float q0 = u * r;
    const float e = fmaf(-6.0f, q0, u);
    const float fast = copysignf(fmaf(e, r, q0), u);
    const int same = (isnan(want) && isnan(fast)) || asu(want) == asu(fast);
    if (!same) unguarded_bad++;
    const float ay = fabsf(y);
    const int fast_ok = ay >= 0x1p-119f && ay < 0x1p+125f; /* false for NaN */
    if (!fast_ok) guarded++;
    if (fast_ok && !same) bad++;
    /* NaN y: the kernels' min3/max3 ignore NaN and may take the fast path; it must then still give NaN */
    if (isnan(y) && !isnan(fast)) bad++;
  }
  printf("mismatches inside the guard: %ld (must be 0); unguarded formula wrong for %ld inputs; guard sends %ld inputs to the division\n",
         bad, unguarded_bad, guarded);
  return bad != 0;
}
