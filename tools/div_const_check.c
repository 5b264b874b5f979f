// div_const_check.c -- the exhaustive check behind div_nv() of flowonthego_amd/csrc/lk.hip.h: for EVERY finite float x and every
// element count NV = ps*ps*noc of a supported patch, x / NV == fma(fma(-q0, NV, x), r, q0) with r = RN(1/NV), q0 = x * r
// whenever |x| >= 2^-100 (below that the remainder can underflow; -0 also differs in sign).
//   gcc -O2 -mfma -fopenmp -ffp-contract=off -o /tmp/div_const_check tools/div_const_check.c -lm      (about 5 CPU-minutes)
// Output of the committed version: "mismatches ..., of them with |x| >= 2^-100: 0" for 16, 48, 64, 192, 144, 432, 256, 768.
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>
int main(void)
{
  const float bs[] = {16, 48, 64, 192, 144, 432, 256, 768};
  for (int k = 0; k < 8; ++k) {
    const float b = bs[k]; volatile float one = 1.0f; const float r = one / b;
    long bad = 0, bad_guarded = 0;
#pragma omp parallel for reduction(+:bad, bad_guarded) schedule(static)
    for (long i = 0; i < (1L << 32); ++i) {
      uint32_t u = (uint32_t)i; float x; memcpy(&x, &u, 4);
      if (!isfinite(x)) continue;
      volatile float xv = x; const float ref = xv / b;
      const float q0 = x * r; const float rem = fmaf(-q0, b, x); const float q = fmaf(rem, r, q0);
      uint32_t a, c; memcpy(&a, &ref, 4); memcpy(&c, &q, 4);
      if (a != c) { ++bad; if (fabsf(x) >= 0x1p-100f) ++bad_guarded; }
    }
    printf("b = %g: mismatches %ld, of them with |x| >= 2^-100: %ld\n", b, bad, bad_guarded);
  }
  return 0;
}
