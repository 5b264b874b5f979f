// div_hoist_probe.hip -- is  x / d  == the IEEE division sequence with its reciprocal refinement hoisted and its scaling / fix-up
// steps dropped, for every (x, d) the guard of lk.hip.h lets through?   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// -fhip-fp32-correctly-rounded-divide-sqrt tools/div_hoist_probe.hip -o tools/div_hoist_probe && tools/div_hoist_probe [rounds]
// Every thread draws (x, d) pairs: random bit patterns inside the guarded exponent ranges, plus edge classes (powers of two,
// all-ones mantissas, x = k d +- 1 ulp, quotients next to the guard's limits).  Counts the pairs whose guarded fast quotient
// differs in any bit from x / d.  2^35 pairs by default.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include "../flowonthego_amd/csrc/fdiv_hoist.h"

__device__ inline uint32_t rng(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); }
__device__ inline float mk(uint32_t sign, int exp2, uint32_t mant) { return __uint_as_float((sign << 31) | ((uint32_t)(exp2 + 127) << 23) | (mant & 0x7fffffu)); }

__global__ void probe(unsigned long long *bad, unsigned long long *checked, unsigned long long *passed, int rounds, float *first_bad)
{
  uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
  unsigned long long nb = 0, nc = 0, np = 0;
  for (int it = 0; it < rounds; ++it) {
    const uint32_t a = rng(s), b = rng(s), c = rng(s);
    // denominator anywhere a context could hold one; the guard decides
    int ed = (int)(a % 140) - 70;
    uint32_t md = b;
    const uint32_t cls = c & 15;
    if (cls == 0) md = 0; else if (cls == 1) md = 0x7fffff; else if (cls == 2) md = 1; else if (cls == 3) md = 0x400000;
    const float d = mk((c >> 4) & 1, ed, md);
    int ex = ed + (int)((a >> 8) % 100) - 50;
    if (ex < -126) ex = -126; if (ex > 126) ex = 126;
    uint32_t mx = rng(s);
    float x = mk((c >> 5) & 1, ex, mx);
    if (cls == 4) x = d * (float)((a >> 20) & 255);                                   // exact multiples
    if (cls == 5) x = __uint_as_float(__float_as_uint(d * (float)((a >> 20) & 255)) + 1);
    if (cls == 6) x = __uint_as_float(__float_as_uint(d * (float)((a >> 20) & 255)) - 1);
    if (cls == 7) x = 0.0f;
    if (cls == 8) x = -0.0f;
    if (cls == 9) x = mk((c >> 5) & 1, ed - 40 + (int)(a & 3) - 1, mx);              // quotient next to the guard's lower limit
    if (cls == 10) x = mk((c >> 5) & 1, ed + 40 - (int)(a & 3) + 1, mx);             // ... upper limit
    const fotg::InvDiv v = fotg::make_invdiv(d);
    const float q = fotg::fdiv_fast(x, v);
    const bool ok = v.ok && fotg::fdiv_in_range(q);
    const float ref = x / d;
    ++nc;
    if (ok) {
      ++np;
      if (__float_as_uint(q) != __float_as_uint(ref)) { if (nb == 0 && first_bad) { first_bad[0] = x; first_bad[1] = d; first_bad[2] = q; first_bad[3] = ref; } ++nb; }
    }
  }
  atomicAdd(bad, nb); atomicAdd(checked, nc); atomicAdd(passed, np);
}

int main(int argc, char **argv)
{
  const int rounds = argc > 1 ? atoi(argv[1]) : 32768;
  unsigned long long *d_bad, h[3] = {0, 0, 0};
  float *d_first, hf[4] = {0, 0, 0, 0};
  hipMalloc(&d_bad, 3 * sizeof(unsigned long long)); hipMemset(d_bad, 0, 3 * sizeof(unsigned long long));
  hipMalloc(&d_first, 16); hipMemset(d_first, 0, 16);
  hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, d_bad, d_bad + 1, d_bad + 2, rounds, d_first);
  hipDeviceSynchronize();
  hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hf, d_first, 16, hipMemcpyDeviceToHost);
  printf("{\"tool\": \"tools/div_hoist_probe\", \"pairs\": %llu, \"passed_guard\": %llu, \"mismatches\": %llu, \"first_mismatch_x_d_fast_ieee\": [\"%a\", \"%a\", \"%a\", \"%a\"]}\n", h[1], h[2], h[0], hf[0], hf[1], hf[2], hf[3]);
  return h[0] ? 1 : 0;
}
