// bit-exact check of wave_sum_multi<N> (common.h) against the xor 32,16,8,4,2,1 tree: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/reduce_probe.hip -o tools/reduce_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../flowonthego_amd/csrc/common.h"
template <int N>
__global__ void k(const float *in, float *out)
{
  float v[N];
  for (int q = 0; q < N; ++q) v[q] = in[q * 64 + threadIdx.x];
  fotg::wave_sum_multi<N>(v);
  for (int q = 0; q < N; ++q) out[q * 64 + threadIdx.x] = v[q];
}
static float tree(const float *v)
{
  float l[64]; memcpy(l, v, sizeof(l));
  for (int k = 32; k >= 1; k >>= 1) { float t[64]; for (int i = 0; i < 64; ++i) t[i] = l[i] + l[i ^ k]; memcpy(l, t, sizeof(t)); }
  return l[0];
}
template <int N>
static int run(float *d_in, float *d_out, const float *h_in)
{
  float h_out[8 * 64];
  k<N><<<1, 64>>>(d_in, d_out);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int q = 0; q < N; ++q) {
    const float ref = tree(h_in + q * 64);
    for (int l = 0; l < 64; ++l) if (memcmp(&h_out[q * 64 + l], &ref, 4)) ++bad;
  }
  printf("N=%d mismatches=%d\n", N, bad);
  return bad;
}
int main()
{
  float h_in[8 * 64], *d_in, *d_out;
  srand(7);
  for (int i = 0; i < 8 * 64; ++i) h_in[i] = (float)rand() / RAND_MAX * 200.f - 100.f + (float)rand() / RAND_MAX * 1e-3f;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_in));
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  int bad = run<1>(d_in, d_out, h_in) + run<2>(d_in, d_out, h_in) + run<3>(d_in, d_out, h_in) + run<4>(d_in, d_out, h_in) +
            run<5>(d_in, d_out, h_in) + run<6>(d_in, d_out, h_in) + run<8>(d_in, d_out, h_in);
  printf(bad ? "FAIL\n" : "OK\n");
  return bad != 0;
}
