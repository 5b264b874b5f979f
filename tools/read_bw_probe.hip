// what a pure streaming read of 1 GB reaches on this GPU (reference for pyr_base_kernel's 6.7 TB/s):
// hipcc --offload-arch=gfx950 -O3 tools/read_bw_probe.hip -o /tmp/read_bw_probe && /tmp/read_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float vf4 __attribute__((ext_vector_type(4)));
template <int K, bool NT>
__global__ __launch_bounds__(256) void rd(const vf4 *__restrict__ p, float *out, size_t n4)
{
  const size_t base = ((size_t)blockIdx.x * 256 * K) + threadIdx.x;
  vf4 acc = {0, 0, 0, 0};
  vf4 v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = NT ? __builtin_nontemporal_load(p + base + (size_t)k * 256) : p[base + (size_t)k * 256];
#pragma unroll
  for (int k = 0; k < K; ++k) acc += v[k];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1.f;
}
template <int K, bool NT>
static void run(const vf4 *d, float *o, size_t n4, const char *name)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned grid = (unsigned)(n4 / (256 * K));
  for (int i = 0; i < 3; ++i) rd<K, NT><<<grid, 256>>>(d, o, n4);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) rd<K, NT><<<grid, 256>>>(d, o, n4);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
  printf("%s: %.4f ms  %.2f TB/s\n", name, ms, n4 * 16.0 / ms / 1e9);
}
int main()
{
  const size_t n4 = (size_t)64 * 2 * 1920 * 1080 / 4;        // the batch's two frames per pair: 1.06 GB
  vf4 *d; float *o;
  (void)hipMalloc(&d, n4 * 16); (void)hipMalloc(&o, 4);
  (void)hipMemset(d, 0, n4 * 16);
  run<4, false>(d, o, n4, "16 B x 4 per lane"); run<8, false>(d, o, n4, "16 B x 8 per lane"); run<16, false>(d, o, n4, "16 B x 16 per lane");
  run<8, true>(d, o, n4, "16 B x 8 per lane, nontemporal"); run<16, true>(d, o, n4, "16 B x 16 per lane, nontemporal");
  return 0;
}
