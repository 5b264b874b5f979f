// queue_probe.hip -- how many kernels of different streams run at once?  S streams x M dependent launches of a kernel that
// occupies `wgs` workgroups for ~T us each; wall time against S (tools only).  GPU_MAX_HW_QUEUES in the environment matters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
__global__ void spin(long ticks, int *sink)
{
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (sink && threadIdx.x == 9999) *sink = 1;
}
int main(int argc, char **argv)
{
  const int M = 200;
  const double T_us = argc > 1 ? atof(argv[1]) : 30.0;
  const int wgs = argc > 2 ? atoi(argv[2]) : 64;
  for (int S : {1, 2, 3, 4, 6, 8}) {
    std::vector<hipStream_t> st(S);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (auto &s : st) spin<<<wgs, 256, 0, s>>>(100, nullptr);
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int m = 0; m < M; ++m)
      for (auto &s : st) spin<<<wgs, 256, 0, s>>>((long)(T_us * 100), nullptr);
    hipDeviceSynchronize();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%d streams x %d launches of %.0f us (%d workgroups): %.1f us per launch round (ideal %.0f), concurrency %.2f\n", S, M, T_us, wgs, us / M, T_us, S * T_us / (us / M));
    for (auto &s : st) hipStreamDestroy(s);
  }
  return 0;
}
