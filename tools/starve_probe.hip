// starve_probe.hip -- does a flood of small workgroups on one stream keep the large workgroups of another stream from being placed?
// Stream A: back-to-back launches of F workgroups x 256 threads that each run ~tA us (the base pyramid: 17408 x 256, ~20 us a
// workgroup).  Stream B: dependent launches of 64 workgroups x NT threads with LDS bytes of dynamic LDS that each run ~tB us (a
// solver chain: 64 x 1024 threads, 93 KB).  Reports B's time per launch alone and under the flood.   (tools only)
//   hipcc -O2 --offload-arch=gfx950 tools/starve_probe.hip -o tools/starve_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
extern __shared__ int dyn[];
__global__ void spin(long ticks, int use_lds)
{
  if (use_lds && threadIdx.x == 0) dyn[0] = 1;
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static double run(hipStream_t sa, hipStream_t sb, bool flood, int nt, int lds, double tb_us, int nB, int fwgs, double ta_us)
{
  hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  if (flood) for (int k = 0; k < 40; ++k) spin<<<fwgs, 256, 0, sa>>>((long)(ta_us * 100), 0);
  for (int m = 0; m < nB; ++m) spin<<<64, nt, lds, sb>>>((long)(tb_us * 100), lds > 0);
  hipStreamSynchronize(sb);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  hipDeviceSynchronize();
  return us / nB;
}
int main()
{
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const double tb = 30.0, ta = 20.0;
  const int nB = 50, fwgs = 17408;
  printf("{\"flood\": \"40 launches of %d workgroups x 256 threads, ~%.0f us per workgroup\", \"victim\": \"%d dependent launches of 64 workgroups, ~%.0f us each\"", fwgs, ta, nB, tb);
  struct { int nt, lds; } cfg[] = {{1024, 93 * 1024}, {1024, 0}, {512, 93 * 1024}, {512, 0}, {256, 93 * 1024}, {256, 0}, {64, 0}};
  for (auto &c : cfg) {
    run(sa, sb, false, c.nt, c.lds, tb, 5, fwgs, ta);
    const double alone = run(sa, sb, false, c.nt, c.lds, tb, nB, fwgs, ta);
    const double flooded = run(sa, sb, true, c.nt, c.lds, tb, nB, fwgs, ta);
    printf(", \"threads_%d_lds_%dk\": {\"us_per_launch_alone\": %.1f, \"us_per_launch_under_flood\": %.1f}", c.nt, c.lds / 1024, alone, flooded);
  }
  printf("}\n");
  return 0;
}
