// barrier_probe.hip -- cost of a per-step s_barrier for a few near-lockstep waves on gfx950 (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int NI>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a, int active)
{
  __shared__ float sh[1024 * 2];
  const int wv = threadIdx.x >> 6;
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f;
  for (int i = 0; i < iters; ++i) {
    if (wv < active) {
#pragma unroll
      for (int u = 0; u < NI / 2; ++u) { x0 = x0 * a; x1 = x1 * a; }
      sh[threadIdx.x] = x0;
    }
    if (MODE == 1) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    if (MODE == 2) { asm volatile("s_waitcnt lgkmcnt(1)\n\ts_barrier" ::: "memory"); }
    if (wv < active) x1 += sh[(threadIdx.x + 65) & 1023];
  }
  out[blockIdx.x * 1024 + threadIdx.x] = x0 + x1;
}
int main()
{
  float *d; hipMalloc(&d, 64 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 20000;
  auto run = [&](const char *name, auto kern, int waves, int active) {
    kern<<<64, waves * 64>>>(d, 100, 1.0001f, active); hipDeviceSynchronize();
    hipEventRecord(a); kern<<<64, waves * 64>>>(d, iters, 1.0001f, active); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s waves %2d active %2d: %8.2f ns per step\n", name, waves, active, ms * 1e6 / iters);
  };
  for (int w : {1, 3, 4, 6, 8, 16}) {
    run("no barrier, 40 instr", k<0, 40>, w, w);
    run("barrier lgkm(0), 40 instr", k<1, 40>, w, w);
    run("barrier lgkm(1), 40 instr", k<2, 40>, w, w);
  }
  run("barrier lgkm(0), 40 instr", k<1, 40>, 8, 3);
  run("barrier lgkm(0), 40 instr", k<1, 40>, 16, 6);
  run("no barrier, 8 instr", k<0, 8>, 6, 6);
  run("barrier lgkm(0), 8 instr", k<1, 8>, 6, 6);
  return 0;
}
