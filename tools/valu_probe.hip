// valu_probe.hip -- VALU throughput per SIMD on gfx950 as a function of waves per SIMD (tools only):
// independent v_mul_f32 / v_pk_mul_f32 / v_rcp_f32 streams, one workgroup on one CU, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a)
{
  float x[8];
  float2 p[4];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int i = 0; i < 4; ++i) p[i] = make_float2(x[2 * i], x[2 * i + 1]);
  long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[u]) : "v"(a));
      if (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[u & 3]) : "v"(p[(u + 1) & 3]));
      if (MODE == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[u]));
      if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[u]) : "v"(a));
      if (MODE == 4) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(x[u]) : "v"(a));
      if (MODE == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[u]) : "v"(a));
    }
  }
  long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
  if (threadIdx.x == 0) reinterpret_cast<long *>(out + (1 << 18))[blockIdx.x] = t1 - t0;
}
int main()
{
  float *d; hipMalloc(&d, 1 << 22);
  const int iters = 4000;
  auto run = [&](const char *name, auto kern) {
    for (int waves : {1, 2, 4, 8, 16}) {          // waves per workgroup = per CU; /4 per SIMD
      kern<<<1, 64 * waves>>>(d, 10, 1.0001f); hipDeviceSynchronize();
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a); kern<<<1, 64 * waves>>>(d, iters, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      long cyc; hipMemcpy(&cyc, d + (1 << 18), 8, hipMemcpyDeviceToHost);
      const double ninstr = (double)iters * 8;
      printf("%-18s %2d waves/CU: %6.2f shader cycles per instr per wave; per SIMD one instr every %5.2f cycles (%0.2f ns/instr/wave)\n", name, waves,
             cyc / ninstr, cyc / ninstr / (waves > 4 ? waves / 4.0 : 1.0), ms * 1e6 / ninstr);
    }
  };
  run("v_mul_f32", k<0>);
  run("v_pk_mul_f32", k<1>);
  run("v_rcp_f32", k<2>);
  run("v_fma_f32", k<3>);
  run("v_div_fixup_f32", k<4>);
  run("v_cndmask_b32", k<5>);
  return 0;
}
