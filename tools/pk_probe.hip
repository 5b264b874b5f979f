// Issue rate of packed-f32 against scalar f32 VALU instructions on gfx950 (is v_pk_mul_f32 / v_pk_add_f32 one pass or two?)
//   hipcc -O2 --offload-arch=gfx950 tools/pk_probe.hip -o tools/pk_probe && tools/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s\"}\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void spin(float *out, int iters, float m, float a)
{
  // 8 independent chains per lane; MODE 0: 8 scalar mul + 8 scalar add, MODE 1: 4 pk mul + 4 pk add (the same 16 flops per lane),
  // MODE 2: 8 scalar fma, MODE 3: 4 pk fma
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 0.001f + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(m)); }
#pragma unroll
        for (int k = 0; k < 8; ++k) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a)); }
      } else if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          v2f v = {x[k], x[k + 1]}; const v2f mm = {m, m};
          asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(mm));
          x[k] = v.x; x[k + 1] = v.y;
        }
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          v2f v = {x[k], x[k + 1]}; const v2f aa = {a, a};
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(aa));
          x[k] = v.x; x[k + 1] = v.y;
        }
      } else if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(m), "v"(a)); }
      } else {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          v2f v = {x[k], x[k + 1]}; const v2f mm = {m, m}, aa = {a, a};
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(mm), "v"(aa));
          x[k] = v.x; x[k + 1] = v.y;
        }
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static float run(float *out, int wgs, int iters)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  spin<MODE><<<wgs, 256>>>(out, 16, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  spin<MODE><<<wgs, 256>>>(out, iters, 0.999f, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main()
{
  float *out;
  CHK(hipMalloc(&out, 4096 * 256 * sizeof(float)));
  const int iters = 20000;
  printf("{");
  for (int wgs : {256, 1024, 2048}) {            // 1, 4, 8 waves per SIMD
    const float t0 = run<0>(out, wgs, iters), t1 = run<1>(out, wgs, iters), t2 = run<2>(out, wgs, iters), t3 = run<3>(out, wgs, iters);
    // instructions per wave: iters * 8 reps * (16 | 8 | 8 | 4); waves per SIMD = wgs * 4 / 1024
    const double wps = wgs * 4 / 1024.0;
    auto cyc = [&](float ms, int n) { return ms * 1e-3 * 2.4e9 / ((double)iters * 8 * n * wps); };
    printf("%s\"waves_per_simd_%d\": {\"scalar_mul_add_ms\": %.3f, \"pk_mul_add_ms\": %.3f, \"scalar_fma_ms\": %.3f, \"pk_fma_ms\": %.3f, "
           "\"cycles_per_scalar_instr\": %.2f, \"cycles_per_pk_mul_add_instr\": %.2f, \"cycles_per_scalar_fma\": %.2f, \"cycles_per_pk_fma\": %.2f}",
           wgs == 256 ? "" : ", ", (int)wps, t0, t1, t2, t3, cyc(t0, 16), cyc(t1, 8), cyc(t2, 8), cyc(t3, 4));
  }
  printf(", \"note\": \"cycles at a nominal 2.4 GHz per wave-instruction and SIMD; 4 = one pass of a wave64 over 16 lanes\"}\n");
  return 0;
}
