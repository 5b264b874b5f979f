// issue_probe.hip -- single-wave instruction issue/latency calibration on gfx950 (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a, float b)
{
  float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
  float2 p0 = make_float2(x0, x1), p1 = make_float2(x2, x3);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) { x0 = x0 * a + b; }                                         // dependent chain (mul+add: 2 instr, contraction off -> use explicit)
      if (MODE == 1) { x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; }       // 4 independent muls
      if (MODE == 2) { x0 = x0 * a; x0 = x0 + b; }                                 // dependent mul, add
      if (MODE == 3) { p0.x = p0.x * a; p0.y = p0.y * a; p1.x = p1.x * a; p1.y = p1.y * a; }   // may become v_pk_mul
      if (MODE == 4) { x0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x0), 0x138, 0xF, 0xF, true)) * a; }  // dpp on chain
      if (MODE == 5) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(p1)); }
      if (MODE == 6) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x0) : "v"(a)); }
      if (MODE == 7) { asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2" : "+v"(x0), "+v"(x1) : "v"(a)); }
      if (MODE == 8) { asm volatile("s_add_u32 s20, s20, 1" ::: "s20"); }
      if (MODE == 9) { asm volatile("v_mul_f32 %0, %0, %1\n\ts_add_u32 s20, s20, 1" : "+v"(x0) : "v"(a) : "s20"); }
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + p0.x + p0.y + p1.x + p1.y;
}
int main()
{
  float *d; hipMalloc(&d, 1 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 20000;
  auto run = [&](const char *name, auto kern, double instr_per_u) {
    kern<<<64, 64>>>(d, 100, 1.0001f, 1e-7f); hipDeviceSynchronize();
    hipEventRecord(a); kern<<<64, 64>>>(d, iters, 1.0001f, 1e-7f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s %8.2f ns per unrolled unit, %6.2f ns per instr\n", name, ms * 1e6 / (iters * 16.0), ms * 1e6 / (iters * 16.0) / instr_per_u);
  };
  run("0 dependent mul+add (2 instr)", k<0>, 2);
  run("1 four independent muls", k<1>, 4);
  run("2 dependent mul, add", k<2>, 2);
  run("3 float2 muls (pk?)", k<3>, 4);
  run("4 dpp wave_shr + mul chain (2-3 instr)", k<4>, 2);
  run("5 asm v_pk_mul_f32 dependent", k<5>, 1);
  run("6 asm v_mul_f32 dependent", k<6>, 1);
  run("7 asm 2 independent v_mul", k<7>, 2);
  run("8 asm s_add dependent", k<8>, 1);
  run("9 asm v_mul + s_add", k<9>, 2);
  return 0;
}
