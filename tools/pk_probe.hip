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


// one instruction type, 8 independent chains per lane, 8 x 8 per loop trip
#define OPK(NAME, ASM) \
__global__ __launch_bounds__(256) void NAME(float *out, int iters, float m, float a) \
{ \
  float x[8]; \
  _Pragma("unroll") for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 0.001f + k; \
  for (int i = 0; i < iters; ++i) { \
    _Pragma("unroll") for (int rep = 0; rep < 8; ++rep) { \
      _Pragma("unroll") for (int k = 0; k < 8; ++k) { asm volatile(ASM : "+v"(x[k]) : "v"(m), "v"(a)); } \
    } \
  } \
  float s = 0.f; \
  _Pragma("unroll") for (int k = 0; k < 8; ++k) s += x[k]; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = s; \
}
OPK(op_mul, "v_mul_f32 %0, %0, %1")
OPK(op_max, "v_max_f32 %0, %0, %1")
OPK(op_addu, "v_add_u32 %0, %0, %1")
OPK(op_lshl, "v_lshlrev_b32 %0, 1, %0")
OPK(op_and, "v_and_b32 %0, %0, %1")
OPK(op_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OPK(op_mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
OPK(op_add_dpp, "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
OPK(op_floor, "v_floor_f32 %0, %0")
OPK(op_cvt, "v_cvt_i32_f32 %0, %0")
OPK(op_rcp, "v_rcp_f32 %0, %0")
OPK(op_mad_u32, "v_mad_u32_u24 %0, %0, %1, %2")
OPK(op_fma, "v_fma_f32 %0, %0, %1, %2")
OPK(op_sub_abs, "v_sub_f32 %0, |%0|, %1")


// selects and compares: the mask in VCC (implicit, VOP2) or in an SGPR pair (VOP3), written once before the loop
__global__ __launch_bounds__(256) void op_cnd_sgpr(float *out, int iters, float m, float a)
{
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 0.001f + k;
  unsigned long long mask = __ballot(threadIdx.x & 1);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[k]) : "v"(m), "s"(mask)); }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void op_cnd_vcc_set(float *out, int iters, float m, float a)
{
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 0.001f + k;
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(m), "v"(a) : "vcc");
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(m) : ); }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void op_cmp(float *out, int iters, float m, float a)
{
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 0.001f + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x[k]), "v"(a) : "vcc"); }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void op_mul_sgpr(float *out, int iters, float m, float a)
{
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 0.001f + k;
  const float ms = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)) ? m : a;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[k]) : "s"(ms)); }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void op_readlane(float *out, int iters, float m, float a)
{
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 0.001f + k;
  int acc = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { int t; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t) : "v"(x[k])); acc ^= t; }
    }
  }
  float s = acc;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// compare + select pairs the way the compiler writes them (mask in VCC, VOP2 select) and with the mask in an SGPR pair
template <int MODE>
__global__ __launch_bounds__(256) void op_cmpsel(float *out, int iters, float m, float a)
{
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 0.001f + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (MODE == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(m), "v"(a) : "vcc");
        if (MODE == 1) asm volatile("v_cmp_lt_f32 s[20:21], %0, %2\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[k]) : "v"(m), "v"(a) : "s20", "s21");
        if (MODE == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(m), "v"(a));
        if (MODE == 3) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\ts_nop 4\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(m), "v"(a) : "vcc");
        if (MODE == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\ts_mov_b64 s[20:21], vcc\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[k]) : "v"(m), "v"(a) : "vcc", "s20", "s21");
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// one compare, then N selects on the same VCC (what a compiler makes of `if (c) { a = ..; b = ..; ... }`)
template <int N, bool E64>
__global__ __launch_bounds__(256) void op_cmp_nsel(float *out, int iters, float m, float a)
{
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 0.001f + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 64 / N; ++rep) {
      if (E64) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" :: "v"(x[rep & 7]), "v"(a) : "s20", "s21");
      else asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x[rep & 7]), "v"(a) : "vcc");
#pragma unroll
      for (int k = 0; k < N; ++k) {
        if (E64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[k & 7]) : "v"(m));
        else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k & 7]) : "v"(m));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*opk_t)(float *, int, float, float);
static float run_op(opk_t k, float *out, int wgs, int iters)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<wgs, 256>>>(out, 16, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<<<wgs, 256>>>(out, iters, 0.999f, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
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
  {
    struct { const char *name; opk_t k; } ops[] = {{"v_mul_f32", op_mul}, {"v_max_f32", op_max}, {"v_fma_f32", op_fma}, {"v_sub_f32_abs", op_sub_abs}, {"v_add_u32", op_addu},
      {"v_lshlrev_b32", op_lshl}, {"v_and_b32", op_and}, {"v_mad_u32_u24", op_mad_u32}, {"v_cndmask_b32", op_cndmask}, {"v_mov_b32_dpp", op_mov_dpp},
      {"v_add_f32_dpp", op_add_dpp}, {"v_cndmask_b32_e64_sgpr_pair", op_cnd_sgpr}, {"v_cndmask_b32_vcc_written_per_64", op_cnd_vcc_set}, {"v_cmp_lt_f32_vcc", op_cmp}, {"v_mul_f32_sgpr_operand", op_mul_sgpr}, {"v_readlane_b32", op_readlane}, {"pair_v_cmp_vcc+v_cndmask_vcc", op_cmpsel<0>}, {"pair_v_cmp_sgpr+v_cndmask_e64_sgpr", op_cmpsel<1>}, {"v_cndmask_b32_e64_vcc_operand", op_cmpsel<2>}, {"pair_v_cmp_vcc+s_nop4+v_cndmask_vcc", op_cmpsel<3>}, {"triple_v_cmp_vcc+s_mov+v_cndmask_e64_sgpr", op_cmpsel<4>}, {"cmp_vcc_then_1_sel_per64", op_cmp_nsel<1, false>}, {"cmp_vcc_then_2_sel_per64", op_cmp_nsel<2, false>}, {"cmp_vcc_then_4_sel_per64", op_cmp_nsel<4, false>}, {"cmp_vcc_then_8_sel_per64", op_cmp_nsel<8, false>}, {"cmp_sgpr_then_4_sel_per64", op_cmp_nsel<4, true>}, {"cmp_sgpr_then_8_sel_per64", op_cmp_nsel<8, true>}, {"v_floor_f32", op_floor}, {"v_cvt_i32_f32", op_cvt}, {"v_rcp_f32", op_rcp}};
    printf(", \"cycles_per_instr_at_8_waves_per_simd\": {");
    bool first = true;
    for (auto &o : ops) {
      const float ms = run_op(o.k, out, 2048, 4000);
      printf("%s\"%s\": %.2f", first ? "" : ", ", o.name, ms * 1e-3 * 2.4e9 / (4000.0 * 64 * 8));
      first = false;
    }
    printf("}");
    for (int wgs : {256, 1024}) {
      printf(", \"cycles_per_instr_at_%d_waves_per_simd\": {", wgs / 256);
      struct { const char *name; opk_t k; } few[] = {{"v_mul_f32", op_mul}, {"v_cndmask_b32", op_cndmask}, {"v_cndmask_b32_e64_sgpr_pair", op_cnd_sgpr}, {"v_mov_b32_dpp", op_mov_dpp}};
      bool f1 = true;
      for (auto &o : few) {
        const float ms = run_op(o.k, out, wgs, 4000);
        printf("%s\"%s\": %.2f", f1 ? "" : ", ", o.name, ms * 1e-3 * 2.4e9 / (4000.0 * 64 * (wgs / 256)));
        f1 = false;
      }
      printf("}");
    }
  }
  printf(", \"note\": \"cycles at a nominal 2.4 GHz per wave-instruction and SIMD; 4 = one pass of a wave64 over 16 lanes\"}\n");
  return 0;
}
