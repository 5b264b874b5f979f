// common.h -- shared definitions of the fotg HIP engine (gfx950 / CDNA4 only).
//
// Numerics contract: every kernel evaluates the reference's (kroeger/) expressions in the
// reference's order with separately rounded IEEE f32 mul/add/div/sqrt.  The library is compiled
// with -ffp-contract=off (no FMA formation) and correctly rounded divide/sqrt, so results are
// bit-identical to the CPU oracle (oracle/dis_oracle.c) -- tests compare with ==.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fotg.h"

#define FOTG_WAVE 64
#define FOTG_MAXLEV 12

namespace fotg {

// geometry of one pyramid level (camparam of kroeger/oflow.cpp:138-157 + grid of patchgrid.cpp:31-76)
struct LevelGeom {
  int lvl;
  int w, h;          // unpadded level size
  int tw, th;        // w+2ps, h+2ps
  int st;            // FDF image_t stride = ceil4(w)  (FDF1.0.1/image.c:22)
  int steps, nopw, noph, nop, offw, offh;
  float lb, ubw, ubh;
};

__host__ __device__ inline int clampi(int v, int n) { return v < 0 ? 0 : (v > n - 1 ? n - 1 : v); }
__host__ __device__ inline int reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// Butterfly wave reduction: v[i] += v[i^1], ^2, ^4, ^8, ^16, ^32 -- the balanced tree the oracle's
// dis_sum() defines.  a+b == b+a in IEEE, so every lane ends with the same bits.
// xor 1,2: DPP quad_perm; xor 4: row_half_mirror (lanes of a quad already agree);
// xor 8: row_mirror; xor 16 / 32: ds_swizzle / permlane-free fallback through __shfl_xor.
__device__ __forceinline__ float wave_sum(float v)
{
  v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  v = v + __shfl_xor(v, 16, 64);
  v = v + __shfl_xor(v, 32, 64);
  return v;
}

}  // namespace fotg
