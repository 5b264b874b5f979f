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

// XCD-aware placement for grids of (work items, images): workgroups are dealt round-robin to the 8 XCDs by linear id
// (x fastest), and a one-workgroup-per-image kernel (grid = images) puts image p on XCD p % 8.  Re-deal the (x, y)
// workgroups so that everything of image p runs on XCD p % 8: each stage then finds what the previous one wrote in the L2
// of its own XCD (L2 is per XCD; a miss goes to the Infinity Cache).  Needs gridDim.y % 8 == 0, otherwise plain order.
struct WgId { int x, y; };
__device__ __forceinline__ WgId xcd_local_wg()
{
  WgId id = {(int)blockIdx.x, (int)blockIdx.y};
  if ((gridDim.y & 7) == 0) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, q = lin >> 3;
    id.y = (q / (int)gridDim.x) * 8 + xcd;
    id.x = q % (int)gridDim.x;
  }
  return id;
}

// Wave reduction in the order of the oracle's dis_sum(): the balanced tree v[i] += v[i^1], ^2, ^4, ^8, ^16, ^32.
// a+b == b+a in IEEE, so a lane may add its partner's partial on either side and get the same bits.
//   xor 1, 2 : DPP quad_perm              xor 4 : row_half_mirror (the lanes of a quad already agree)
//   xor 8    : row_mirror                 xor 16: row_bcast:15 into rows 1,3   xor 32: row_bcast:31 into rows 2,3
// After the last step lane 63 holds ((r3+r2)+(r1+r0)) == the tree's value; it is returned wave-uniform (SGPR).
// All six steps are VALU DPP operations -- no LDS round trips on the LK loop's critical path.
__device__ __forceinline__ float wave_sum(float v)
{
#define FOTG_DPP(x, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rmask, 0xF, false))
  v = v + FOTG_DPP(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
  v = v + FOTG_DPP(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
  v = v + FOTG_DPP(v, 0x141, 0xF);   // row_half_mirror
  v = v + FOTG_DPP(v, 0x140, 0xF);   // row_mirror
  v = v + FOTG_DPP(v, 0x142, 0xA);   // row_bcast:15 -> rows 1 and 3 (others add 0)
  v = v + FOTG_DPP(v, 0x143, 0xC);   // row_bcast:31 -> rows 2 and 3
#undef FOTG_DPP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

}  // namespace fotg
