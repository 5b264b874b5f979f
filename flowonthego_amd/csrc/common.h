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

// The same idea for a launch whose image dimension is not a multiple of 8 (a single 4K pair): workgroups go to the XCDs round
// robin by linear id, so NEIGHBOURING work items -- which read overlapping data (the covering patches of neighbouring pixels)
// -- sit on eight different XCDs and each XCD's L2 fetches that data for itself (measured: densify_kernel<12,1> at 4K level 2
// fetched 8x its inputs).  Deal every XCD one CONTIGUOUS eighth of the x range instead.  The launch rounds gridDim.x up to a
// multiple of 8; returns -1 for the surplus workgroups.
__device__ __forceinline__ int xcd_banded_x(int nwork)
{
  const int per = ((int)gridDim.x + 7) >> 3;             // work items per XCD (gridDim.x is a multiple of 8)
  const int x = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  return ((int)blockIdx.x >> 3) < per && x < nwork ? x : -1;
}

__device__ __forceinline__ float row_allsum(float v)
{
#define FOTG_DPP(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, false))
  v = v + FOTG_DPP(v, 0x128);        // row_ror:8
  v = v + FOTG_DPP(v, 0x124);        // row_ror:4
  v = v + FOTG_DPP(v, 0x122);        // row_ror:2
  v = v + FOTG_DPP(v, 0x121);        // row_ror:1
#undef FOTG_DPP
  return v;
}
}  // namespace fotg
