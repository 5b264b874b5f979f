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

// Wave reductions in the order of the oracle's dis_sum(): the balanced tree v[i] += v[i^32], ^16, ^8, ^4, ^2, ^1 over the 64
// lane values (a+b == b+a in IEEE, so a lane may add its partner's partial on either side and get the same bits).
// Several quantities are reduced TOGETHER, packed into the halves / rows of the wave as the tree narrows:
//   xor 32: v_permlane32_swap of two quantities A, B gives {A.lo, B.lo} and {A.hi, B.hi}; their sum holds A's 32 pair sums in
//           lanes 0-31 and B's in lanes 32-63                                        (2 instructions per 2 quantities)
//   xor 16: v_permlane16_swap of two such registers (swaps the odd rows of the first with the even rows of the second) and an
//           add leave ONE quantity per row of 16 lanes                               (2 instructions per 4 quantities)
//   xor 8, 4, 2, 1 inside the rows: four row_ror DPP adds for the four quantities of a register.  Rotations pair other lanes
//           than xor does, but the partner's partial is the same sum (after step xor 8 the values have period 8 in the row,
//           after xor 4 period 4, ...), so the result is the xor tree's, bit for bit.
// Each result is read back wave-uniform (v_readlane -> SGPR).  N quantities cost about 2.5 N + 5 instructions instead of 9 N
// for N separate six-step DPP butterflies, and the LK loop has 8 such reductions per iteration of a two-patch wave.
__device__ __forceinline__ void permlane32_swap(float &a, float &b)
{
  // swaps lanes 32-63 of a with lanes 0-31 of b; two DISTINCT registers are required
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane16_swap(float &a, float &b)
{
  // swaps rows 1, 3 of a with rows 0, 2 of b
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
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
__device__ __forceinline__ float lane_get(float v, int lane)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// v[0..N) -> the N wave sums (uniform), in place
template <int N>
__device__ __forceinline__ void wave_sum_multi(float (&v)[N])
{
  constexpr int NH = (N + 1) / 2;                 // registers after the xor-32 step: halves {v[2i], v[2i+1]}
  constexpr int NQ = (NH + 1) / 2;                // registers after the xor-16 step: rows {h[2i].lo, h[2i+1].lo, h[2i].hi, h[2i+1].hi}
  float h[NH], q[NQ];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    float a = v[2 * i], b = (2 * i + 1 < N) ? v[2 * i + 1] : v[2 * i];
    permlane32_swap(a, b);
    h[i] = a + b;
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    float a = h[2 * i], b = (2 * i + 1 < NH) ? h[2 * i + 1] : h[2 * i];
    permlane16_swap(a, b);
    q[i] = row_allsum(a + b);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const int hi = k / 2, half = k & 1;           // v[k] sits in half `half` of h[hi]
    const int qi = hi / 2, odd = hi & 1;          // h[hi] went to rows {odd, 2 + odd} of q[qi] (lo half -> row odd, hi half -> row 2 + odd)
    v[k] = lane_get(q[qi], 16 * (2 * half + odd));
  }
}

__device__ __forceinline__ float wave_sum(float v)
{
  float x[1] = {v};
  wave_sum_multi<1>(x);
  return x[0];
}

}  // namespace fotg
