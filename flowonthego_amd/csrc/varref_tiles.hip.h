// varref_tiles.hip.h -- one sor_coupled call (FDF1.0.1/solver.c:77-421) of a TALL level on many CUs.
//
// Levels of more than 96 rows (the fine levels of the quality presets: 240x136, 480x272, 960x544 at 4K) do not fit the LDS
// solvers; round 1 relaxed them with one workgroup per pair (vr_sor_wide_kernel), i.e. on ONE CU, bound by that CU's L2 path
// (three sweeps x 544 cells x 48 B per diagonal): 1.0 .. 1.5 ms per call, 57 % of a 4K operating-point-4 pair.
//
// Here the call is cut into TILES = (sweep n, band b of 128 rows), one single-wave workgroup each, all running at once:
//   tile (b, n) relaxes the cells of its rows diagonal by diagonal (the anti-diagonal wavefront of the lexicographic order) and
//   needs   the NEW values of the row above its band        from tile (b-1, n)    one diagonal back,
//           the OLD values (= sweep n-1) of its own rows     from tile (b,   n-1)  one diagonal ahead,
//           the OLD value of the row below its band          from tile (b+1, n-1)  one diagonal ahead.
// Every dependency points to a smaller b + 2n, so the tiles form a pipeline: each follows its producers at the distance the
// hand-over needs (prefetch depth + publication granularity + latency, ~50 diagonals) and a call takes
// (w + h) + ~50 (bands - 1 + 2 (sweeps - 1)) single-wave steps of ~0.12 us instead of (w + h + 12) steps of 0.7 .. 1 us.
// Same cell updates in the same order as the row-major loop, hence the same bits.
//
// Hand-over through global memory (MI355X_MICROARCH.md, inter-workgroup visibility): sweep n writes the skewed array X[n]
// (sweep 0 reads the level's D, the last sweep also stores its results there), every access of a handed-over cell is a relaxed agent-scope
// 8-byte atomic (global_load / global_store ... sc1); a tile publishes "diagonals <= s - P - 1 are in memory" at step s -- its
// vector-memory operations complete in order, and at step s it has consumed the loads it issued P steps ago, behind those stores;
// X rows are a whole number of 128-byte lines, bands start on a line, a lane's two rows are r and r + 64 so that one store
// instruction writes 64 consecutive cells = four whole lines.  Roles come from a ticket counter in the order of b + 2n: a tile
// only waits for tiles holding lower tickets (running or finished by construction); every wait is bounded.
#pragma once
#include "varref.hip.h"

namespace fotg {

struct TileArgs {
  float2 *X;                 // [pair][nsweeps][S+2][RT] float2, zero outside the image (never written there with non-zeros)
  long x_pair_stride, x_buf_stride;
  int RT;                    // cells per row, multiple of 16
  int NB;                    // bands of 128 rows
  int npairs;
  int *sync;                 // [0] ticket, progress of tile (pair, n, b) at [32 * (1 + (pair * 4 + n) * 32 + b)]; zeroed before every launch
  int *timeouts;             // timed-out waits since the context was created
};
__host__ __device__ inline long tile_sync_words(int npairs) { return 32L * (1 + (long)npairs * 4 * 32); }
#define FOTG_TILE_ROWS 128
#define FOTG_TILE_G 8          // progress is published / checked every G diagonals

__device__ __forceinline__ float2 ld_sc1_f2(const float2 *p)
{
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(float2, v);
}
__device__ __forceinline__ void st_sc1_f2(float2 *p, float2 v)
{
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float dpp_wave_shl1(float v)
{
  // lane L reads lane L+1 (wave_shl:1); lane 63 gets 0 (bound_ctrl)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

template <int P>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void vr_sor_tile_kernel(VrArgs a, TileArgs g, int nsweeps, float omega)
{
  constexpr int G = FOTG_TILE_G, BR = FOTG_TILE_ROWS;
  const int lane = threadIdx.x;
  const int S = a.S, RP = a.RP, RPD = a.RPD, h = a.h;
  // ---- role: ticket -> (pair, tile), tiles in the order of b + 2n
  int t = lane == 0 ? atomicAdd(&g.sync[0], 1) : 0;
  t = __builtin_amdgcn_readfirstlane(t);
  const int pair = t % g.npairs, idx = t / g.npairs;
  int n = -1, b = -1;
  {
    int cnt = 0;
    for (int key = 0; key <= g.NB - 1 + 2 * (nsweeps - 1) && n < 0; ++key)
      for (int nn = 0; nn < nsweeps; ++nn) {
        const int bb = key - 2 * nn;
        if (bb < 0 || bb >= g.NB) continue;
        if (cnt == idx) { n = nn; b = bb; }
        ++cnt;
      }
  }
  if (n < 0) return;
  int *const prog = g.sync + 32 * (1 + (pair * 4 + n) * 32 + b);
  const int *const prog_own = n > 0 ? g.sync + 32 * (1 + (pair * 4 + n - 1) * 32 + b) : nullptr;                      // (b, n-1)
  const int *const prog_bel = (n > 0 && b + 1 < g.NB) ? g.sync + 32 * (1 + (pair * 4 + n - 1) * 32 + b + 1) : nullptr;  // (b+1, n-1)
  const int *const prog_top = b > 0 ? g.sync + 32 * (1 + (pair * 4 + n) * 32 + b - 1) : nullptr;                       // (b-1, n)
  // ---- arrays: input = values of sweep n-1, output = values of sweep n
  float2 *const Dlev = a.Dp(pair);
  float2 *const Xp = g.X + (size_t)pair * g.x_pair_stride;
  const float2 *const Xin = n == 0 ? Dlev : Xp + (size_t)(n - 1) * g.x_buf_stride;
  float2 *const Xout = Xp + (size_t)n * g.x_buf_stride;
  const int pin = n == 0 ? RPD : g.RT, pout = g.RT;
  const bool in_plain = n == 0;                                   // the level's D was written by earlier launches: plain loads
  const bool to_level = n == nsweeps - 1;                         // the last sweep's results also go back to the level's D (plain stores:
                                                                  // nothing in this launch reads them there)
  const int rb = b * BR, r1 = rb + lane, r2 = rb + 64 + lane;
  // lanes whose row lies beyond the image relax padding cells with omega = 0 (they read zeros, write zeros); rows beyond the
  // arrays' pitch are parked on the last padding cell
  const int rmaxin = pin - 1, rmaxout = pout - 1, rmaxc = RP - 1;
  const float om1 = r1 < h ? omega : 0.f, om2 = r2 < h ? omega : 0.f;
  const int c1r = r1 < rmaxc ? r1 : rmaxc, c2r = r2 < rmaxc ? r2 : rmaxc;
  const int i1r = r1 < rmaxin ? r1 : rmaxin, i2r = r2 < rmaxin ? r2 : rmaxin, ibel = rb + BR < rmaxin ? rb + BR : rmaxin;
  const int o1r = r1 < rmaxout ? r1 : rmaxout, o2r = r2 < rmaxout ? r2 : rmaxout;
  const bool st1 = r1 <= rmaxout && r1 < RP, st2 = r2 <= rmaxout && r2 < RP;   // rows that exist in the output array
  const float4 *const C = a.Cp(pair);

  int seen_own = prog_own ? -1 : 0x3fffffff, seen_bel = prog_bel ? -1 : 0x3fffffff, seen_top = prog_top ? -1 : 0x3fffffff;
  auto wait_for = [&](const int *p, int &seen, int need) {
    if (seen >= need) return;
    int spins = 0;
    do {
      seen = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen >= need) break;
      __builtin_amdgcn_s_sleep(2);
    } while (++spins < (1 << 20));
    if (seen < need) {                                            // bounded wait: report and go on (the result is wrong, nothing hangs)
      if (lane == 0) atomicAdd(g.timeouts, 1);
      seen = 0x3fffffff;
    }
  };
  struct Stage { float4 c1[2], c2[2]; float2 right1, right2, below, top; };
  auto load_in = [&](int row, int cell) {
    const float2 *p = Xin + (size_t)row * pin + cell;
    return in_plain ? *p : ld_sc1_f2(p);
  };
  // loads of diagonal step d: system cells of both rows, the (du,dv) of diagonal d+1 (right neighbours = next step's own values;
  // the cell below the band), the new value above the band of diagonal d-1
  auto issue = [&](Stage &st, int d) {
    const int dc = d < S ? d : S;                                 // (row S of C and of the (du,dv) arrays is all zero)
    const int dn = d + 1 < S ? d + 1 : S;
    const float4 *cp = C + ((size_t)dc * RP) * 2;
    st.c1[0] = cp[(size_t)c1r * 2]; st.c1[1] = cp[(size_t)c1r * 2 + 1];
    st.c2[0] = cp[(size_t)c2r * 2]; st.c2[1] = cp[(size_t)c2r * 2 + 1];
    st.right1 = load_in(dn, i1r); st.right2 = load_in(dn, i2r); st.below = load_in(dn, ibel);
    st.top = make_float2(0.f, 0.f);
    if (prog_top && d >= 1 && d - 1 < S) st.top = ld_sc1_f2(Xout + (size_t)(d - 1) * pout + (rb - 1));
  };
  // before the loads of diagonals d .. d + G - 1 are issued their producers must have published them
  auto ensure = [&](int d) {
    const int dmax = d + G - 1;
    const int need_in = dmax + 1 < S - 1 ? dmax + 1 : S - 1, need_top = dmax - 1 < S - 1 ? dmax - 1 : S - 1;
    if (prog_own) wait_for(prog_own, seen_own, need_in);
    if (prog_bel) wait_for(prog_bel, seen_bel, need_in);
    if (prog_top) wait_for(prog_top, seen_top, need_top);
  };
  static_assert(P % G == 0, "the prologue fills the ring in whole check intervals");
  Stage ring[P];
  for (int d0 = 0; d0 < P; d0 += G) {
    ensure(d0);
#pragma unroll
    for (int k = 0; k < G; ++k) issue(ring[(d0 + k) % P], d0 + k);
  }
  // own values of diagonal 0
  float2 own1 = load_in(0, i1r), own2 = load_in(0, i2r);
  float2 prev1 = make_float2(0.f, 0.f), prev2 = prev1;            // results of the previous step (new left values; new top values by DPP)
  float hl1 = 0.f, hl2 = 0.f;
  const int T = ((S + P - 1) / P) * P;                             // steps past S-1 run on the zero row with omega = 0
  for (int s0 = 0; s0 < T; s0 += P) {
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int s = s0 + u;
      Stage &st = ring[u];
      if (u % G == 0) ensure(s + P);                              // (wave-uniform)
      const float o1 = s < S ? om1 : 0.f, o2 = s < S ? om2 : 0.f;
      // new top values: the row above is the same slot of lane - 1 (first row of the band: from the band above; row 64: lane 63's first row)
      float2 top1 = make_float2(dpp_wave_shr1(prev1.x), dpp_wave_shr1(prev1.y));
      float2 top2 = make_float2(dpp_wave_shr1(prev2.x), dpp_wave_shr1(prev2.y));
      const float2 p63 = make_float2(lane_get(prev1.x, 63), lane_get(prev1.y, 63));
      if (lane == 0) { top1 = st.top; top2 = p63; }
      // old bottom values: the row below is the same slot of lane + 1 (row 63: lane 0's second row; row 127: the band below)
      float2 bot1 = make_float2(dpp_wave_shl1(st.right1.x), dpp_wave_shl1(st.right1.y));
      float2 bot2 = make_float2(dpp_wave_shl1(st.right2.x), dpp_wave_shl1(st.right2.y));
      const float2 q0 = make_float2(lane_get(st.right2.x, 0), lane_get(st.right2.y, 0));
      if (lane == 63) { bot1 = q0; bot2 = st.below; }
      const float2 res1 = sor_update(own1, st.c1[0], st.c1[1], hl1, prev1, top1, st.right1, bot1, o1);
      const float2 res2 = sor_update(own2, st.c2[0], st.c2[1], hl2, prev2, top2, st.right2, bot2, o2);
      if (s < S) {
        float2 *row = Xout + (size_t)s * pout;
        if (st1) st_sc1_f2(row + o1r, res1);
        if (st2) st_sc1_f2(row + o2r, res2);
        if (to_level) {
          float2 *lrow = Dlev + (size_t)s * RPD;
          if (r1 < RPD) lrow[r1] = res1;
          if (r2 < RPD) lrow[r2] = res2;
        }
      }
      prev1 = res1; prev2 = res2; hl1 = st.c1[1].y; hl2 = st.c2[1].y;
      own1 = st.right1; own2 = st.right2;
      // this wave's vector-memory operations complete in order: the loads consumed above were issued behind the stores of steps
      // <= s - P - 1, so those diagonals are in memory
      if (u % G == G - 1 && s - P - 1 >= 0) {
        int pv = s - P - 1;
        asm volatile("" : "+v"(pv) : "v"(res1.x), "v"(res2.x));   // (issued behind the consumption of this step's loads)
        __hip_atomic_store(prog, pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      issue(st, s + P);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(prog, 0x3ffffff0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace fotg
