// varref_levelpipe.hip.h -- ALL inner iterations of a tall level (more than 96 rows) in ONE launch.
//
// The fixed-point loop of a level (kroeger/refine_variational.cpp:187-221) is `inner` times { data term (compute_smoothness,
// compute_data, sub_laplacian: opticalflow_aux.c:123-438) ; sor_coupled (solver.c:77-421) }.  With one launch per stage every
// sor_coupled call pays the fill of its wavefront -- S = w + h - 1 steps before the last cell is reached, with a handful of
// workgroups busy -- and the data term in between waits for the whole call: 3 x 1 923 steps at 960 x 544.
// But the data term of iteration k at pixel (i, j) only needs the flow of iteration k - 1 on the diagonals i + j - 2 .. i + j + 2,
// and the first sweep of iteration k only needs the system up to the diagonal it is on: everything can run as ONE pipeline of
// stages that follow each other along the anti-diagonals,
//     sweep 1 > sweep 2 > sweep 3 (call 1) > data term 2 > sweep 1 > sweep 2 > sweep 3 (call 2) > data term 3 > ...
// and a level takes S + (stages x their lag) steps instead of inner x (S + lag).  Same cell updates on the same inputs in the
// same order: the same bits as the launch-per-stage path.
//
// Workgroups (256 threads) take their role from a ticket, in dependency order (a role only waits for lower tickets, which are
// running or finished by construction):
//   * TILE (call k, sweep n, band b of 64 rows): the tile pipeline of varref_tiles.hip.h (solver, writer, poller wave; the fourth
//     wave leaves at once).  Differences: the system cells are read with agent-scope loads (another workgroup of THIS launch wrote
//     them), the last sweep's copy of its results into the level's D is written through (the next data term reads it while this
//     call is still running), and the first sweep of a call k >= 2 also waits for the data term of its band's eight tile rows.
//   * DATA (iteration k >= 2, tile row ty of FOTG_TH = 8 rows): vr_data_kernel's tile computation (same device functions) for the
//     tiles (0, ty), (1, ty), ... of its row, each as soon as the last sweep of call k - 1 has passed the tile's last diagonal + 2 in
//     the bands its rows (+- 2) touch; D is read with agent-scope loads, the system cells are written through, and after every
//     tile the row publishes "every cell of this row up to diagonal 32 (tx + 1) - 1 + 8 ty has its system in place".
// Every wait is bounded (a time-out raises the context's stall word like the tile solver's: the host sync points then recompute
// the batch on the single-wave path).  The system of iteration 1 comes from the set-up launch, which also clears the sync words.
#pragma once
#include "varref_tiles.hip.h"

namespace fotg {

#define FOTG_LP_KMAX 8                       // inner iterations one launch can hold (tv_innerit * (level + 1) of the tall levels: <= 5 at 4K)

struct LevelPipeArgs {
  int K;                                     // inner iterations = sor_coupled calls
  int ntr, tiles_x;                          // tile rows / tiles per row of the data term (FOTG_TW x FOTG_TH pixels)
  int *dprog;                                // [pair][FOTG_LP_KMAX][ntr][FOTG_LP_DW] x 32 ints: progress of the data term of iteration k in tile row ty, per workgroup of the row
  float quarter_alpha, half_delta_over3, half_gamma_over3;
  long long *stamps;                         // diagnosis (dbg & 4): per ticket 8 words: start, end, role, first step / first publish (wall clock, 10 ns)
  int dbg;                                   // FOTG_VR_LEVELPIPE_DBG (diagnosis): 1 = the data term waits for the whole previous call, 2 = a call's first sweep for the whole data term
};
__host__ __device__ inline long lp_tile_words(int npairs, int nbs) { return 32L * (1 + (long)npairs * FOTG_LP_KMAX * 4 * nbs); }
#ifndef FOTG_LP_TH
#define FOTG_LP_TH 8                         // rows of a tile row of the data term (a multiple of FOTG_TH; FOTG_TILE_ROWS is a multiple of it)
#endif
#define FOTG_LP_DW 1                         // workgroups per tile row of the data term (tile tx goes to workgroup tx % FOTG_LP_DW)
__host__ __device__ inline long lp_data_words(int npairs, int ntr) { return 32L * (long)npairs * FOTG_LP_KMAX * ntr * FOTG_LP_DW; }

__device__ __forceinline__ void st_sc1_f4(void *p, float4 v)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f x = {v.x, v.y, v.z, v.w};
  // (s_nop: a VALU write to the data registers of a store of more than 8 bytes needs a wait state behind it -- the compiler inserts
  // it for its own stores and does not see into this block)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ int lp_poll(const int *p)
{
  int v;
  asm volatile("global_load_dword %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(0), "s"(p) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void lp_report_timeout(const TileArgs &g)
{
  asm volatile("global_atomic_add %0, %1, %2\n\tglobal_store_dword %0, %1, %3 sc0 sc1\n\ts_waitcnt vmcnt(0)"
               :: "v"(0), "v"(1), "s"(g.timeouts), "s"(g.stall_flag) : "memory");
}

// ------------------------------------------------------------------------------------------------------------------------------
// DATA role: the tiles of tile row ty of iteration k (0-based call index kc = k: this data term feeds call kc, reads call kc - 1)
// ------------------------------------------------------------------------------------------------------------------------------
// The role walks its tile row in MACRO tiles of FOTG_LP_MW pixels (four of vr_data_kernel's tiles side by side): a macro tile costs one
// round of fixed latencies (poll of the producers' progress, the staging loads, the acknowledgement of the written-through system
// cells: ~10 us) whatever its width, and the wavefront of the solver advances FOTG_LP_MW diagonals in 12.8 us -- with 32-pixel tiles
// (3.2 us of wavefront each) the data term set the pace of the whole pipeline.
#ifndef FOTG_LP_PQ
#define FOTG_LP_PQ 2           // poll rounds (one per G diagonals) the poller wave keeps in flight before it waits for the oldest
#endif
#ifndef FOTG_LP_MW
#define FOTG_LP_MW (1 * FOTG_TW)
#endif
template <int NOC, bool FM>
__device__ __forceinline__ void lp_data_role(const VrArgs &a, const TileArgs &g, const LevelPipeArgs &q, int pair, int kc, int ty, int half, int nsweeps, int ticket)
{
  constexpr int MW = FOTG_LP_MW, NPX = MW / FOTG_TW, NPY = FOTG_LP_TH / 8, NP = NPX * NPY;      // pixels per thread: NPX x NPY (32 x 8 threads)
  constexpr int UW = MW + 4, UH = FOTG_LP_TH + 4, SW = MW + 2, SH = FOTG_LP_TH + 2;
  __shared__ float2 uv[UW * UH];
  __shared__ float sm[SW * SH];
  const int st = a.st, w = a.w, h = a.h, S = a.S;
  const int y0 = ty * FOTG_LP_TH;
  const int lx = threadIdx.x % FOTG_TW, ly = threadIdx.x / FOTG_TW;
  const float *wx = a.single(pair, P_WX), *wy = a.single(pair, P_WY);
  const float2 *D = a.Dp(pair);
  // Two workgroups share a tile row, tile tx belongs to workgroup tx % 2: a tile costs a round of fixed latencies (the producers'
  // progress, the staging loads, the acknowledgement of the written-through cells: 4.5-5 us) while the solver's wavefront crosses
  // it in 3.6 us -- one workgroup per row set the pace of every later stage.  Each publishes how far the row is done AS FAR AS IT
  // IS CONCERNED: behind its tile tx that is the end of tile tx + 1 (the other workgroup's); the row's progress is the minimum.
  int *const myprog = q.dprog + 32 * ((((long)pair * FOTG_LP_KMAX + kc) * q.ntr + ty) * FOTG_LP_DW + half);
  // the last sweep of call kc - 1 in the bands whose rows this tile row reads (rows y0 - 2 .. y0 + FOTG_LP_TH + 1)
  const int blo = (y0 - 2 < 0 ? 0 : y0 - 2) / FOTG_TILE_ROWS, bhi0 = (y0 + FOTG_LP_TH + 1) / FOTG_TILE_ROWS, bhi = bhi0 > g.NB - 1 ? g.NB - 1 : bhi0;
  const int *const p0 = g.sync + 32 * (1 + ((pair * FOTG_LP_KMAX + kc - 1) * 4 + nsweeps - 1) * g.NBS + blo);
  const int *const p1 = g.sync + 32 * (1 + ((pair * FOTG_LP_KMAX + kc - 1) * 4 + nsweeps - 1) * g.NBS + bhi);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)a.Cp(pair), 0, (int)((size_t)a.c_pair_stride * 16), 0x00020000);
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  int seen0 = -1, seen1 = -1;
  const int ntx = (w + MW - 1) / MW;
  if (half >= ntx && threadIdx.x == 0) __hip_atomic_store(myprog, 0x3ffffff0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (no tile of its own)
  for (int tx = half; tx < ntx; tx += FOTG_LP_DW) {
    const int x0 = tx * MW;
    // ---- wait: every cell this macro tile reads (its pixels +- 2) has its final value of iteration kc - 1
    {
      const int dlast = x0 + MW - 1 + y0 + FOTG_LP_TH - 1 + 2;
      const int need = (q.dbg & 1) ? 0x3ffffff0 : dlast < S - 1 ? dlast : S - 1;
      if ((threadIdx.x >> 6) == 0) {
        int spins = 0;
        while (seen0 < need || seen1 < need) {
          if (seen0 < need) seen0 = lp_poll(p0);
          if (seen1 < need) seen1 = lp_poll(p1);
          if (seen0 >= need && seen1 >= need) break;
#ifndef FOTG_LP_DSLEEP
#define FOTG_LP_DSLEEP 4
#endif
          __builtin_amdgcn_s_sleep(FOTG_LP_DSLEEP);
          if (++spins >= (1 << 20)) { if ((threadIdx.x & 63) == 0) lp_report_timeout(g); seen0 = seen1 = 0x3fffffff; }
        }
      }
      __syncthreads();
    }
    // this thread's NPX pixels (one per 32-pixel tile of the macro tile): their own inputs first, the loads overlap the staging below
    PixIn<NOC> pin[NP];
    float2 duv[NP];
    bool inimg[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int i = x0 + (e % NPX) * FOTG_TW + lx, j = y0 + (e / NPX) * 8 + ly;
      inimg[e] = i < w && j < h;
      const int ic0 = inimg[e] ? i : 0, jc0 = inimg[e] ? j : 0;
      pin[e] = data_load<NOC>(a, pair, ic0, jc0);
      duv[e] = ld_sc1_f2(&D[a.didx(ic0, jc0)]);
    }
    for (int k = threadIdx.x; k < UW * UH; k += 256) {
      const int jj = clampi(y0 - 2 + k / UW, h), ii = clampi(x0 - 2 + k % UW, w);
      const int qq = jj * st + ii;
      const float2 d = ld_sc1_f2(&D[a.didx(ii, jj)]);
      uv[k] = make_float2(wx[qq] + d.x, wy[qq] + d.y);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < SW * SH; k += 256) {
      const int sy = k / SW, sx = k % SW;
      const int c = (sy + 1) * UW + (sx + 1);
      sm[k] = smooth_w<FM>(uv[c - 1], uv[c], uv[c + 1], uv[c - UW], uv[c + UW], y0 - 1 + sy, h, q.quarter_alpha);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      if (!inimg[e]) continue;
      const int i = x0 + (e % NPX) * FOTG_TW + lx, j = y0 + (e / NPX) * 8 + ly;
      const int sc = ((e / NPX) * 8 + ly + 1) * SW + ((e % NPX) * FOTG_TW + lx + 1);
      const float s_o = sm[sc];
      const float hr = (i < w - 1) ? s_o + sm[sc + 1] : 0.0f;
      const float hl = (i > 0) ? sm[sc - 1] + s_o : 0.0f;
      const float vb = (j < h - 1) ? s_o + sm[sc + SW] : 0.0f;
      const float vt = (j > 0) ? sm[sc - SW] + s_o : 0.0f;
      float4 c0, c1;
      data_term_cell<NOC, FM>(a, i, j, pin[e], hr, hl, vb, vt, duv[e].x, duv[e].y, q.half_delta_over3, q.half_gamma_over3, c0, c1);
      const unsigned off = (unsigned)(a.cidx(i, j) * 16);
      const v4u u0 = {__float_as_uint(c0.x), __float_as_uint(c0.y), __float_as_uint(c0.z), __float_as_uint(c0.w)};
      const v4u u1 = {__float_as_uint(c1.x), __float_as_uint(c1.y), __float_as_uint(c1.z), __float_as_uint(c1.w)};
      __builtin_amdgcn_raw_buffer_store_b128(u0, rsC, off, 0, 16);          // (aux 16 = sc1: written through)
      __builtin_amdgcn_raw_buffer_store_b128(u1, rsC, off + 16, 0, 16);
    }
    // ---- publish: this wave's stores have completed, then all waves', then the row's progress.  The polls for the NEXT tile are
    // requested in front of the wait, so that their round trip overlaps the acknowledgement of the written-through cells.
    int pre0 = -1, pre1 = -1;
    if ((threadIdx.x >> 6) == 0) asm volatile("global_load_dword %0, %2, %3 sc1\n\tglobal_load_dword %1, %2, %4 sc1" : "=&v"(pre0), "=&v"(pre1) : "v"(0), "s"(p0), "s"(p1) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x >> 6) == 0) {
      pre0 = __builtin_amdgcn_readfirstlane(pre0); pre1 = __builtin_amdgcn_readfirstlane(pre1);
      seen0 = pre0 > seen0 ? pre0 : seen0; seen1 = pre1 > seen1 ? pre1 : seen1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int pub = tx + FOTG_LP_DW >= ntx ? 0x3ffffff0 : x0 + FOTG_LP_DW * MW - 1 + y0;
      __hip_atomic_store(myprog, pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (q.stamps && tx / FOTG_LP_DW < 4) q.stamps[(long)ticket * 8 + 3 + tx / FOTG_LP_DW] = wall_clock64();
    }
  }
  if (q.stamps && threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    q.stamps[(long)ticket * 8 + 7] = ((long long)(xcc & 0xf) << 60) | ((long long)(hw & 0xffff) << 44);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// TILE role: varref_tiles.hip.h's tile (sweep n, band b) of call kc
// ------------------------------------------------------------------------------------------------------------------------------
template <int P, bool FMA, bool CSC1>
__device__ __forceinline__ void lp_tile_role(const VrArgs &a, const TileArgs &g, const LevelPipeArgs &q, int pair, int kc, int n, int b, int nsweeps, float omega, int ticket)
{
  constexpr int G = FOTG_TILE_G, BR = FOTG_TILE_ROWS, U = FOTG_TILE_U, RING = 2 * G, W = FOTG_TILE_W;
  static_assert(U % P == 0 && P % G == 0 && U % G == 0 && U % RING == 0, "ring slots and barrier phase are compile-time");
  __shared__ float2 res_ring[RING][BR];
  __shared__ int seen_lds[4];                                     // own, below, top, data term (min over the band's tile rows)
  constexpr int NDW = (FOTG_TILE_ROWS / FOTG_LP_TH) * FOTG_LP_DW;       // data-term words of a band
  static_assert(NDW <= 16 && FOTG_TILE_ROWS % FOTG_LP_TH == 0 && FOTG_LP_TH % 8 == 0, "one poll lane per word, row rotations over 16 lanes");
  __shared__ int dp_lds[NDW];                          // the band's data-term progress words (tile row x workgroup) as the poller last saw them
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (wv == 3) return;                                            // (a barrier counts the waves that have not ended)
  const int S = a.S, RP = a.RP, RPD = a.RPD, h = a.h;
  auto pw = [&](int kk, int nn, int bb) { return g.sync + 32 * (1 + ((pair * FOTG_LP_KMAX + kk) * 4 + nn) * g.NBS + bb); };
  int *const prog = pw(kc, n, b);
  const int *const prog_own = n > 0 ? pw(kc, n - 1, b) : nullptr;
  const int *const prog_bel = (n > 0 && b + 1 < g.NB) ? pw(kc, n - 1, b + 1) : nullptr;
  const int *const prog_top = b > 0 ? pw(kc, n, b - 1) : nullptr;
  // the data term of this call in the band's tile rows (first sweep of calls >= 1; the others follow the first sweep)
  const bool wdata = n == 0 && kc > 0;
  const int ty0 = b * (BR / FOTG_LP_TH), nty = (q.ntr - ty0) < BR / FOTG_LP_TH ? (q.ntr - ty0) : BR / FOTG_LP_TH;
  const int *const dpw = q.dprog + 32 * ((((long)pair * FOTG_LP_KMAX + kc) * q.ntr + ty0) * FOTG_LP_DW);
  // what workgroup hh of tile row ty0 + r claims before it has finished a tile: the tiles in front of its first one are not its
  // business (hh = 1: "up to the end of tile 0"); and no tile row has cells on the diagonals above its first row
  auto floor_of = [&](int r, int hh) { const int y0r = (ty0 + r) * FOTG_LP_TH; const int own = hh * FOTG_LP_MW - 1 + y0r; const int none = y0r - 1; return hh > 0 && own > none ? own : none; };
  float2 *const Dlev = a.Dp(pair);
  float2 *const Xp = g.X + (size_t)pair * g.x_pair_stride;
  const float2 *const Xin = n == 0 ? Dlev : Xp + (size_t)(n - 1) * g.x_buf_stride;
  float2 *const Xout = Xp + (size_t)n * g.x_buf_stride;
  const int pin = n == 0 ? RPD : g.RT, pout = g.RT;
  const bool to_level = n == nsweeps - 1;
  const int rb = b * BR, r1 = rb + lane;
  const int T = ((S + U - 1) / U) * U;
  const int NBAR = T / G + 1;

  if (threadIdx.x < 3) seen_lds[threadIdx.x] = (threadIdx.x == 0 ? prog_own : threadIdx.x == 1 ? prog_bel : prog_top) ? -1 : 0x3fffffff;
  if (threadIdx.x == 3) seen_lds[3] = wdata ? -1 : 0x3fffffff;
  if (threadIdx.x >= 32 && threadIdx.x < 32 + NDW) dp_lds[threadIdx.x - 32] = (wdata && ((int)threadIdx.x - 32) / FOTG_LP_DW < nty) ? -1 : 0x3fffffff;
  __syncthreads();

  // ======================================== poller wave ========================================
  if (wv == 2) {
    typedef __attribute__((address_space(1))) const void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    for (int k = 0; k < NBAR; ++k) {
      asm volatile("s_barrier" ::: "memory");
#ifndef FOTG_LP_POLL_EVERY
#define FOTG_LP_POLL_EVERY 1
#endif
      if (lane == 0 && (k % FOTG_LP_POLL_EVERY) == 0) {
        if (prog_own) __builtin_amdgcn_global_load_lds((gvoid *)prog_own, lds_addr_of(&seen_lds[0]), 4, 0, 16);      // (aux 16 = sc1)
        if (prog_bel) __builtin_amdgcn_global_load_lds((gvoid *)prog_bel, lds_addr_of(&seen_lds[1]), 4, 0, 16);
        if (prog_top) __builtin_amdgcn_global_load_lds((gvoid *)prog_top, lds_addr_of(&seen_lds[2]), 4, 0, 16);
      }
      if (wdata) {
        // the rows' words, one lane each; the minimum of what has landed so far goes to seen_lds[3] (a stale view only delays)
        if (lane < nty * FOTG_LP_DW) __builtin_amdgcn_global_load_lds((gvoid *)(dpw + 32 * lane), lds_addr_of(&dp_lds[0]), 4, 0, 16);
        if (lane == 0) {
          int m = 0x3fffffff;
#pragma unroll
          for (int r = 0; r < NDW; ++r) {
            int v = __hip_atomic_load(&dp_lds[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int fl = floor_of(r / FOTG_LP_DW, r % FOTG_LP_DW);
            v = v < fl ? fl : v;
            m = v < m ? v : m;
          }
          __hip_atomic_store(&seen_lds[3], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * FOTG_LP_PQ) : "memory");          // at most FOTG_LP_PQ intervals' polls in flight
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (FOTG_LP_PQ + 1)) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // nothing may land after the workgroup's LDS is released
    return;
  }

  // ======================================== writer wave ========================================
  if (wv == 1) {
    static_assert(G % 2 == 0, "two diagonals per store instruction");
    const int hi = lane >> 5, q2 = (lane & 31) * 2;
    const bool x1 = rb + q2 < pout, l1 = rb + q2 < RPD;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)Xout, 0, (S + 1 + FOTG_TILE_DUMP) * pout * 8, 0x00020000);
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr int NST = G / 2;
    for (int k = 0; k < NBAR; ++k) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (to_level) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W * (2 * NST + 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W * (NST + 1)) : "memory");
      {
        const int kc2 = k - 2 - W;
        const int pub = kc2 < 0 ? -1 : (kc2 * G + G - 1 < S - 1 ? kc2 * G + G - 1 : S - 1);
        __hip_atomic_store(prog, pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (k == 0) continue;
      const int d0 = (k - 1) * G;
#pragma unroll
      for (int jj = 0; jj < G; jj += 2) {
        const int d = d0 + jj + hi;
        const float4 v = *reinterpret_cast<const float4 *>(&res_ring[d % RING][q2]);
        const bool live = d < S;
        const unsigned dumpo = (unsigned)((S + 1 + (d & (FOTG_TILE_DUMP - 1))) * pout + q2) * 8u;
        const unsigned xo = (live && x1) ? (unsigned)(d * pout + rb + q2) * 8u : dumpo;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rsW, xo, 0, 16);
        if (to_level) {
          // written through: the data term of the next iteration reads these cells while this launch is running
          float2 *const dst = (live && l1) ? Dlev + (size_t)d * RPD + rb + q2 : Xout + (size_t)(S + 1 + (d & (FOTG_TILE_DUMP - 1))) * pout + q2;
          st_sc1_f4(dst, v);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(prog, 0x3ffffff0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }

  // ======================================== solver wave ========================================
  typedef float v2f __attribute__((ext_vector_type(2)));
  const int rmaxin = pin - 1, rmaxc = RP - 1;
  const float om1 = r1 < h ? omega : 0.f;
  const unsigned c1o = (unsigned)(r1 < rmaxc ? r1 : rmaxc) * 32u, c1o16 = c1o + 16u;
  const unsigned i1o = (unsigned)(r1 < rmaxin - 1 ? r1 : rmaxin - 1) * 8u;
  const unsigned tpo = (unsigned)(b > 0 ? rb - 1 : 0) * 8u + 0u * lane;
  const unsigned cpitch = (unsigned)RP * 32u, ipitch = (unsigned)pin * 8u, tpitch = (unsigned)pout * 8u;
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)a.Cp(pair), 0, (S + 1) * cpitch, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc((void *)Xin, 0, (S + 1) * ipitch, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc((void *)Xout, 0, (S + 1) * tpitch, 0x00020000);
  // (the system cells come from another workgroup of this launch: agent-scope loads, aux 16 = sc1)
  // (CSC1 = false: call 0, whose system was written by the set-up launch -- ordinary loads)
#ifndef FOTG_LP_EXP
#define FOTG_LP_EXP 0          // timing-only builds (WRONG results): 2 system cells with ordinary loads in every call, 4 neighbour rows too
#endif
  constexpr int AUXC = (CSC1 && !(FOTG_LP_EXP & 2)) ? 16 : 0, AUXX = (FOTG_LP_EXP & 4) ? 0 : 16;
  auto ld_c = [&](unsigned vo, unsigned so) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsC, vo, so, AUXC)); };
  auto ld_x = [&](__amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned so) { return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, AUXX)); };
  auto ld_x2 = [&](__amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned so) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, AUXX)); };

  int seen_own = prog_own ? -1 : 0x3fffffff, seen_bel = prog_bel ? -1 : 0x3fffffff, seen_top = prog_top ? -1 : 0x3fffffff, seen_dat = wdata ? -1 : 0x3fffffff;
  auto wait_for = [&](const int *p, int &seen, int need) __attribute__((always_inline)) {
    if (seen >= need) return;
    int spins = 0;
    do {
      seen = lp_poll(p);
      if (seen >= need) break;
      __builtin_amdgcn_s_sleep(2);
    } while (++spins < (1 << 20));
    if (seen < need) {
      if (lane == 0) lp_report_timeout(g);
      seen = 0x3fffffff;
    }
  };
  // the data term: the slowest of the band's tile rows.  One poll per LANE (lane r reads word r: all words in one round trip -- the
  // first sweep of a call runs right behind the data term and comes here every few steps), then the minimum over the lanes.
  const int myfl = lane < nty * FOTG_LP_DW ? floor_of(lane / FOTG_LP_DW, lane % FOTG_LP_DW) : 0x3fffffff;
  const int *const mydp = dpw + 32 * (lane < nty * FOTG_LP_DW ? lane : 0);
  auto wait_data = [&](int need) __attribute__((always_inline)) {
    if (seen_dat >= need) return;
    int spins = 0;
    do {
      int v;
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(mydp) : "memory");
      v = v < myfl ? myfl : v;
      // minimum over the first 16 lanes (the others hold "everything"): row rotations, like row_allsum
#define FOTG_DPPI(x, ctrl) __builtin_amdgcn_update_dpp(0x3fffffff, x, ctrl, 0xF, 0xF, false)
      { int o = FOTG_DPPI(v, 0x128); v = o < v ? o : v; }
      { int o = FOTG_DPPI(v, 0x124); v = o < v ? o : v; }
      { int o = FOTG_DPPI(v, 0x122); v = o < v ? o : v; }
      { int o = FOTG_DPPI(v, 0x121); v = o < v ? o : v; }
#undef FOTG_DPPI
      seen_dat = __builtin_amdgcn_readfirstlane(v);
      if (seen_dat >= need) break;
      __builtin_amdgcn_s_sleep(1);
    } while (++spins < (1 << 20));
    if (seen_dat < need) {
      if (lane == 0) lp_report_timeout(g);
      seen_dat = 0x3fffffff;
    }
  };
  long long nblock = 0;                       // diagnosis (stamps): waits in which the wave had to poll for itself (total | own | below | top | data term: 6 bits each from bit 8)
  struct Stage { float4 c1[2]; float4 rb; float2 top; };
  auto load = [&](Stage &st, unsigned oc, unsigned oi, unsigned ot) {
    st.c1[0] = ld_c(c1o, oc); st.c1[1] = ld_c(c1o16, oc);
    st.rb = ld_x2(rsI, i1o, oi);
    st.top = ld_x(rsT, tpo, ot);
  };
  auto issue = [&](Stage &st, int d) {
    const int dc = d < S ? d : S;
    const int dn = d + 1 < S ? d + 1 : S;
    const int dt = (b > 0 && d >= 1 && d - 1 < S) ? d - 1 : S;
    load(st, (unsigned)dc * cpitch, (unsigned)dn * ipitch, (unsigned)dt * tpitch);
  };
  unsigned rc = (unsigned)P * cpitch, ri = (unsigned)(P + 1) * ipitch;
  unsigned rt = b > 0 ? (unsigned)(P - 1) * tpitch : (unsigned)S * tpitch;
  const unsigned tstep = b > 0 ? tpitch : 0;
  const int *const pw_own = prog_own ? prog_own : prog, *const pw_bel = prog_bel ? prog_bel : prog, *const pw_top = prog_top ? prog_top : prog;
  auto need_in_of = [&](int d) { const int dmax = d + G - 1; return dmax + 1 < S - 1 ? dmax + 1 : S - 1; };
  auto need_top_of = [&](int d) { const int dmax = d + G - 1; return dmax - 1 < S - 1 ? dmax - 1 : S - 1; };
  auto need_dat_of = [&](int d) { const int dmax = d + G - 1; return (q.dbg & 2) ? 0x3ffffff0 : dmax < S - 1 ? dmax : S - 1; };
  auto ensure_blocking = [&](int d) __attribute__((always_inline)) {
    wait_for(pw_own, seen_own, need_in_of(d));
    wait_for(pw_bel, seen_bel, need_in_of(d));
    wait_for(pw_top, seen_top, need_top_of(d));
    if (wdata) wait_data(need_dat_of(d));
  };
  auto ensure = [&](int d) __attribute__((always_inline)) {
    const int h0 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&seen_lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)),
              h1 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&seen_lds[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)),
              h2 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&seen_lds[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)),
              h3 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&seen_lds[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    seen_own = h0 > seen_own ? h0 : seen_own;
    seen_bel = h1 > seen_bel ? h1 : seen_bel;
    seen_top = h2 > seen_top ? h2 : seen_top;
    seen_dat = h3 > seen_dat ? h3 : seen_dat;
    if (seen_own < need_in_of(d) || seen_bel < need_in_of(d) || seen_top < need_top_of(d) || seen_dat < need_dat_of(d)) {
      if (q.stamps) nblock += 1 + ((seen_own < need_in_of(d)) << 8) + ((seen_bel < need_in_of(d)) << 14) + ((seen_top < need_top_of(d)) << 20) + ((long long)(seen_dat < need_dat_of(d)) << 26);
      ensure_blocking(d);
    }
  };
  Stage ring[P];
#pragma unroll
  for (int d0 = 0; d0 < P; d0 += G) {
    ensure_blocking(d0);
#pragma unroll
    for (int k = 0; k < G; ++k) issue(ring[(d0 + k) % P], d0 + k);
  }
  if (q.stamps && lane == 0) q.stamps[(long)ticket * 8 + 3] = wall_clock64();          // (the first P diagonals have been requested)
  const float2 o1_ = ld_x(rsI, i1o, 0);
  v2f own1 = {o1_.x, o1_.y};
  v2f prev1 = {0.f, 0.f};
  float hl1 = 0.f;
  auto relax = [&](v2f own, float4 c0, float4 c1, float hl, v2f left, v2f top, v2f right, v2f bottom, float om) {
    const v2f a1 = {c0.x, c0.y}, bb = {c0.z, c0.w};
    const float a22 = c1.x, hr = c1.y, vb = c1.z, vt = c1.w;
    if constexpr (FMA) {
      const v2f vhr = {hr, hr}, vvt = {vt, vt}, vvb = {vb, vb}, vhl = {hl, hl}, vom = {om, om};
      v2f sv = __builtin_elementwise_fma(vhr, right, bb);
      sv = __builtin_elementwise_fma(vvt, top, sv);
      sv = __builtin_elementwise_fma(vvb, bottom, sv);
      const v2f B = __builtin_elementwise_fma(vhl, left, sv);
      const v2f col0 = {c0.x, c0.y}, col1 = {c0.y, a22}, bx = {B.x, B.x}, by = {B.y, B.y};
      v2f tt = __builtin_elementwise_fma(col0, bx, col1 * by);
      tt = tt - own;
      return __builtin_elementwise_fma(vom, tt, own);
    }
    v2f sv = hr * right;
    const v2f vbt = {vb, vt};
    v2f vtt;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(vtt) : "v"(top), "v"(vbt));
    sv = sv + vtt;
    sv = sv + vb * bottom;
    sv = sv + bb;
    const v2f B = hl * left + sv;
    const v2f pa = a1 * B;
    float t0 = pa.x + pa.y, t1 = c0.y * B.x + a22 * B.y;
    asm("" : "+v"(t0));
    asm("" : "+v"(t1));
    v2f tt = {t0, t1};
    tt = tt - own;
    return own + om * tt;
  };
  auto trip = [&](int s0, auto fast_tag) {
    constexpr bool fast = decltype(fast_tag)::value;
    if (q.stamps && lane == 0 && (s0 & 511) == 0 && (s0 >> 9) < 4) q.stamps[(long)ticket * 8 + 4 + (s0 >> 9)] = wall_clock64();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = s0 + u;
      Stage &st = ring[u % P];
      if (u % G == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        ensure(s + P);
      }
      const float o1 = (fast || s < S) ? om1 : 0.f;
      const v2f top1 = {dpp_wave_shr1_old(st.top.x, prev1.x), dpp_wave_shr1_old(st.top.y, prev1.y)};
      const v2f rg1 = {st.rb.x, st.rb.y};
      const v2f bot1 = {st.rb.z, st.rb.w};
      const v2f res1 = relax(own1, st.c1[0], st.c1[1], hl1, prev1, top1, rg1, bot1, o1);
      res_ring[u % RING][lane] = make_float2(res1.x, res1.y);
      prev1 = res1; hl1 = st.c1[1].y;
      own1 = rg1;
      if (fast) { load(st, rc, ri, rt); rc += cpitch; ri += ipitch; rt += tstep; }
      else issue(st, s + P);
    }
  };
  int s0 = 0;
  for (; s0 + U - 1 + P <= S - 2; s0 += U) trip(s0, std::true_type());
  for (; s0 < T; s0 += U) trip(s0, std::false_type());
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (q.stamps && lane == 0) {                 // diagnosis: where the role ran (XCC, HW_ID) and how often it had to poll for itself
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    q.stamps[(long)ticket * 8 + 7] = ((long long)(xcc & 0xf) << 60) | ((long long)(hw & 0xffff) << 44) | (nblock & 0xfffffffffffll);
  }
}

template <int NOC, int P, bool FMA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void vr_level_pipe_kernel(VrArgs a, TileArgs g, LevelPipeArgs q, int nsweeps, float omega)
{
  __shared__ int role_s;
  if (threadIdx.x == 0) role_s = atomicAdd(&g.sync[0], 1);
  __syncthreads();
  const int t = role_s;
  const int pair = t % g.npairs;
  int idx = t / g.npairs;
  const int T = g.NB * nsweeps;                                    // tiles of one call
  // ticket order: tiles of call 0 | data term 1 | tiles of call 1 | data term 2 | ...
  int kc = 0;
  bool isdata = false;
  if (idx >= T) {
    idx -= T;
    const int seg = q.ntr * FOTG_LP_DW + T;
    kc = 1 + idx / seg;
    idx %= seg;
    if (idx < q.ntr * FOTG_LP_DW) isdata = true; else idx -= q.ntr * FOTG_LP_DW;
  }
  if (kc >= q.K) return;
  if (q.stamps && threadIdx.x == 0) { q.stamps[(long)t * 8] = wall_clock64(); q.stamps[(long)t * 8 + 2] = ((long long)(isdata ? 1 : 0) << 60) | ((long long)kc << 40) | idx; }
  if (isdata) {
    lp_data_role<NOC, FMA>(a, g, q, pair, kc, idx / FOTG_LP_DW, idx % FOTG_LP_DW, nsweeps, t);
    if (q.stamps && threadIdx.x == 0) q.stamps[(long)t * 8 + 1] = wall_clock64();
    return;
  }
  // tiles of a call in the order of b + 2 n
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
  if (q.stamps && threadIdx.x == 0) q.stamps[(long)t * 8 + 2] |= ((long long)n << 20) | ((long long)b << 30) | (1ll << 59);
  if (kc == 0) lp_tile_role<P, FMA, false>(a, g, q, pair, kc, n, b, nsweeps, omega, t);
  else lp_tile_role<P, FMA, true>(a, g, q, pair, kc, n, b, nsweeps, omega, t);
  if (q.stamps && threadIdx.x == 0) q.stamps[(long)t * 8 + 1] = wall_clock64();
}

}  // namespace fotg
