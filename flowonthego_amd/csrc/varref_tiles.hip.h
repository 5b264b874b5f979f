// varref_tiles.hip.h -- one sor_coupled call (FDF1.0.1/solver.c:77-421) of a TALL level on many CUs.
//
// Levels of more than 96 rows (the fine levels of the quality presets: 240x136, 480x272, 960x544 at 4K) do not fit the LDS
// solvers; round 1 relaxed them with one workgroup per pair (vr_sor_wide_kernel), i.e. on ONE CU, bound by that CU's L2 path
// (three sweeps x 544 cells x 48 B per diagonal): 1.0 .. 1.5 ms per call, 57 % of a 4K operating-point-4 pair.
//
// Here the call is cut into TILES = (sweep n, band b of 64 rows), one workgroup (solver, writer and poller wave) each, all running at once:
//   tile (b, n) relaxes the cells of its rows diagonal by diagonal (the anti-diagonal wavefront of the lexicographic order) and
//   needs   the NEW values of the row above its band        from tile (b-1, n)    one diagonal back,
//           the OLD values (= sweep n-1) of its own rows     from tile (b,   n-1)  one diagonal ahead,
//           the OLD value of the row below its band          from tile (b+1, n-1)  one diagonal ahead.
// Every dependency points to a smaller b + 2n, so the tiles form a pipeline: each follows its producers at the distance the
// hand-over needs (prefetch depth + publication granularity + latency, ~22 diagonals) and a call takes
// (w + h) + ~22 (bands - 1 + 2 (sweeps - 1)) single-wave steps of ~0.13 us instead of (w + h + 12) steps of 0.7 .. 1 us
// (measured: 66 / 113 / 204 us per call at 240x136 / 480x272 / 960x544; DESIGN.md section 5, "Tile pipeline").
// Same cell updates in the same order as the row-major loop, hence the same bits.
//
// Hand-over through global memory (MI355X_MICROARCH.md, inter-workgroup visibility): sweep n writes the skewed array X[n]
// (sweep 0 reads the level's D, the last sweep also stores its results there), every access of a handed-over cell is a relaxed agent-scope
// 8-byte atomic (global_load / global_store ... sc1).  A tile is THREE waves: the solver wave only loads data (a wave's vector-memory
// operations complete in order, so a load issued behind a write-through store is not back before that store is acknowledged --
// microseconds -- and a data load issued behind a poll of a word another CU keeps writing waits for that poll), hands its
// results to the WRITER wave through a small LDS ring, one barrier per G diagonals, and learns how far its producers are from
// LDS words the POLLER wave keeps fresh with direct-to-LDS loads of the progress words; the writer
// stores the results, and publishes a chunk's progress two chunks later, when a counted vmcnt says those stores have completed
// (stores that have no cell to go to land in a dump area used round robin: write-through stores to ONE line queue up);
// X rows are a whole number of 128-byte lines, bands start on a line, one store instruction writes a band's 64 consecutive
// cells = four whole lines.  Roles come from a ticket counter in the order of b + 2n: a tile
// only waits for tiles holding lower tickets (running or finished by construction); every wait is bounded.
#pragma once
#include <type_traits>
#include "varref.hip.h"

namespace fotg {

struct TileArgs {
  float2 *X;                 // [pair][nsweeps][S+1+FOTG_TILE_DUMP][RT] float2: rows 0..S-1 diagonals (zero outside the image), row S zeros, then the dump area
  long x_pair_stride, x_buf_stride;
  int RT;                    // cells per row, multiple of 16
  int NB;                    // bands of FOTG_TILE_ROWS rows
  int npairs;
  int *sync;                 // [0] ticket, progress of tile (pair, n, b) at [32 * (1 + (pair * 4 + n) * NBS + b)]; zeroed before every launch
  int NBS;                   // bands per (pair, sweep) in that array: the band count of the context's tallest tiled level
  int *timeouts;             // timed-out waits since the context was created
  int *stall_flag;           // host-visible word (pinned host memory): set on a time-out, read by the product API's host sync points
#ifdef FOTG_TILE_STATS
  long long *stats;          // [ticket][16]: see the end of the solver / writer wave (diagnostic builds only)
#endif
};
__host__ __device__ inline long tile_sync_words(int npairs, int nbs) { return 32L * (1 + (long)npairs * 4 * nbs); }
#define FOTG_TILE_ROWS 64      // rows of a band = lanes of the solver wave (one row per lane)
#define FOTG_TILE_DUMP 16      // rows of the dump area behind the zero row of every X buffer
#ifndef FOTG_TILE_G
#define FOTG_TILE_G 4          // progress is published / checked every G diagonals
#endif
#ifndef FOTG_TILE_W
#define FOTG_TILE_W 1          // chunks of write-through stores the writer keeps in flight
#endif
#ifndef FOTG_TILE_U
#define FOTG_TILE_U 32         // steps per loop trip
#endif
#define FOTG_TILE_THREADS 192  // solver wave, writer wave, poller wave
#ifndef FOTG_TILE_DBG
#define FOTG_TILE_DBG 0        // timing-only elimination builds (wrong results): 1 no polls, 2 no writer / barriers, 4 no loads in the loop, 8 blocking polls only, 16 writer without stores
#endif

__device__ __forceinline__ float2 ld_sc1_f2(const float2 *p)
{
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(float2, v);
}
__device__ __forceinline__ void st_sc1_f2(float2 *p, float2 v)
{
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The LDS address of a __shared__ object for a direct-to-LDS load: the low 32 bits of its generic address.  (The address-space cast
// `(address_space(3) void *)&word` carries a null check that this compiler version sometimes lowers to an illegal VALU compare with
// the aperture register -- "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base" -- depending on the code around it.)
typedef __attribute__((address_space(3))) void fotg_lds_void;
__device__ __forceinline__ fotg_lds_void *lds_addr_of(const void *p) { return (fotg_lds_void *)(unsigned)(unsigned long long)p; }
// lane L reads src of lane L-1; lane 0 gets `old`
__device__ __forceinline__ float dpp_wave_shr1_old(float old, float src)
{
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x138, 0xF, 0xF, false));
}

// FMA (fotg_params::fast_math, the tolerance mode): the cell update with fused multiply-adds -- 9 packed instructions instead of 16,
// a shorter dependent chain behind the new left value; not bit-identical to sor_coupled (every product keeps its operands, the
// sums are rounded once instead of twice).  The parity mode instantiates FMA = false.
template <int P, bool FMA = false>
__global__ __launch_bounds__(FOTG_TILE_THREADS) __attribute__((amdgpu_waves_per_eu(1, 2))) void vr_sor_tile_kernel(VrArgs a, TileArgs g, int nsweeps, float omega)
{
  constexpr int G = FOTG_TILE_G, BR = FOTG_TILE_ROWS, U = FOTG_TILE_U, RING = 2 * G, W = FOTG_TILE_W;
  static_assert(U % P == 0 && P % G == 0 && U % G == 0 && U % RING == 0, "ring slots and barrier phase are compile-time");
  __shared__ float2 res_ring[RING][BR];
  __shared__ int role_s;
  __shared__ int seen_lds[4];                                     // the producers' progress words as the poller wave last saw them (own, below, top)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int S = a.S, RP = a.RP, RPD = a.RPD, h = a.h;
  // ---- role: ticket -> (pair, tile), tiles in the order of b + 2n
  if (threadIdx.x == 0) role_s = atomicAdd(&g.sync[0], 1);
  __syncthreads();
  const int t = role_s;
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
  int *const prog = g.sync + 32 * (1 + (pair * 4 + n) * g.NBS + b);
  const int *const prog_own = n > 0 ? g.sync + 32 * (1 + (pair * 4 + n - 1) * g.NBS + b) : nullptr;                      // (b, n-1)
  const int *const prog_bel = (n > 0 && b + 1 < g.NB) ? g.sync + 32 * (1 + (pair * 4 + n - 1) * g.NBS + b + 1) : nullptr;  // (b+1, n-1)
  const int *const prog_top = b > 0 ? g.sync + 32 * (1 + (pair * 4 + n) * g.NBS + b - 1) : nullptr;                       // (b-1, n)
  // ---- arrays: input = values of sweep n-1, output = values of sweep n
  float2 *const Dlev = a.Dp(pair);
  float2 *const Xp = g.X + (size_t)pair * g.x_pair_stride;
  const float2 *const Xin = n == 0 ? Dlev : Xp + (size_t)(n - 1) * g.x_buf_stride;
  float2 *const Xout = Xp + (size_t)n * g.x_buf_stride;
  const int pin = n == 0 ? RPD : g.RT, pout = g.RT;
  const bool to_level = n == nsweeps - 1;                         // the last sweep's results also go back to the level's D (plain stores:
                                                                  // nothing in this launch reads them there)
  const int rb = b * BR, r1 = rb + lane;
  const int T = ((S + U - 1) / U) * U;                             // solver steps (those past S-1 run on the zero row with omega = 0)
  const int NBAR = T / G + 1;                                     // barriers every wave executes

  // the producers' progress as last seen by the poller wave ("everything" for producers a tile does not have)
  if (threadIdx.x < 3) seen_lds[threadIdx.x] = (threadIdx.x == 0 ? prog_own : threadIdx.x == 1 ? prog_bel : prog_top) ? -1 : 0x3fffffff;
  __syncthreads();

  // ======================================== poller wave ========================================
  // Behind every barrier it asks for the three progress words with direct-to-LDS loads (no registers, nothing to wait for):
  // they land in seen_lds whenever they land, and the solver wave reads them there with an LDS load.  The polls used to be
  // vector-memory loads of the solver wave itself: a wave's loads return in issue order, so every poll of a line another CU keeps
  // writing through held back the data loads behind it, and a wait on a poll was a wait on a full memory round trip (a third
  // of the solver's time at 4K).  Only a solver that has caught up with a producer polls itself (a blocking load, below).
  if (wv == 2) {
    if (FOTG_TILE_DBG & (1 | 8)) return;
    typedef __attribute__((address_space(1))) const void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    for (int k = 0; k < NBAR; ++k) {
      asm volatile("s_barrier" ::: "memory");
      if (lane == 0) {
        if (prog_own) __builtin_amdgcn_global_load_lds((gvoid *)prog_own, lds_addr_of(&seen_lds[0]), 4, 0, 16);      // (aux 16 = sc1)
        if (prog_bel) __builtin_amdgcn_global_load_lds((gvoid *)prog_bel, lds_addr_of(&seen_lds[1]), 4, 0, 16);
        if (prog_top) __builtin_amdgcn_global_load_lds((gvoid *)prog_top, lds_addr_of(&seen_lds[2]), 4, 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");            // at most three intervals' polls in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // nothing may land after the workgroup's LDS is released
    return;
  }

  // ======================================== writer wave ========================================
  if ((FOTG_TILE_DBG & 2) && wv == 1) return;
  if (wv == 1) {
    // interval k (behind barrier #k): the results of chunk k-1 (diagonals (k-1) G .. k G - 1) are in the LDS ring.  Every row store
    // is issued by every lane (lanes whose row does not exist in the array store into the spare row S+1 of X, which nobody
    // reads), so a chunk is exactly NST instructions and "all but the newest NST + 1 have completed" = the chunk before the
    // previous one is in memory: its progress is published now.
    // A store instruction writes TWO diagonals, 16 bytes (two rows' cells) per lane: lanes 0..31 the first, lanes 32..63 the
    // second -- half the instructions and half the fabric writes of one 8-byte cell per lane (a write-through store is one
    // fabric write per lane), and the writer, not the solver, was what a step waited for (a chunk's stores are only
    // acknowledged after a memory round trip and at most W chunks are in flight).  The pitches are even: a pair of rows never
    // straddles the end of an array.
    static_assert(G % 2 == 0, "two diagonals per store instruction");
    const int hi = lane >> 5, q2 = (lane & 31) * 2;
    const bool x1 = rb + q2 < pout, l1 = rb + q2 < RPD;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)Xout, 0, (S + 1 + FOTG_TILE_DUMP) * pout * 8, 0x00020000);
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr int NST = G / 2;                                    // X stores per chunk (and as many plain stores into the level's D)
#ifdef FOTG_TILE_STATS
    long long wstall = 0;
#endif
    for (int k = 0; k < NBAR; ++k) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // W chunks (+ their progress stores) may stay in flight: a write-through store is acknowledged after a memory round trip,
      // longer than the solver needs for a chunk
#ifdef FOTG_TILE_STATS
      const long long tw0 = clock64();
#endif
      if (to_level) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W * (2 * NST + 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W * (NST + 1)) : "memory");
#ifdef FOTG_TILE_STATS
      wstall += clock64() - tw0;
#endif
      {
        const int kc = k - 2 - W;                                 // the newest chunk whose stores have completed
        const int pub = kc < 0 ? -1 : (kc * G + G - 1 < S - 1 ? kc * G + G - 1 : S - 1);
        __hip_atomic_store(prog, pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (k == 0 || (FOTG_TILE_DBG & 16)) continue;
      const int d0 = (k - 1) * G;
#pragma unroll
      for (int j = 0; j < G; j += 2) {
        const int d = d0 + j + hi;
        const float4 v = *reinterpret_cast<const float4 *>(&res_ring[d % RING][q2]);
        const bool live = d < S;
        // (stores that have no cell to go to -- diagonals past the end, rows beyond the level's own array -- land in a dump
        // area of FOTG_TILE_DUMP rows used round robin: write-through stores to one and the same line would queue up behind
        // each other)
        const unsigned dumpo = (unsigned)((S + 1 + (d & (FOTG_TILE_DUMP - 1))) * pout + q2) * 8u;
        const unsigned xo = (live && x1) ? (unsigned)(d * pout + rb + q2) * 8u : dumpo;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rsW, xo, 0, 16);       // (aux 16 = sc1, like st_sc1_f2)
        if (to_level) {
          float2 *const dst = (live && l1) ? Dlev + (size_t)d * RPD + rb + q2 : Xout + (size_t)(S + 1 + (d & (FOTG_TILE_DUMP - 1))) * pout + q2;
          *reinterpret_cast<float4 *>(dst) = v;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(prog, 0x3ffffff0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef FOTG_TILE_STATS
    if (lane == 0) { g.stats[t * 32 + 8] = wstall; g.stats[t * 32 + 9] = wall_clock64(); }
#endif
    return;
  }

  // ======================================== solver wave ========================================
  // One row per lane.  Lanes whose row lies beyond the image relax padding cells with omega = 0 (they read zeros, write zeros);
  // rows beyond the arrays' pitch are parked on the last padding cell.  Every address is (wave-uniform running row pointer) +
  // (lane-constant offset); every load is unconditional (what does not exist -- the row above band 0, diagonals < 0 or >= S -- is
  // read from the all-zero row S), so a step is one straight-line block.
  typedef float v2f __attribute__((ext_vector_type(2)));
  const int rmaxin = pin - 1, rmaxc = RP - 1;
  const float om1 = r1 < h ? omega : 0.f;
  // Loads go through buffer resources: address = resource base + per-lane byte offset (VGPR, constant) + row offset (one
  // SGPR per array that advances by the pitch every step) -- no vector and no 64-bit scalar address arithmetic in the loop.
  const unsigned c1o = (unsigned)(r1 < rmaxc ? r1 : rmaxc) * 32u, c1o16 = c1o + 16u;
  // (the 16-byte load of rows r, r+1 stays inside the diagonal's row: lanes parked on the last cells are rows beyond the image)
  const unsigned i1o = (unsigned)(r1 < rmaxin - 1 ? r1 : rmaxin - 1) * 8u;
  const unsigned tpo = (unsigned)(b > 0 ? rb - 1 : 0) * 8u + 0u * lane;
  const unsigned cpitch = (unsigned)RP * 32u, ipitch = (unsigned)pin * 8u, tpitch = (unsigned)pout * 8u;
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)a.Cp(pair), 0, (S + 1) * cpitch, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc((void *)Xin, 0, (S + 1) * ipitch, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc((void *)Xout, 0, (S + 1) * tpitch, 0x00020000);
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  auto ld_c = [&](unsigned vo, unsigned so) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsC, vo, so, 0)); };
  auto ld_x = [&](__amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned so) {        // (aux 16 = sc1: agent-coherent like ld_sc1_f2)
    return __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 16));
  };

  auto ld_x2 = [&](__amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned so) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 16));
  };

  int seen_own = prog_own ? -1 : 0x3fffffff, seen_bel = prog_bel ? -1 : 0x3fffffff, seen_top = prog_top ? -1 : 0x3fffffff;
#ifdef FOTG_TILE_STATS
  long long st_spins[3] = {0, 0, 0}, st_bar = 0, st_vm = 0;
  const long long st_t0 = wall_clock64();
  int st_which = 0;
#endif
  // The solver's own (blocking) polls are hand-placed loads and waits in inline assembly: a load the compiler can see inside the
  // conditional poll code makes it give up its exact vmcnt bookkeeping for the whole step loop -- every interval then waited for
  // all but the data loads it had just issued instead of those of two intervals ago.
  auto poll_now = [&](const int *p) __attribute__((always_inline)) {
    int v;
    asm volatile("global_load_dword %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(0), "s"(p) : "memory");
    return __builtin_amdgcn_readfirstlane(v);                     // (every lane read the same word: keep the control flow scalar)
  };
  auto wait_for = [&](const int *p, int &seen, int need) __attribute__((always_inline)) {
    if (seen >= need) return;
    int spins = 0;
    do {
      seen = poll_now(p);
      if (seen >= need) break;
      __builtin_amdgcn_s_sleep(2);
    } while (++spins < (1 << 20));
#ifdef FOTG_TILE_STATS
    st_spins[st_which] += spins + 1;
#endif
    if (seen < need) {                                            // bounded wait: report and go on (the result is wrong, nothing hangs)
      if (lane == 0) {
        asm volatile("global_atomic_add %0, %1, %2\n\tglobal_store_dword %0, %1, %3 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     :: "v"(0), "v"(1), "s"(g.timeouts), "s"(g.stall_flag) : "memory");
      }
      seen = 0x3fffffff;
    }
  };
  struct Stage { float4 c1[2]; float4 rb; float2 top; };        // rb = (du,dv) of diagonal d+1 at rows r, r+1: the right and the bottom neighbour
  // loads of diagonal step d: the row's system cell, the (du,dv) of diagonal d+1 (right neighbour = next step's own value; the
  // cell below the band), the new value above the band of diagonal d-1
  auto load = [&](Stage &st, unsigned oc, unsigned oi, unsigned ot) {
    st.c1[0] = ld_c(c1o, oc); st.c1[1] = ld_c(c1o16, oc);
    // one 16-byte load per lane: cells (d+1, r) and (d+1, r+1) are neighbours in the row -- lane 63's second cell is the first row
    // of the band below (what a separate wave-uniform load + two DPP moves delivered before); past the end of the array the
    // buffer returns 0
    st.rb = ld_x2(rsI, i1o, oi);
    st.top = ld_x(rsT, tpo, ot);
  };
  auto issue = [&](Stage &st, int d) {
    const int dc = d < S ? d : S;                                 // (row S of C and of the (du,dv) arrays is all zero)
    const int dn = d + 1 < S ? d + 1 : S;
    const int dt = (b > 0 && d >= 1 && d - 1 < S) ? d - 1 : S;
    load(st, (unsigned)dc * cpitch, (unsigned)dn * ipitch, (unsigned)dt * tpitch);
  };
  // The same rows as running offsets for the loop trips whose diagonals need no clamping (all but the last one or two):
  // three scalar adds per step.
  unsigned rc = (unsigned)P * cpitch, ri = (unsigned)(P + 1) * ipitch;           // rows of diagonal d = s + P at s = 0
  unsigned rt = b > 0 ? (unsigned)(P - 1) * tpitch : (unsigned)S * tpitch;
  const unsigned tstep = b > 0 ? tpitch : 0;
  // Before the loads of diagonals d .. d + G - 1 are issued their producers must have published them: the check behind every
  // barrier reads the poller wave's latest view from LDS; in the steady state -- a tile follows its producers at their pace --
  // that is enough, and only a tile that has caught up polls the words itself (the words of producers it does not have read
  // "everything" from the start; the poll addresses are then its own word, never used).
  const int *const pw_own = prog_own ? prog_own : prog, *const pw_bel = prog_bel ? prog_bel : prog, *const pw_top = prog_top ? prog_top : prog;
  auto need_in_of = [&](int d) { const int dmax = d + G - 1; return dmax + 1 < S - 1 ? dmax + 1 : S - 1; };
  auto need_top_of = [&](int d) { const int dmax = d + G - 1; return dmax - 1 < S - 1 ? dmax - 1 : S - 1; };
  auto ensure_blocking = [&](int d) __attribute__((always_inline)) {
    if (FOTG_TILE_DBG & 1) return;
#ifdef FOTG_TILE_STATS
    st_which = 0;
#endif
    wait_for(pw_own, seen_own, need_in_of(d));
#ifdef FOTG_TILE_STATS
    st_which = 1;
#endif
    wait_for(pw_bel, seen_bel, need_in_of(d));
#ifdef FOTG_TILE_STATS
    st_which = 2;
#endif
    wait_for(pw_top, seen_top, need_top_of(d));
  };
  auto ensure = [&](int d) __attribute__((always_inline)) {
    if (FOTG_TILE_DBG & 1) return;
    if (!(FOTG_TILE_DBG & 8)) {
      const int h0 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&seen_lds[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)),
                h1 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&seen_lds[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)),
                h2 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&seen_lds[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      seen_own = h0 > seen_own ? h0 : seen_own;
      seen_bel = h1 > seen_bel ? h1 : seen_bel;
      seen_top = h2 > seen_top ? h2 : seen_top;
    }
    if (seen_own < need_in_of(d) || seen_bel < need_in_of(d) || seen_top < need_top_of(d)) ensure_blocking(d);
  };
  Stage ring[P];
#pragma unroll
  for (int d0 = 0; d0 < P; d0 += G) {
    ensure_blocking(d0);
#pragma unroll
    for (int k = 0; k < G; ++k) issue(ring[(d0 + k) % P], d0 + k);
  }

  // own value of diagonal 0
  const float2 o1_ = ld_x(rsI, i1o, 0);
  v2f own1 = {o1_.x, o1_.y};
  v2f prev1 = {0.f, 0.f};                                         // result of the previous step (new left value; new top value by DPP)
  float hl1 = 0.f;
  // one cell update of sor_coupled in sor_update()'s expression order, the (du,dv) pair as packed f32 (each half rounded like the scalar op)
  auto relax = [&](v2f own, float4 c0, float4 c1, float hl, v2f left, v2f top, v2f right, v2f bottom, float om) {
    const v2f a1 = {c0.x, c0.y}, bb = {c0.z, c0.w};
    const float a22 = c1.x, hr = c1.y, vb = c1.z, vt = c1.w;
    if constexpr (FMA) {
      const v2f vhr = {hr, hr}, vvt = {vt, vt}, vvb = {vb, vb}, vhl = {hl, hl}, vom = {om, om};
      v2f sv = __builtin_elementwise_fma(vhr, right, bb);
      sv = __builtin_elementwise_fma(vvt, top, sv);
      sv = __builtin_elementwise_fma(vvb, bottom, sv);
      const v2f B = __builtin_elementwise_fma(vhl, left, sv);
      // (a11 B.x + a12 B.y, a12 B.x + a22 B.y)
      const v2f col0 = {c0.x, c0.y}, col1 = {c0.y, a22}, bx = {B.x, B.x}, by = {B.y, B.y};
      v2f tt = __builtin_elementwise_fma(col0, bx, col1 * by);
      tt = tt - own;
      return __builtin_elementwise_fma(vom, tt, own);
    }
    v2f sv = hr * right;
    // vt * top with vt read in place (the high half of the register pair the cell was loaded into: the compiler copies it to a
    // pair of its own first)
    const v2f vbt = {vb, vt};
    v2f vtt;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(vtt) : "v"(top), "v"(vbt));
    sv = sv + vtt;
    sv = sv + vb * bottom;
    sv = sv + bb;
    const v2f B = hl * left + sv;
    const v2f pa = a1 * B;
    // two scalar additions (as one packed addition the operands have to be shuffled into pairs first: two copies more)
    float t0 = pa.x + pa.y, t1 = c0.y * B.x + a22 * B.y;
    asm("" : "+v"(t0));
    asm("" : "+v"(t1));
    v2f tt = {t0, t1};
    tt = tt - own;
    return own + om * tt;
  };
  auto trip = [&](int s0, auto fast_tag) {
    constexpr bool fast = decltype(fast_tag)::value;
#ifdef FOTG_TILE_STATS
    if ((s0 & 127) == 0 && (s0 >> 7) < 16 && lane == 0) g.stats[t * 32 + 16 + (s0 >> 7)] = wall_clock64();
#endif
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = s0 + u;
      Stage &st = ring[u % P];
      if (u % G == 0) {                                           // (wave-uniform)
#ifdef FOTG_TILE_STATS
        const long long tb0 = clock64();
#endif
        if (!(FOTG_TILE_DBG & 2)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the writer may take the previous G diagonals
#ifdef FOTG_TILE_STATS
        const long long tb1 = clock64();
        st_bar += tb1 - tb0;
#endif
        ensure(s + P);
      }
      const float o1 = (fast || s < S) ? om1 : 0.f;
      // new top value: the row above is lane - 1; lane 0 has no source lane and keeps `old` = the value loaded from the band above
      const v2f top1 = {dpp_wave_shr1_old(st.top.x, prev1.x), dpp_wave_shr1_old(st.top.y, prev1.y)};
      // old bottom value: the row below is lane + 1; lane 63 keeps the value loaded from the band below
      const v2f rg1 = {st.rb.x, st.rb.y};
      const v2f bot1 = {st.rb.z, st.rb.w};
      const v2f res1 = relax(own1, st.c1[0], st.c1[1], hl1, prev1, top1, rg1, bot1, o1);
      res_ring[u % RING][lane] = make_float2(res1.x, res1.y);     // -> writer wave (U is a multiple of RING: s % RING == u % RING)
      prev1 = res1; hl1 = st.c1[1].y;
      own1 = rg1;
      if (fast && (FOTG_TILE_DBG & 4)) { asm volatile("" : "+v"(st.rb.x), "+v"(st.rb.y), "+v"(st.rb.z), "+v"(st.rb.w), "+v"(st.c1[0].x), "+v"(st.c1[0].y), "+v"(st.c1[0].z), "+v"(st.c1[0].w), "+v"(st.c1[1].x), "+v"(st.c1[1].y), "+v"(st.c1[1].z), "+v"(st.c1[1].w), "+v"(st.top.x), "+v"(st.top.y)); }
      else if (fast) { load(st, rc, ri, rt); rc += cpitch; ri += ipitch; rt += tstep; }
      else issue(st, s + P);
    }
  };
  // U steps per loop trip: the compiler drains the outstanding loads at the loop's back edge, one exposed memory latency per trip
  int s0 = 0;
  for (; s0 + U - 1 + P <= S - 2; s0 += U) trip(s0, std::true_type());
  for (; s0 < T; s0 += U) trip(s0, std::false_type());
  if (!(FOTG_TILE_DBG & 2)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");            // barrier #T/G: the last chunk goes to the writer
#ifdef FOTG_TILE_STATS
  if (lane == 0) {
    long long *o = g.stats + t * 32;
    o[0] = st_t0; o[1] = wall_clock64(); o[2] = st_spins[0]; o[3] = st_spins[1]; o[4] = st_spins[2]; o[5] = st_bar; o[6] = st_vm;
    o[7] = ((long long)pair << 32) | (n << 16) | b;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    o[10] = hw; o[11] = xcc;
  }
#endif
}

}  // namespace fotg
