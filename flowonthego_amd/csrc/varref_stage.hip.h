// varref_stage.hip.h -- the fixed-point loop of one refinement level as a PIPELINE OF WORKGROUPS
// (kroeger/refine_variational.cpp:182-221: inner x { compute_smoothness, compute_data, sub_laplacian, sor_coupled }).
//
// Why: one level is `inner` dependent iterations of (per-pixel data term -> three lexicographic SOR sweeps), and a sweep is a
// dependent chain of w+h-1 anti-diagonals.  Run one after the other on one CU per pair (round 1) the level costs
// inner x (S + 16) diagonal steps and leaves 3/4 of the chip idle at batch 64.  But iteration it+1 needs from iteration it only
// the final (du,dv) of the diagonals s-2 .. s+2 to build the system of diagonal s, and its first sweep then follows two diagonals
// behind -- the iterations can run as a wavefront pipeline.  Here iteration k of a pair is STAGE k = one workgroup (one CU):
//
//   stage k-1  --(du,dv) of iteration k-1, diagonal by diagonal, through global memory + progress flag-->  stage k  --> ...
//
// Every stage streams the anti-diagonals through LDS rings (the structure of vr_sor_stream_kernel) and does ALL the work of its
// iteration on them: smoothness weights (A), data term + sub_laplacian + 2x2 block inverse = the system C (B), the three sweeps,
// and the hand-over of the finished diagonals.  A level of S diagonals and `inner` iterations then takes about
// S + inner x LAG steps instead of inner x S, on inner x pairs CUs instead of pairs.  Same operations in the same order on every
// cell as the sequential loop, so the result is bit-identical (tests compare with ==).
//
// Workgroup = 16 waves, all in lock step by one s_barrier per INTERVAL of M = 4 diagonal steps (a chunk = M diagonals):
//   waves 0-2      solver, sweep n = wave n, two rows per lane, packed f32 (vr_sor_stream_kernel's step)
//   wave 3         loader: polls the predecessor's progress word, loads chunk I of its (du,dv) in interval I, writes it to the D ring
//   wave 7         writer: copies the chunk the last sweep left tW intervals ago to the stage's output buffer and publishes the
//                  progress one interval later (when those stores have completed); the LAST stage writes flow = w + d instead
//   11 data waves  in NG groups of NBW waves; a group takes every NG-th chunk and has NG intervals for it: smoothness weights
//                  (A) of chunk c+2, then the system cells (B) of chunk c
// Timeline of chunk c in a stage (interval numbers): loaded c, visible c+1; A'(c) [diagonals 4c+1..4c+4] in c-2+TB; B(c) in
// c+TB .. c+TB+NG-1; sweep n relaxes it in c+T0+2n (T0 = TB+NG+1; each sweep reads C one and D two diagonals ahead); written out in
// c+tW (tW = T0 + 2(nsweeps-1) + 1), published in c+tW+1.  Ring sizes follow from these numbers (stage_geom()).
//
// Cross-workgroup hand-over (MI355X_MICROARCH.md, "inter-workgroup visibility"): payload and flag are agent-scope relaxed atomics
// (global_store/load ... sc1: write-through, L1-bypassing); the writer waits for its stores (s_waitcnt vmcnt(0)) before the flag
// store; the loader wave that polled is the wave that loads the payload; everybody else sees it through LDS behind a barrier.  A
// row of the hand-over buffer is a whole number of 128-byte lines and each line is written by one store instruction.  Every row
// is written once and read once per launch (each stage has its own output buffer), so no cache can hold an older copy of it.
//
// Who is who: roles are dealt by a ticket counter at workgroup start (ticket t = stage t / pairs of pair t % pairs), never by
// blockIdx: a workgroup only ever waits for a lower ticket, which by construction is running or finished -- no assumption about
// dispatch order or residency, no deadlock when the grid exceeds the chip.  Every wait is bounded (error word, see poll()).
#pragma once
#include "varref.hip.h"

namespace fotg {

struct StageArgs {
  int inner;              // stages per pair = inner iterations of the level
  int npairs;
  float qa, hd, hg, omega;
  float *flow;            // [pair][h][w][2], written by the last stage
  long flow_stride;
  float2 *DS;             // hand-over buffers [pair][inner-1][rows][RD] float2 (stage k writes buffer k)
  long ds_pair_stride, ds_stage_stride;   // in float2
  int *sync;              // [0] ticket counter, [1] timed-out waits, progress word of (pair, stage k) at [32 * (1 + pair * 16 + k)]
  unsigned long long *stamps;   // -DFOTG_STAGE_STAMPS builds only: [ticket][wave][8] s_memrealtime stamps (100 MHz)
};
#define FOTG_STAGE_MAXINNER 16
__host__ __device__ inline long stage_sync_words(int npairs) { return 32L * (1 + (long)npairs * FOTG_STAGE_MAXINNER); }

// schedule constants shared by host (LDS size, eligibility) and device
struct StageGeom { int nl, HR, E, omax, NBW, NG, T0, tW, RDN, NCW, NCB, NI; };
#define FOTG_STAGE_M 4
#define FOTG_STAGE_SD 8        // diagonals between consecutive sweeps (sor_sync_wave's DS1 for M = 4)
#define FOTG_STAGE_TB 5
#define FOTG_STAGE_LA 3        // intervals between the issue of a chunk's hand-over loads and its arrival in the D ring
#define FOTG_STAGE_RCN 32      // C ring slots (diagonals)
#define FOTG_STAGE_NS 24       // smoothness ring slots
#define FOTG_STAGE_NDW 11      // data waves
#ifndef FOTG_STAGE_DBG
#define FOTG_STAGE_DBG 0       // timing experiments in separate builds only (wrong results): 1 loader loads nothing, 2 no data term,
#endif                         // 4 no smoothness weights, 8 writer stores nothing, 16 solver waves only count barriers
__host__ __device__ inline StageGeom stage_geom(int w, int h, int nsweeps)
{
  constexpr int M = FOTG_STAGE_M, U = 16, UT = 8;
  StageGeom q;
  const int S = w + h - 1;
  q.nl = (h + 1) / 2; q.HR = 2 * q.nl;
  int E = 0;
  while (E + U <= S) E += U;
  while (E < S) E += UT;
  q.E = E;
  q.omax = (nsweeps > 0 ? nsweeps - 1 : 0) * FOTG_STAGE_SD;
  q.NBW = (M * q.HR + 63) / 64;
  q.NG = 2 * q.NBW <= FOTG_STAGE_NDW ? 2 : 1;
  q.T0 = FOTG_STAGE_TB + q.NG + 1;
  q.tW = q.T0 + q.omax / M + 1;
  q.RDN = M * (q.tW + 1);
  q.NCW = (S + M - 1) / M;
  q.NCB = (E + 1 + M) / M;
  q.NI = q.T0 + E / M + q.omax / M + 4;
  return q;
}
template <int RD, int RCW>
__host__ __device__ inline int stage_lds_bytes(const StageGeom &q)
{
  return 128 + FOTG_STAGE_RCN * 2 * RCW * 16 + q.RDN * RD * 8 + FOTG_STAGE_NS * RD * 4 + RD * 8;
}

// relaxed agent-scope accesses: global_load / global_store ... sc1
__device__ __forceinline__ float2 ld_agent_f2(const float2 *p)
{
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(float2, v);
}
__device__ __forceinline__ void st_agent_f2(float2 *p, float2 v)
{
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NOC, int RD, int RCW>
__global__ __launch_bounds__(1024) void vr_stage_kernel(VrArgs a, StageArgs g)
{
  constexpr int M = FOTG_STAGE_M, SD = FOTG_STAGE_SD, TB = FOTG_STAGE_TB, RCN = FOTG_STAGE_RCN, NS = FOTG_STAGE_NS, U = 16, UT = 8;
  constexpr int DB = RD * 8, CB = RCW * 16, CSLOT = 2 * CB, SB = RD * 4;          // bytes per ring slot: D, C plane, C, smoothness
  static_assert(RD % 16 == 0, "a hand-over row is a whole number of 128-byte lines");
  static_assert(RD <= 128, "two 8-byte accesses per lane cover a row");
  typedef float v2f __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int w = a.w, h = a.h, S = a.S, RQ = a.RPD;
  const StageGeom q = stage_geom(w, h, a.nsweeps);
  const int NI = q.NI, RDN = q.RDN, NG = q.NG, T0 = q.T0;
  const unsigned CRING = (unsigned)RCN * CSLOT, DRING = (unsigned)RDN * DB;
  const unsigned CBASE = 128, DBASE = CBASE + CRING, SBASE = DBASE + DRING, DUMP = SBASE + NS * SB;
  char *lds = lds_bytes();
  auto ld_f2 = [&](unsigned off) { return *reinterpret_cast<const float2 *>(lds + off); };
  auto ld_f4 = [&](unsigned off) { return *reinterpret_cast<const float4 *>(lds + off); };
  auto ld_f1 = [&](unsigned off) { return *reinterpret_cast<const float *>(lds + off); };
#define FOTG_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef FOTG_STAGE_STAMPS
  int stamp_ticket = 0;
#define FOTG_STAMP(slot) do { if (g.stamps && lane == 0) g.stamps[((size_t)stamp_ticket * 16 + wv) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define FOTG_STAMP_AT(I) do { if ((I) == 8) FOTG_STAMP(1); else if ((I) == 24) FOTG_STAMP(2); else if ((I) == 40) FOTG_STAMP(3); else if ((I) == 56) FOTG_STAMP(4); } while (0)
#else
#define FOTG_STAMP(slot) do { } while (0)
#define FOTG_STAMP_AT(I) do { } while (0)
#endif

  // ---- role: ticket -> (stage, pair), stage-major: every stage k-1 workgroup holds a lower ticket than any stage k one
  if (threadIdx.x == 0) *reinterpret_cast<int *>(lds) = atomicAdd(&g.sync[0], 1);
  if (threadIdx.x < RD) *reinterpret_cast<float2 *>(lds + DUMP + threadIdx.x * 8) = make_float2(0.f, 0.f);
  __syncthreads();
  const int ticket = *reinterpret_cast<const int *>(lds);
  const int stage = ticket / g.npairs, pair = ticket - stage * g.npairs;
#ifdef FOTG_STAGE_STAMPS
  stamp_ticket = ticket;
#endif
  FOTG_STAMP(0);
  if (stage >= g.inner) return;                                    // (grid = inner * npairs workgroups: never)
  const bool last = stage == g.inner - 1;
  int *const prog_out = g.sync + 32 * (1 + pair * FOTG_STAGE_MAXINNER + stage);
  const int *const prog_in = g.sync + 32 * (1 + pair * FOTG_STAGE_MAXINNER + stage - 1);
  float2 *const out_buf = g.DS + (size_t)pair * g.ds_pair_stride + (size_t)stage * g.ds_stage_stride;          // (unused by the last stage)
  const float2 *const in_buf = g.DS + (size_t)pair * g.ds_pair_stride + (size_t)(stage > 0 ? stage - 1 : 0) * g.ds_stage_stride;

  // ================================================= solver waves =================================================
  if (wv < 3) {
    if (wv >= a.nsweeps || (FOTG_STAGE_DBG & 16)) { for (int I = 0; I < NI; ++I) FOTG_BAR(); return; }
    const int off = T0 * M + wv * SD;
    const int nl = q.nl;
    const float om0 = g.omega, om1 = (2 * lane + 1 < h) ? g.omega : 0.f;
    const unsigned vD = DBASE + (unsigned)lane * 16, vC = CBASE + (unsigned)lane * 32;
    int nbar = 0;
    for (; nbar < off / M; ++nbar) FOTG_BAR();
    if (lane < nl) {
      unsigned d0 = 0, d1 = DB, d2 = 2 * DB, c1o = CSLOT;
      float4 ow = ld_f4(d0 + vD);                                 // own values of rows 2L, 2L+1 (diagonal s)
      float4 nx = ld_f4(d1 + vD);                                 // diagonal s+1: rows 2L, 2L+1 = right of both rows, bottom of row 2L
      float2 nb = ld_f2(d1 + vD + 16);                            // diagonal s+1, row 2L+2: bottom of row 2L+1
      float4 ca0 = ld_f4(vC), cb0 = ld_f4(vC + 16), ca1 = ld_f4(vC + CB), cb1 = ld_f4(vC + CB + 16);   // cells (plane 0 | plane 1) of both rows
      v2f p0 = {0.f, 0.f}, p1 = {0.f, 0.f};
      float hl0 = 0.f, hl1 = 0.f;
      auto relax = [&](v2f own, float4 c0, float4 c1, float hl, v2f left, v2f top, v2f right, v2f bottom, float om) {
        const v2f a1 = {c0.x, c0.y}, bb = {c0.z, c0.w};
        const float a22 = c1.x, hr = c1.y, vb = c1.z, vt = c1.w;
        v2f sv = hr * right;
        sv = sv + vt * top;
        sv = sv + vb * bottom;
        sv = sv + bb;
        const v2f B = hl * left + sv;
        const v2f pa = a1 * B;
        v2f t = {pa.x + pa.y, c0.y * B.x + a22 * B.y};
        t = t - own;
        return own + om * t;
      };
      auto step2 = [&](auto tail_tag, int u, int s) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const float o0 = (!TAIL || s < S) ? om0 : 0.f, o1 = (!TAIL || s < S) ? om1 : 0.f;
        // top of row 2L: row 2L-1's result of the previous step, in lane L-1 (lane 0: no row above, 0)
        const v2f top0 = {dpp_wave_shr1(p1.x), dpp_wave_shr1(p1.y)};
        if (u % M == 0) FOTG_BAR();
        const float4 nnx = ld_f4(d2 + vD);                        // diagonal s+2
        const float2 nnb = ld_f2(d2 + vD + 16);
        const float4 na0 = ld_f4(c1o + vC), nb0 = ld_f4(c1o + vC + 16), na1 = ld_f4(c1o + vC + CB), nb1 = ld_f4(c1o + vC + CB + 16);
        const v2f own0 = {ow.x, ow.y}, own1 = {ow.z, ow.w}, r0 = {nx.x, nx.y}, r1 = {nx.z, nx.w}, bt1 = {nb.x, nb.y};
        const v2f q0 = relax(own0, ca0, ca1, hl0, p0, top0, r0, r1, o0);
        const v2f q1 = relax(own1, cb0, cb1, hl1, p1, p0, r1, bt1, o1);      // its top (s-1, 2L) is this lane's previous row-0 result
        *reinterpret_cast<float4 *>(lds + ((!TAIL || s < S) ? d0 + vD : DUMP + (unsigned)lane * 16)) = make_float4(q0.x, q0.y, q1.x, q1.y);
        hl0 = ca1.y; hl1 = cb1.y;
        p0 = q0; p1 = q1; ow = nx; nx = nnx; nb = nnb; ca0 = na0; cb0 = nb0; ca1 = na1; cb1 = nb1;
        d0 = d1; d1 = d2; d2 += DB; if (d2 == DRING) d2 = 0;
        c1o += CSLOT; if (c1o == CRING) c1o = 0;
      };
      int t0 = 0;
      for (; t0 + U <= S; t0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) step2(std::false_type{}, u, t0 + u);
      }
      for (; t0 < S; t0 += UT) {
#pragma unroll
        for (int u = 0; u < UT; ++u) step2(std::true_type{}, u, t0 + u);
      }
    }                                                             // (the wave executes the loop's E / M barriers once, whatever its exec mask)
    nbar += q.E / M;
    FOTG_STAMP(5);
    for (; nbar < NI; ++nbar) FOTG_BAR();
    FOTG_STAMP(6);
    return;
  }

  // ================================================= loader =================================================
  if (wv == 3) {
    // Chunk c (diagonals 4c .. 4c+3) is written to the D ring in interval c and visible to everybody from interval c+1; its loads
    // are ISSUED LA intervals earlier (an agent-scope load is a 0.5-2 us round trip behind the data waves' traffic -- more than an
    // interval), as soon as the predecessor's progress word covers it.  Rows >= S and stage 0 (du = dv = 0 before the first
    // iteration, refine_variational.cpp:185-186) are zero rows.
    constexpr int LA = FOTG_STAGE_LA;
    int ready = stage > 0 ? 0 : S;
    const bool hi = lane + 64 < RD;
    struct Chunk { float2 v0[M], v1[M]; };
    auto issue = [&](Chunk &ck, int c) {
      if (FOTG_STAGE_DBG & 1) {
#pragma unroll
        for (int k = 0; k < M; ++k) { ck.v0[k] = make_float2(0.f, 0.f); ck.v1[k] = make_float2(0.f, 0.f); }
        return;
      }
      const int need = (c * M + M < S) ? c * M + M : S;
      if (ready < need) {
        int spins = 0;
        do {
          ready = __hip_atomic_load(prog_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ready >= need) break;
          __builtin_amdgcn_s_sleep(1);
        } while (++spins < (1 << 21));
        if (ready < need) {                                       // bounded wait: report and go on (the result is wrong, nothing hangs)
          if (lane == 0) atomicAdd(&g.sync[1], 1);
          ready = S;
        }
      }
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const int d = c * M + k;
        ck.v0[k] = make_float2(0.f, 0.f); ck.v1[k] = make_float2(0.f, 0.f);
        if (stage > 0 && d < S) {
          const float2 *row = in_buf + (size_t)d * RD;
          if (lane < RD) ck.v0[k] = ld_agent_f2(row + lane);
          if (hi) ck.v1[k] = ld_agent_f2(row + lane + 64);
        }
      }
    };
    auto write = [&](const Chunk &ck, int c) {
      unsigned slot = (unsigned)((c * M) % RDN) * DB;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        if (lane < RD) *reinterpret_cast<float2 *>(lds + DBASE + slot + lane * 8) = ck.v0[k];
        if (hi) *reinterpret_cast<float2 *>(lds + DBASE + slot + (lane + 64) * 8) = ck.v1[k];
        slot += DB; if (slot == DRING) slot = 0;
      }
    };
    static_assert(LA == 3, "the loop below rotates four chunk buffers");
    Chunk b0, b1, b2, b3;
    issue(b0, 0); issue(b1, 1); issue(b2, 2);
    for (int I = 0; I < NI; I += 4) {
      FOTG_BAR(); FOTG_STAMP_AT(I); issue(b3, I + 3); write(b0, I);
      if (I + 1 < NI) { FOTG_BAR(); issue(b0, I + 4); write(b1, I + 1); }
      if (I + 2 < NI) { FOTG_BAR(); issue(b1, I + 5); write(b2, I + 2); }
      if (I + 3 < NI) { FOTG_BAR(); issue(b2, I + 6); write(b3, I + 3); }
    }
    FOTG_STAMP(6);
    return;
  }

  // ================================================= writer =================================================
  if (wv == 7) {
    const bool hi = lane + 64 < RD;
    const float *wxq = a.single(pair, P_WX), *wyq = a.single(pair, P_WY);
    float *fl = g.flow + (size_t)pair * g.flow_stride;
    float2 *Dtap = a.taps ? a.Dp(pair) : nullptr;
    // Not the last stage: per interval exactly NST * M row stores + 1 progress store (rows >= S go to the spare rows behind the
    // buffer, the progress word is re-stored when there is nothing new), so "all but the newest NST * M + 1 operations have
    // completed" (s_waitcnt vmcnt) means: the rows stored two intervals ago are in memory -- their progress is published now,
    // while the stores of the previous interval are still in flight.  A row is RD cells: one store instruction for cells 0..63
    // (RD < 64: lanes >= RD repeat cells 0 ..) and one for cells 64 .. RD-1 (the lanes beyond repeat them: same bytes).
    constexpr int NST = RD > 64 ? 2 : 1;
    const int cell0 = RD >= 64 ? lane : lane % RD, cell1 = RD > 64 ? 64 + lane % (RD > 64 ? RD - 64 : 1) : 0;
    int pub1 = 0, pub2 = 0;                                       // progress after the stores of the previous interval / of the one before
    float2 wq_n[M][2] = {};                                       // last stage: (wx, wy) of the next interval's cells
    for (int I = 0; I < NI; ++I) {
      FOTG_BAR();
      FOTG_STAMP_AT(I);
      const int c = I - q.tW;
      if (!last) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST * M + 1) : "memory");
        __hip_atomic_store(prog_out, pub2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pub2 = pub1;
        if (c < 0 || (FOTG_STAGE_DBG & 8)) continue;
        unsigned slot = (unsigned)((c * M) % RDN) * DB;
#pragma unroll
        for (int k = 0; k < M; ++k) {
          const int d = c * M + k;
          const float2 x0 = ld_f2(DBASE + slot + cell0 * 8);
          float2 x1 = x0;
          if constexpr (NST == 2) x1 = ld_f2(DBASE + slot + cell1 * 8);
          slot += DB; if (slot == DRING) slot = 0;
          float2 *row = out_buf + (size_t)(d < S ? d : S + k) * RD;          // rows S .. S+M-1: spare
          st_agent_f2(row + cell0, x0);
          if constexpr (NST == 2) st_agent_f2(row + cell1, x1);
        }
        pub1 = (c * M + M < S) ? c * M + M : S;
        continue;
      }
      // last stage: flow = (wx + du, wy + dv) (refine_variational.cpp:208-221); cell (diagonal d, row r) is pixel (d - r, r).
      // wx, wy of the NEXT interval's chunk are loaded now (a load -> store chain per interval would expose an L2 round trip).
      float2 wc[M][2];
#pragma unroll
      for (int k = 0; k < M; ++k)
#pragma unroll
        for (int half = 0; half < 2; ++half) wc[k][half] = wq_n[k][half];
      {
        const int cn = c + 1;
#pragma unroll
        for (int k = 0; k < M; ++k)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int dd = cn * M + k, r = lane + 64 * half, i = dd - r;
            const bool ok = cn >= 0 && r < h && i >= 0 && i < w;
            const int qi = ok ? dd * RQ + r : 0;
            wq_n[k][half] = make_float2(wxq[qi], wyq[qi]);
          }
      }
      if (c < 0 || c >= q.NCW) continue;
      unsigned slot = (unsigned)((c * M) % RDN) * DB;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const int d = c * M + k;
        const float2 x0 = lane < RD ? ld_f2(DBASE + slot + lane * 8) : make_float2(0.f, 0.f);
        const float2 x1 = hi ? ld_f2(DBASE + slot + (lane + 64) * 8) : make_float2(0.f, 0.f);
        slot += DB; if (slot == DRING) slot = 0;
        if (d >= S) continue;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int r = lane + 64 * half, i = d - r;
          const float2 x = half ? x1 : x0;
          if (r < h && i >= 0 && i < w)
            *reinterpret_cast<float2 *>(fl + 2 * (size_t)(r * w + i)) = make_float2(wc[k][half].x + x.x, wc[k][half].y + x.y);
          if (Dtap && r < a.RPD && r < RD) Dtap[(size_t)d * a.RPD + r] = x;
        }
      }
    }
    if (!last) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(prog_out, pub1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    FOTG_STAMP(6);
    return;
  }

  // ================================================= data waves =================================================
  {
    int dw;                                                       // waves 4,5,6, 8,9,10, 12,13,14, 11, 15 -> 0 .. 10
    if ((wv & 3) != 3) dw = (wv / 4 - 1) * 3 + (wv & 3); else dw = wv == 11 ? 9 : 10;
    const int NBW = q.NBW, HR = q.HR;
    const int grp = dw / NBW, kk = dw - grp * NBW;
    if (grp >= NG) { for (int I = 0; I < NI; ++I) FOTG_BAR(); return; }
    // this lane's cell of a chunk: (diagonal d of the chunk, row r), the same in every pass
    const int e = kk * 64 + lane;
    const bool cellok = e < M * HR;
    const int d = cellok ? e / HR : 0, r = cellok ? e - d * HR : 0;
    // first pass: chunk c = -3 (only A'(-1), i.e. the smoothness weight of diagonal 0), groups alternate
    int c = NG == 2 ? (grp == 1 ? -3 : -2) : -3;
    int nbar = 0;
    for (; nbar < c + TB; ++nbar) FOTG_BAR();
    const float *Qm = a.single(pair, P_MASK), *Qwx = a.single(pair, P_WX), *Qwy = a.single(pair, P_WY);
    // Everything a pass reads from global memory (the constant planes of its cells) is loaded ONE PASS AHEAD: an L2 round trip
    // under the CU's own traffic is as long as the arithmetic of a pass.
    struct AW { float wxc, wxl, wxr, wxt, wxb, wyc, wyl, wyr, wyt, wyb; };
    auto load_b = [&](int cc, PixIn<NOC> &pin) {                  // the cell (4 cc + d, r) of B(cc)
      const int sD = cc * M + d, iB = sD - r;
      const bool inB = cellok && cc >= 0 && r < h && iB >= 0 && iB < w;
      const int qB = inB ? sD * RQ + r : 0;
      const int ql = (inB && iB > 0) ? qB - RQ : qB, qr = (inB && iB < w - 1) ? qB + RQ : qB;
      const int qt = (inB && r > 0) ? qB - RQ - 1 : qB, qb = (inB && r < h - 1) ? qB + RQ + 1 : qB;
#pragma unroll
      for (int ch = 0; ch < NOC; ++ch) {
        pin.Ix[ch] = a.color(pair, C_IX, ch)[qB]; pin.Iy[ch] = a.color(pair, C_IY, ch)[qB]; pin.Iz[ch] = a.color(pair, C_IZ, ch)[qB];
        pin.Ixx[ch] = a.color(pair, C_IXX, ch)[qB]; pin.Ixy[ch] = a.color(pair, C_IXY, ch)[qB]; pin.Iyy[ch] = a.color(pair, C_IYY, ch)[qB];
        pin.Ixz[ch] = a.color(pair, C_IXZ, ch)[qB]; pin.Iyz[ch] = a.color(pair, C_IYZ, ch)[qB];
      }
      pin.m = Qm[qB];
      pin.wxc = Qwx[qB]; pin.wxl = Qwx[ql]; pin.wxr = Qwx[qr]; pin.wxt = Qwx[qt]; pin.wxb = Qwx[qb];
      pin.wyc = Qwy[qB]; pin.wyl = Qwy[ql]; pin.wyr = Qwy[qr]; pin.wyt = Qwy[qt]; pin.wyb = Qwy[qb];
    };
    auto load_a = [&](int cc, AW &aw) {                           // the cell (4 (cc+2) + 1 + d, r) of A'(cc + 2)
      const int sA = (cc + 2) * M + 1 + d, iA = sA - r;
      const bool inA = cellok && sA >= 0 && r < h && iA >= 0 && iA < w;
      const int qA = inA ? sA * RQ + r : 0;
      const int ql = (inA && iA > 0) ? qA - RQ : qA, qr = (inA && iA < w - 1) ? qA + RQ : qA;
      const int qt = (inA && r > 0) ? qA - RQ - 1 : qA, qb = (inA && r < h - 1) ? qA + RQ + 1 : qA;
      aw.wxc = Qwx[qA]; aw.wxl = Qwx[ql]; aw.wxr = Qwx[qr]; aw.wxt = Qwx[qt]; aw.wxb = Qwx[qb];
      aw.wyc = Qwy[qA]; aw.wyl = Qwy[ql]; aw.wyr = Qwy[qr]; aw.wyt = Qwy[qt]; aw.wyb = Qwy[qb];
    };
    constexpr bool PFB = NOC == 1;                                // (RGB: 35 values per cell -- a second set does not fit the registers)
    PixIn<NOC> pin_n = {};
    AW aw_n = {};
    if (PFB && !(FOTG_STAGE_DBG & 2)) load_b(c, pin_n);
    if (!(FOTG_STAGE_DBG & 4)) load_a(c, aw_n);
    for (; c + TB + NG - 1 < NI; c += NG) {
      FOTG_BAR(); ++nbar;                                         // barrier #(c + TB)
      FOTG_STAMP_AT(c + TB); FOTG_STAMP_AT(c + TB - 1);
      PixIn<NOC> pin = pin_n;
      const AW aw = aw_n;
      if (!(FOTG_STAGE_DBG & 2)) { if constexpr (PFB) load_b(c + NG, pin_n); else load_b(c, pin); }
      if (!(FOTG_STAGE_DBG & 4)) load_a(c + NG, aw_n);
      const int sD = c * M + d, iB = sD - r;
      const bool inB = cellok && c >= 0 && r < h && iB >= 0 && iB < w;          // (sD < S follows)
      const bool doB = cellok && c >= 0 && c < q.NCB;
      const bool fl_ = inB && iB > 0, fr_ = inB && iB < w - 1, ft_ = inB && r > 0, fb_ = inB && r < h - 1;
      // ---------- A'(c + 2): smoothness weight (compute_smoothness first half, opticalflow_aux.c:126-139) of diagonal 4(c+2)+1+d
      if (!(FOTG_STAGE_DBG & 4)) {
        const int sA = (c + 2) * M + 1 + d, iA = sA - r;
        const bool inA = cellok && sA >= 0 && r < h && iA >= 0 && iA < w;
        const bool al = inA && iA > 0, ar = inA && iA < w - 1, at = inA && r > 0, ab = inA && r < h - 1;
        // (du,dv) of the five cells from the D ring (replicate at the image border like the 3-tap filters, image.c:436-464)
        const int sAc = inA ? sA : 0;
        const unsigned s0 = (unsigned)(sAc % RDN), sm = s0 == 0 ? RDN - 1 : s0 - 1, sp = s0 + 1 == (unsigned)RDN ? 0 : s0 + 1;
        const unsigned oc = DBASE + s0 * DB + r * 8;
        const unsigned ol = al ? DBASE + sm * DB + r * 8 : oc, orr = ar ? DBASE + sp * DB + r * 8 : oc;
        const unsigned ot = at ? DBASE + sm * DB + (r - 1) * 8 : oc, ob = ab ? DBASE + sp * DB + (r + 1) * 8 : oc;
        const float2 dc = ld_f2(oc), dl = ld_f2(ol), dr = ld_f2(orr), dt = ld_f2(ot), db = ld_f2(ob);
        const int jj = at ? (ab ? 1 : h - 1) : 0;                  // smooth_w only tests j == 0 / j == h-1
        const float sval = smooth_w(make_float2(aw.wxl + dl.x, aw.wyl + dl.y), make_float2(aw.wxc + dc.x, aw.wyc + dc.y), make_float2(aw.wxr + dr.x, aw.wyr + dr.y),
                                    make_float2(aw.wxt + dt.x, aw.wyt + dt.y), make_float2(aw.wxb + db.x, aw.wyb + db.y), jj, h, g.qa);
        if (inA) *reinterpret_cast<float *>(lds + SBASE + (unsigned)(sA % NS) * SB + r * 4) = sval;
      }
      if (NG == 2) { FOTG_BAR(); ++nbar; }                        // barrier #(c + TB + 1)
      // ---------- B(c): pair sums of the smoothness weights (:141-163), data term, laplacian, block inverse -> C ring
      if (doB) {
        float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
        {
          const int sDc = inB ? sD : 1;
          const unsigned n0 = (unsigned)(sDc % NS), nm = n0 == 0 ? NS - 1 : n0 - 1, np = n0 + 1 == NS ? 0 : n0 + 1;
          const float s_o = ld_f1(SBASE + n0 * SB + r * 4);
          const float s_r = ld_f1(SBASE + np * SB + r * 4), s_l = ld_f1(SBASE + nm * SB + r * 4);
          const float s_b = ld_f1(SBASE + np * SB + (r + 1) * 4), s_t = ld_f1(SBASE + nm * SB + (r > 0 ? r - 1 : 0) * 4);
          const float hr = fr_ ? s_o + s_r : 0.0f, hl = fl_ ? s_l + s_o : 0.0f, vb = fb_ ? s_o + s_b : 0.0f, vt = ft_ ? s_t + s_o : 0.0f;
          const float2 duv = ld_f2(DBASE + (unsigned)(sDc % RDN) * DB + r * 8);
          // data_term_cell only tests i > 0, i < w-1, j > 0, j < h-1: hand it border-equivalent coordinates
          const int ii = fl_ ? (fr_ ? 1 : w - 1) : 0, jj = ft_ ? (fb_ ? 1 : h - 1) : 0;
          float4 x0, x1;
          if (!(FOTG_STAGE_DBG & 2)) {
            data_term_cell<NOC>(a, ii, jj, pin, hr, hl, vb, vt, duv.x, duv.y, g.hd, g.hg, x0, x1);
            if (inB) { c0 = x0; c1 = x1; }                         // cells outside the image are zero (a fixed point of the update)
          }
        }
        const unsigned co = CBASE + (unsigned)(((sD % RCN) + RCN) % RCN) * CSLOT + r * 16;
        *reinterpret_cast<float4 *>(lds + co) = c0;
        *reinterpret_cast<float4 *>(lds + co + CB) = c1;
        if (last && a.taps && inB) {                                // test taps: the last system in the global skewed array
          float4 *Cg = a.Cp(pair) + a.cidx(iB, r);
          Cg[0] = c0; Cg[1] = c1;
        }
      }
    }
    for (; nbar < NI; ++nbar) FOTG_BAR();
    FOTG_STAMP(6);
  }
#undef FOTG_BAR
}

}  // namespace fotg
