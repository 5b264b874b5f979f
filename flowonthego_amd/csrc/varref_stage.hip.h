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
  const void *zero;       // >= 2 KB of zeros (rows that do not exist: stage 0's input, diagonals >= S)
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
#ifndef FOTG_STAGE_LA
#define FOTG_STAGE_LA 2
#endif
#ifndef FOTG_STAGE_AUX
#define FOTG_STAGE_AUX 16       // cache policy of the hand-over row loads: sc1
#endif
// intervals between the issue of a chunk's loads (direct to LDS) and the interval it has landed in
#define FOTG_STAGE_RCN 32      // C ring slots (diagonals)
#define FOTG_STAGE_NS 24       // smoothness ring slots
#define FOTG_STAGE_RWN 16      // (wx,wy) ring slots
#define FOTG_STAGE_RUN 16      // (uu,vv) = (wx+du, wy+dv) ring slots
#define FOTG_STAGE_NDW 10      // data waves
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
  q.RDN = M * (q.tW + FOTG_STAGE_LA + 1);
  q.NCW = (S + M - 1) / M;
  q.NCB = (E + 1 + M) / M;
  q.NI = q.T0 + E / M + q.omax / M + 4;
  return q;
}
template <int RD, int RCW>
__host__ __device__ inline int stage_lds_bytes(const StageGeom &q)
{
  return 128 + FOTG_STAGE_RCN * 2 * RCW * 16 + (q.RDN + FOTG_STAGE_RWN + FOTG_STAGE_RUN) * RD * 8 + FOTG_STAGE_NS * RD * 4 + RD * 8;
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

#ifndef FOTG_STAGE_WPE
#define FOTG_STAGE_WPE 4
#endif
// (a 1024-thread workgroup is 4 waves per SIMD whatever the register count: tell the scheduler, so it spends registers on
// instruction-level parallelism instead of saving them for an occupancy this kernel cannot have)
template <int NOC, int RD, int RCW>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(FOTG_STAGE_WPE, FOTG_STAGE_WPE))) void vr_stage_kernel(VrArgs a, StageArgs g)
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
  constexpr int RWN = FOTG_STAGE_RWN, RUN = FOTG_STAGE_RUN;
  const unsigned CBASE = 128, DBASE = CBASE + CRING, WBASE = DBASE + DRING, UBASE = WBASE + RWN * DB, SBASE = UBASE + RUN * DB, DUMP = SBASE + NS * SB;
  char *lds = lds_bytes();
  auto ld_f2 = [&](unsigned off) { return *reinterpret_cast<const float2 *>(lds + off); };
  auto ld_f4 = [&](unsigned off) { return *reinterpret_cast<const float4 *>(lds + off); };
  auto ld_f1 = [&](unsigned off) { return *reinterpret_cast<const float *>(lds + off); };
#if defined(FOTG_STAGE_STAMPS) && FOTG_STAGE_STAMPS == 2
  // arrival / release time of this wave at barriers #FOTG_STAGE_B0 .. +3 (slots 2k, 2k+1)
#ifndef FOTG_STAGE_B0
#define FOTG_STAGE_B0 30
#endif
  int stamp_ticket = 0, bc = 0;
#define FOTG_STAMPB(k) do { if (g.stamps && lane == 0) g.stamps[((size_t)stamp_ticket * 16 + wv) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define FOTG_BAR() do { const bool st_ = bc >= FOTG_STAGE_B0 && bc < FOTG_STAGE_B0 + 4; if (st_) FOTG_STAMPB(2 * (bc - FOTG_STAGE_B0)); \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); if (st_) FOTG_STAMPB(2 * (bc - FOTG_STAGE_B0) + 1); ++bc; } while (0)
#define FOTG_STAMP(slot) do { } while (0)
#define FOTG_STAMP_AT(I) do { } while (0)
#else
#define FOTG_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef FOTG_STAGE_STAMPS
  int stamp_ticket = 0;
#define FOTG_STAMP(slot) do { if (g.stamps && lane == 0) g.stamps[((size_t)stamp_ticket * 16 + wv) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define FOTG_STAMP_AT(I) do { if ((I) == 8) FOTG_STAMP(1); else if ((I) == 24) FOTG_STAMP(2); else if ((I) == 40) FOTG_STAMP(3); else if ((I) == 56) FOTG_STAMP(4); } while (0)
#else
#define FOTG_STAMP(slot) do { } while (0)
#define FOTG_STAMP_AT(I) do { } while (0)
#endif
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
  // Roles by wave id.  Waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically: class = wave & 3).
  // Measured SIMD time per interval on full diagonals: a data wave's half pass 0.45 us, a solver wave 0.25 us, uv / zero 0.25,
  // writer 0.15, loader 0.05 -- dealt so that every class carries about 1.4 us and waves of both data groups:
  //   class 0: 0 solver (sweep 0), 4 solver (sweep 1), 8 data, 12 data      class 1: 1 solver (sweep 2), 5 uv / zero, 9 data, 13 data
  //   class 2: 2 writer, 6 data, 10 data, 14 data                           class 3: 3 loader, 7 data, 11 data, 15 data
  constexpr int WV_WRITER = 2, WV_LOADER = 3, WV_UV = 5;
  if (wv == 0 || wv == 4 || wv == 1) {
    const int sweep = wv == 0 ? 0 : wv == 4 ? 1 : 2;
    if (sweep >= a.nsweeps || (FOTG_STAGE_DBG & 16)) { for (int I = 0; I < NI; ++I) FOTG_BAR(); return; }
    const int off = T0 * M + sweep * SD;
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
  if (wv == WV_LOADER) {
    // Chunk c (diagonals 4c .. 4c+3): its rows of the predecessor's (du,dv) and of (wx,wy) are loaded DIRECTLY INTO the D and w
    // rings (global_load_lds_dwordx4: 16 bytes per lane, no registers; the hand-over rows agent-scope = sc1), issued in interval
    // c - LA, as soon as the predecessor's progress word covers them, and waited for with a counted vmcnt in interval c: they
    // have landed before barrier #c+1.  Exactly 2M loads per interval (rows that do not exist -- stage 0's input: du = dv = 0
    // before the first iteration, refine_variational.cpp:185-186; diagonals >= S -- come from a zero buffer), so "all but the
    // newest 2M * LA" is chunk c.  Nothing else in this wave touches vector memory except the poll, which drains the counter.
    constexpr int LA = FOTG_STAGE_LA;
    int ready = stage > 0 ? 0 : S;
    const bool act = lane < RD / 2;
    const char *zrow = reinterpret_cast<const char *>(g.zero) + lane * 16;
    const char *inb = reinterpret_cast<const char *>(in_buf) + lane * 16;
    const char *wq = reinterpret_cast<const char *>(a.extra(pair, 8)) + lane * 16;
    auto issue = [&](int c) {
      if (!(FOTG_STAGE_DBG & 1) && stage > 0) {
        const int need = (c * M + M < S) ? c * M + M : S;
        if (ready < need) {
          int spins = 0;
          do {
            int v;
            asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(prog_in) : "memory");
            ready = __builtin_amdgcn_readfirstlane(v);
            if (ready >= need) break;
            __builtin_amdgcn_s_sleep(1);
          } while (++spins < (1 << 21));
          if (ready < need) {                                     // bounded wait: report and go on (the result is wrong, nothing hangs)
            if (lane == 0) atomicAdd(&g.sync[1], 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ready = S;
          }
        }
      }
      unsigned dslot = (unsigned)((c * M) % RDN) * DB, wslot = (unsigned)((c * M) % RWN) * DB;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const int row = c * M + k;
        const char *sd = (stage > 0 && row < S && !(FOTG_STAGE_DBG & 1)) ? inb + (size_t)row * DB : zrow;
        const char *sw = row < S ? wq + (size_t)row * (RQ * 8) : zrow;
        if (act) {
          typedef __attribute__((address_space(1))) const void gvoid;
          typedef __attribute__((address_space(3))) void lvoid;
          __builtin_amdgcn_global_load_lds((gvoid *)sd, (lvoid *)(lds + DBASE + dslot), 16, 0, FOTG_STAGE_AUX);      // aux 16 = sc1
          __builtin_amdgcn_global_load_lds((gvoid *)sw, (lvoid *)(lds + WBASE + wslot), 16, 0, 0);
        }
        dslot += DB; if (dslot == DRING) dslot = 0;
        wslot += DB; if (wslot == RWN * DB) wslot = 0;
      }
    };
    for (int c = 0; c < LA; ++c) issue(c);
    for (int I = 0; I < NI; ++I) {
      FOTG_BAR();
      FOTG_STAMP_AT(I);
      issue(I + LA);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * M * LA) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // nothing may land after the workgroup's LDS is released
    FOTG_STAMP(6);
    return;
  }

  // ================================================= writer =================================================
  if (wv == WV_WRITER) {
    const bool hi = lane + 64 < RD;
    const float2 *wq2 = reinterpret_cast<const float2 *>(a.extra(pair, 8));
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
            wq_n[k][half] = wq2[qi];
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

  // ================================================= uv / zero wave =================================================
  if (wv == WV_UV) {
    // interval I: (uu,vv) = (wx + du, wy + dv) (refine_variational.cpp:208-214) of chunk I-1, whose rows have landed in the D and w
    // rings before barrier #I, into the uv ring (read by the smoothness weights from interval I+1 on).
    // It also zeroes the C-ring cells of chunk I - TB that lie OUTSIDE the image (rows < max(0, s-w+1) or > min(h-1, s) of
    // diagonal s, up to the solver's HR rows): the data waves write the cells inside in the same intervals, the previous
    // occupant of the slots was last read in interval I - 1, the first sweep reads them from I + NG on.  A zero system cell
    // keeps a zero (du,dv) cell zero (fixed point), which is what the solver's predicate-free steps rely on.
    // (all LDS reads of a chunk are issued before the first use: branch-free, lanes beyond the row repeat its last cell)
    const unsigned cell0 = (unsigned)(lane < RD ? lane : RD - 1) * 8, cell1 = (unsigned)(lane + 64 < RD ? lane + 64 : RD - 1) * 8;
    auto uv = [&](int c) {
      if (c < 0) return;
      unsigned dslot = (unsigned)((c * M) % RDN) * DB, wslot = (unsigned)((c * M) % RWN) * DB, uslot = (unsigned)((c * M) % RUN) * DB;
      float2 dv_[M][2], wv_[M][2];
      unsigned us_[M];
#pragma unroll
      for (int k = 0; k < M; ++k) {
        dv_[k][0] = ld_f2(DBASE + dslot + cell0); wv_[k][0] = ld_f2(WBASE + wslot + cell0);
        if constexpr (RD > 64) { dv_[k][1] = ld_f2(DBASE + dslot + cell1); wv_[k][1] = ld_f2(WBASE + wslot + cell1); }
        us_[k] = uslot;
        dslot += DB; dslot = dslot == DRING ? 0 : dslot;
        wslot += DB; wslot = wslot == RWN * DB ? 0 : wslot;
        uslot += DB; uslot = uslot == RUN * DB ? 0 : uslot;
      }
#pragma unroll
      for (int k = 0; k < M; ++k) {
        *reinterpret_cast<float2 *>(lds + UBASE + us_[k] + cell0) = make_float2(wv_[k][0].x + dv_[k][0].x, wv_[k][0].y + dv_[k][0].y);
        if constexpr (RD > 64)
          *reinterpret_cast<float2 *>(lds + UBASE + us_[k] + cell1) = make_float2(wv_[k][1].x + dv_[k][1].x, wv_[k][1].y + dv_[k][1].y);
      }
    };
    auto zero_c = [&](int c) {
      if (c < 0 || c >= q.NCB) return;
      // (wave-uniform early out: chunks whose four diagonals are full rows -- the middle of the level -- have nothing to zero)
      const int s_first = c * M, s_last = c * M + M - 1;
      if (s_first >= h - 1 && s_last <= w - 1 && q.HR == h) return;
      const unsigned cb0 = CBASE + (unsigned)((c * M) % RCN) * CSLOT;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const int sd = c * M + k;
        const int lo = sd - (w - 1) > 0 ? sd - (w - 1) : 0, hi2 = sd < h - 1 ? sd : h - 1;
#pragma unroll
        for (int half = 0; half < (RCW > 64 ? 2 : 1); ++half) {
          const int rr = lane + 64 * half;
          if (rr < q.HR && (rr < lo || rr > hi2)) {
            *reinterpret_cast<float4 *>(lds + cb0 + k * CSLOT + rr * 16) = z;
            *reinterpret_cast<float4 *>(lds + cb0 + k * CSLOT + CB + rr * 16) = z;
          }
        }
      }
    };
    for (int I = 0; I < NI; ++I) {
      FOTG_BAR();
      uv(I - 1);
      zero_c(I - TB);
    }
    FOTG_STAMP(6);
    return;
  }

  // ================================================= data waves =================================================
  {
    // data waves 8, 9, 6, 7, 10 (group 0) and 12, 13, 14, 11, 15 (group 1)
    int dw;
    switch (wv) { case 8: dw = 0; break; case 9: dw = 1; break; case 6: dw = 2; break; case 7: dw = 3; break; case 10: dw = 4; break;
                  case 12: dw = 5; break; case 13: dw = 6; break; case 14: dw = 7; break; case 11: dw = 8; break; default: dw = 9; break; }
    const int NBW = q.NBW, HR = q.HR;
    const int grp = dw / NBW, kk = dw - grp * NBW;
    if (grp >= NG) { for (int I = 0; I < NI; ++I) FOTG_BAR(); return; }
    // The cells of a chunk: only the pixels INSIDE the image are enumerated (a diagonal s holds rows max(0, s-w+1) .. min(h-1, s);
    // the skewed bounding parallelogram is 1.5-2x the image), densely over the lanes of the group's waves: lane slot e = 64 kk + lane
    // -> (diagonal d of the chunk, row r) through the prefix sums of the four diagonals' lengths (wave-uniform).  Waves whose
    // slots lie beyond the chunk's cell count skip the pass (they only keep the barriers).  The cells outside the image are
    // zeroed in the C ring by wave 15.  All rings are a whole number of chunks long, so a chunk's M slots are contiguous.
    const int e = kk * 64 + lane;
    struct Cell { int d, r, i; bool ok; };
    auto cell_of = [&](int cc, int &ntot) {
      int off[M + 1], rmin[M];
      off[0] = 0;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const int sd = cc * M + k;
        const int lo = sd - (w - 1) > 0 ? sd - (w - 1) : 0, hi = sd < h - 1 ? sd : h - 1;
        rmin[k] = lo;
        off[k + 1] = off[k] + ((sd >= 0 && hi >= lo) ? hi - lo + 1 : 0);
      }
      ntot = off[M];
      Cell cl;
      cl.d = (e >= off[1]) + (e >= off[2]) + (e >= off[3]);
      const int o = cl.d == 0 ? off[0] : cl.d == 1 ? off[1] : cl.d == 2 ? off[2] : off[3];
      const int rm = cl.d == 0 ? rmin[0] : cl.d == 1 ? rmin[1] : cl.d == 2 ? rmin[2] : rmin[3];
      cl.ok = e < off[M];
      cl.r = cl.ok ? rm + e - o : 0;
      if (!cl.ok) cl.d = 0;
      cl.i = cc * M + cl.d - cl.r;
      return cl;
    };
    // addresses of a cell and of its (s-1) / (s+1) neighbours in a ring of pitch `pb` given the bases of chunks x-1, x, x+1
    auto nb3 = [&](const Cell &cl, unsigned bm, unsigned b0, unsigned bp, unsigned pb, unsigned rb, unsigned &ac, unsigned &am, unsigned &ap) {
      ac = b0 + (unsigned)cl.d * pb + rb;
      am = cl.d > 0 ? ac - pb : bm + (M - 1) * pb + rb;
      ap = cl.d < M - 1 ? ac + pb : bp + rb;
    };
    auto wrapm = [](int x, int n) { return x < 0 ? x + n : (x >= n ? x - n : x); };
    constexpr int NU = RUN / M, NSC = NS / M, NCC = RCN / M;
    const int ND = RDN / M;
    // first pass: chunk c = -2 (only A(0): the smoothness weights of chunk 0), groups alternate
    int c = NG == 2 ? (grp == 0 ? -2 : -1) : -2;
    int kU = wrapm(c % NU + NU, NU), kS = wrapm(c % NSC + NSC, NSC), kC = wrapm(c % NCC + NCC, NCC), kD = wrapm(c + ND, ND);
    int nbar = 0;
    for (; nbar < c + TB; ++nbar) FOTG_BAR();
    const float *Qm = a.single(pair, P_MASK);
    // the planes of a pass's cell are loaded ONE PASS AHEAD: an L2 round trip under the CU's own traffic is as long as a pass
    typedef PixDiff<NOC> Planes;
    auto load_b = [&](int cc, const Cell &cl, Planes &pl) {       // this lane's cell of B(cc)
      const unsigned qB = cl.ok ? (unsigned)((cc * M + cl.d) * RQ + cl.r) : 0u;
#pragma unroll
      for (int ch = 0; ch < NOC; ++ch) {
        pl.Ix[ch] = a.color(pair, C_IX, ch)[qB]; pl.Iy[ch] = a.color(pair, C_IY, ch)[qB]; pl.Iz[ch] = a.color(pair, C_IZ, ch)[qB];
        pl.Ixx[ch] = a.color(pair, C_IXX, ch)[qB]; pl.Ixy[ch] = a.color(pair, C_IXY, ch)[qB]; pl.Iyy[ch] = a.color(pair, C_IYY, ch)[qB];
        pl.Ixz[ch] = a.color(pair, C_IXZ, ch)[qB]; pl.Iyz[ch] = a.color(pair, C_IYZ, ch)[qB];
      }
      pl.m = Qm[qB];
#pragma unroll
      for (int k = 0; k < 8; ++k) pl.d[k] = a.extra(pair, k)[qB];
    };
    constexpr bool PFB = NOC == 1;                                // (RGB: 33 values per cell -- a second set does not fit the registers)
    Planes pl_n = {};
    // cell maps: of this pass's B chunk (c), and of chunk c+2 = this pass's A chunk; with NG = 2 the latter is also the next
    // pass's B chunk, so one map is computed per pass
    int nB, nA;
    Cell cb = cell_of(c, nB), ca = cell_of(c + 2, nA);
    if (PFB && !(FOTG_STAGE_DBG & 2)) load_b(c, cb, pl_n);
    for (; c + TB + NG - 1 < NI; c += NG) {
      FOTG_BAR(); ++nbar;                                         // barrier #(c + TB)
      FOTG_STAMP_AT(c + TB); FOTG_STAMP_AT(c + TB - 1);
      Planes pl = pl_n;
      if (!(FOTG_STAGE_DBG & 2)) {
        if constexpr (PFB) {
          if (NG == 2) load_b(c + 2, ca, pl_n);
          else { int nn; const Cell cn = cell_of(c + 1, nn); load_b(c + 1, cn, pl_n); }
        } else load_b(c, cb, pl);
      }
      // ---------- A(c + 2): smoothness weights (compute_smoothness first half, opticalflow_aux.c:126-139) of chunk c+2 from the
      // (uu,vv) ring: cell, left (s-1, r), right (s+1, r), top (s-1, r-1), bottom (s+1, r+1); replicate at the image border
      // like the 3-tap filters (image.c:436-464)
      if (!(FOTG_STAGE_DBG & 4) && kk * 64 < nA) {                 // (wave-uniform)
        const bool inA = ca.ok;
        const bool al = inA && ca.i > 0, ar = inA && ca.i < w - 1, at = inA && ca.r > 0, ab = inA && ca.r < h - 1;
        unsigned ac, am, ap;
        nb3(ca, UBASE + (unsigned)wrapm(kU + 1, NU) * (M * DB), UBASE + (unsigned)wrapm(kU + 2, NU) * (M * DB), UBASE + (unsigned)wrapm(kU + 3, NU) * (M * DB),
            DB, (unsigned)ca.r * 8, ac, am, ap);
        const float2 uc = ld_f2(ac), ul = ld_f2(al ? am : ac), ur = ld_f2(ar ? ap : ac), ut = ld_f2(at ? am - 8 : ac), ub = ld_f2(ab ? ap + 8 : ac);
        const int jj = at ? (ab ? 1 : h - 1) : 0;                  // smooth_w only tests j == 0 / j == h-1
        const float sval = smooth_w(ul, uc, ur, ut, ub, jj, h, g.qa);
        if (inA) *reinterpret_cast<float *>(lds + SBASE + (unsigned)wrapm(kS + 2, NSC) * (M * SB) + (unsigned)ca.d * SB + ca.r * 4) = sval;
      }
      // ---------- B(c): pair sums of the smoothness weights (:141-163), data term (:310-438), sub_laplacian (:172-199), block
      // inverse (solver.c:115-120) -> C ring
      // With two intervals per pass (NG = 2) the pass's second barrier sits INSIDE the data term, between its colour and its
      // gradient part (about half of the pass's arithmetic on either side); a wave without cells just executes it.
      const bool has_b = c >= 0 && kk * 64 < nB && !(FOTG_STAGE_DBG & 2);      // (wave-uniform)
      if (!has_b && NG == 2) { FOTG_BAR(); ++nbar; }              // barrier #(c + TB + 1)
      if (has_b) {
        const bool inB = cb.ok;
        const int iB = cb.i, r = cb.r;
        const bool fl_ = inB && iB > 0, fr_ = inB && iB < w - 1, ft_ = inB && r > 0, fb_ = inB && r < h - 1;
        unsigned sc, sm, sp;
        nb3(cb, SBASE + (unsigned)wrapm(kS - 1, NSC) * (M * SB), SBASE + (unsigned)kS * (M * SB), SBASE + (unsigned)wrapm(kS + 1, NSC) * (M * SB),
            SB, (unsigned)r * 4, sc, sm, sp);
        const float s_o = ld_f1(sc), s_r = ld_f1(fr_ ? sp : sc), s_l = ld_f1(fl_ ? sm : sc), s_b = ld_f1(fb_ ? sp + 4 : sc), s_t = ld_f1(ft_ ? sm - 4 : sc);
        const float hr = fr_ ? s_o + s_r : 0.0f, hl = fl_ ? s_l + s_o : 0.0f, vb = fb_ ? s_o + s_b : 0.0f, vt = ft_ ? s_t + s_o : 0.0f;
        const float2 duv = ld_f2(DBASE + (unsigned)kD * (M * DB) + (unsigned)cb.d * DB + r * 8);
        const Planes &pin = pl;
        // data_term_cell only tests i > 0, i < w-1, j > 0, j < h-1: hand it border-equivalent coordinates
        const int ii = fl_ ? (fr_ ? 1 : w - 1) : 0, jj = ft_ ? (fb_ ? 1 : h - 1) : 0;
        float4 c0, c1;
        auto mid = [&](float &x0, float &x1, float &x2, float &x3, float &x4) {
          if (NG == 2) {
#if defined(FOTG_STAGE_STAMPS) && FOTG_STAGE_STAMPS == 2
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4));
            FOTG_BAR();
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4));
#else
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4) :: "memory");
#endif
          }
        };
        data_term_cell<NOC>(a, ii, jj, pin, hr, hl, vb, vt, duv.x, duv.y, g.hd, g.hg, c0, c1, mid);
        if (NG == 2) ++nbar;                                       // barrier #(c + TB + 1), executed inside the data term
        if (inB) {
          const unsigned co = CBASE + (unsigned)kC * (M * CSLOT) + (unsigned)cb.d * CSLOT + r * 16;
          *reinterpret_cast<float4 *>(lds + co) = c0;
          *reinterpret_cast<float4 *>(lds + co + CB) = c1;
          if (last && a.taps) {                                     // test taps: the last system in the global skewed array
            float4 *Cg = a.Cp(pair) + a.cidx(iB, r);
            Cg[0] = c0; Cg[1] = c1;
          }
        }
      }
      if (NG == 2) { cb = ca; nB = nA; } else cb = cell_of(c + 1, nB);
      ca = cell_of(c + NG + 2, nA);
      kU = wrapm(kU + NG, NU); kS = wrapm(kS + NG, NSC); kC = wrapm(kC + NG, NCC); kD = wrapm(kD + NG, ND);
    }
    for (; nbar < NI; ++nbar) FOTG_BAR();
    FOTG_STAMP(6);
  }
#undef FOTG_BAR
}

}  // namespace fotg
