// varref_resident.hip.h -- the fixed-point loop of one refinement level in ONE launch, the solver RESIDENT over the inner
// iterations and the data term on other CUs (kroeger/refine_variational.cpp:182-221: inner x { compute_smoothness,
// compute_data, sub_laplacian, sor_coupled }, FDF1.0.1/opticalflow_aux.c:123-438, solver.c:77-421).
//
// Why.  Launch-per-stage (varref.hip.h) runs a level as inner x { vr_data_kernel ; vr_sor_stream_kernel }: ten launches at
// 1080p level 4, each solver launch with its own prologue (three chunks of LDS-DMA loads before the first diagonal), and the
// data term and the sweeps of an iteration strictly one after the other although sweep 0 at diagonal s only needs the system of
// diagonals <= s+1.  Here a pair owns 1 + NDW workgroups for the whole level:
//
//   SOLVER workgroup (one CU): vr_sor_stream_kernel's machinery -- three barrier-stepped solver waves (one per sweep, two rows
//     per lane, packed f32), three loader waves (direct-to-LDS loads of the (du,dv) and system diagonals, three intervals
//     ahead), a writer wave -- running over ALL inner iterations as one sequence of E-diagonal rounds (E = S rounded up; the
//     rounds' surplus diagonals are all-zero rows, a fixed point of the update), rings never drained, no prologue between
//     iterations.  The writer hands the finished (du,dv) diagonals of iteration it to memory and publishes how many.
//   NDW DATA workgroups (other CUs), every WAVE of them an independent worker (16 NDW per pair): chunk c (M diagonals) of
//     iteration it+1 is built as soon as the solver has published the final (du,dv) of iteration it up to diagonal M c + M + 1
//     (the smoothness weights reach one diagonal further than the cell, the pair sums another one): (uu,vv) -> smoothness
//     weights -> data term + sub_laplacian + 2x2 block inverse -> the system cells of the chunk, written through to memory,
//     then the chunk is published.  A worker needs no barrier (its LDS slab is private, a wave's LDS accesses execute in
//     order), so the memory round trips of a chunk (poll, (du,dv) rows, planes, store acknowledgement: ~10 us) overlap
//     across the 16 waves of a CU.  The solver's loaders wait for a chunk before they load its diagonals.  The data term of
//     iteration it+1 thus runs WHILE the sweeps of iteration it are still on their way down the level, and the solver never
//     waits for it as long as the workers keep up.
//
// WRITE ONCE.  The per-XCD L2s are not coherent with each other: a line an XCD has read stays readable there after another
// XCD rewrote it.  So nothing that crosses workgroups is ever rewritten inside the launch: every inner iteration has its own
// system array C_it (E rows; rows S.. and the cells outside the image are zero for the life of the context) and its own (du,dv)
// array (iteration it reads array it -- array 0 is all zero -- and writes array it + 1; E rows each, rows S.. zero for good);
// each line is written once -- through to memory -- and only read after that.  (The progress words are polled with sc1 loads
// and only ever grow.)
//
// Same operations in the same order on every cell as the sequential loop: bit-identical (tests compare with ==).
//
// Hand-over (MI355X_MICROARCH.md, "inter-workgroup visibility", first row of the table of measured hand-offs): every handed-off
// byte is stored sc1 (write-through) and loaded sc1 (L1-bypassing: ld_sc1_f2, buffer loads / LDS-DMA loads with aux sc1); a
// storing workgroup drains its stores (s_waitcnt vmcnt(0), barrier) before ONE lane publishes the progress word with an sc1
// store; the wave that polled loads only after its poll matched, the other waves of its workgroup after a barrier it joins.
// Constant inputs (the level's planes, written by the set-up launch) are ordinary loads.
//
// Roles are dealt by a ticket counter, never by blockIdx: tickets (1 + NDW) p .. (1 + NDW) p + NDW are pair p's, so at any time
// at most ONE pair of a launch is partly started and every other started workgroup has all its partners running -- no deadlock
// when the grid exceeds the chip or shares it with other launches.  Every wait is bounded; a time-out is counted and raised
// to the host (ResArgs::stall_flag -> FOTG_ERR_STALL).
#pragma once
#include "varref.hip.h"

namespace fotg {

#define FOTG_RES_NDW 3          // data workgroups per pair
#define FOTG_RES_M 4            // diagonals per barrier interval of the solver workgroup
#ifndef FOTG_RES_LQ
#define FOTG_RES_LQ 1           // chunks of direct loads a loader keeps in flight (an sc1 load is a fabric round trip: microseconds under load)
#endif
#define FOTG_RES_LI (FOTG_RES_LQ + 2)   // load lead of the solver's loader waves, in intervals: chunk c is issued in interval c - LI, waited
                                        // for in interval c - LI + LQ - 1, complete before barrier c - LI + LQ, visible from interval c - 1 on
#ifndef FOTG_RES_DBG
#define FOTG_RES_DBG 0          // timing experiments in separate builds only (wrong results): 1 writer stores nothing, 2 loaders load
#endif                          // without sc1, 4 data workers publish without computing, 8 writer never waits for its stores,
                                // 16 solver waves only count barriers, 32 loaders issue no loads, 64 writer does nothing but barriers
#ifndef FOTG_RES_WQ
#define FOTG_RES_WQ 6           // intervals a write-through (du,dv) row may take to be acknowledged before its count is published
#endif
#ifndef FOTG_RES_NH
#define FOTG_RES_NH 1           // waves per loader role (1 or 2)
#endif
#define FOTG_RES_NWK_MAX (16 * FOTG_RES_NDW)   // data workers (waves) per pair at most

struct ResArgs {
  int inner;                 // inner iterations of the level
  int npairs;
  int nww;                   // worker waves per data workgroup (16 unless their LDS slabs would not fit)
  long c_it_stride;          // float4 units between the system arrays of consecutive inner iterations of a pair
  long d_it_stride;          // float2 units between their (du,dv) arrays
  float qa, hd, hg, omega;
  float *flow;               // [pair][h][w][2], written by the solver's writer wave in the last iteration
  long flow_stride;
  const void *zero;          // >= 4 KB of zeros: rows that do not exist ((du,dv) before the first iteration, diagonals >= S)
  int *sync;                 // [0] ticket counter, [1] timed-out waits; pair p (a line = 32 ints): line 1 + 3 p = finished (du,dv) diagonals,
                             // lines + 1, + 2 = chunks finished by data worker 0 .. NWK-1 (one int each)
  int *stall_flag;           // pinned host word raised on a time-out
  long long *stats;          // -DFOTG_RES_STATS builds only: pair 0's waves report [cycles waited at barriers, cycles in all, role, polls that had to spin]
};
__host__ __device__ inline long res_sync_words(int npairs) { return 32L * (1 + (long)npairs * 3); }
static_assert(FOTG_RES_NWK_MAX <= 64, "one progress int per worker in two lines, one lane per worker in the loaders' poll");

struct ResGeom { int E, omax, WO, RDN, RCN, NCH, slab, nww; };
__host__ __device__ inline ResGeom res_geom(int w, int h, int nsweeps)
{
  constexpr int M = FOTG_RES_M, U = 32, UT = 8;
  ResGeom q;
  const int S = w + h - 1;
  int E = 0;
  while (E + U <= S) E += U;
  while (E < S) E += UT;
  q.E = E;
  q.omax = (nsweeps > 0 ? nsweeps - 1 : 0) * 8;                   // sweep spacing of the barrier-stepped waves for M = 4
  q.WO = q.omax / M + 1;
  q.RDN = M * (FOTG_RES_LI + q.WO + 1);
  q.RCN = q.RDN - M;
  q.NCH = E / M;                                                  // data chunks per iteration = the loaders' chunks (surplus ones are empty)
  // a worker's LDS slab: (uu,vv) of M + 4 diagonals, (du,dv) of M, smoothness weights of M + 2, pitch RD = 72 or 100
  const int rp = (h + 1) / 2 * 2, rd = (rp + 1 <= 70 && h + 2 <= 72) ? 72 : 100;       // (the host's choice of ring geometry)
  q.slab = ((M + 4) * 8 + M * 8 + (M + 2) * 4) * rd;
  q.nww = (160 * 1024) / q.slab < 16 ? (160 * 1024) / q.slab : 16;
  return q;
}
template <int RD, int RCW>
__host__ __device__ inline int res_lds_bytes(const ResGeom &q)
{
  const int solver = q.RCN * 2 * RCW * 16 + q.RDN * RD * 8 + RD * 8 + 6 * 256;      // C ring | D ring | spare row | the loaders' poll words
  const int data = q.nww * q.slab;
  return solver > data ? solver : data;
}

__device__ __forceinline__ float2 res_ld_sc1_f2(const float2 *p)
{
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(float2, v);
}
// one 16-byte-per-lane direct load with sc1 (agent scope: served by L2 / the fabric, never by this CU's L1)
__device__ __forceinline__ void glds16_sc1(const void *src, unsigned lds_dst_byte)
{
  typedef __attribute__((address_space(1))) const void gvoid;
  typedef __attribute__((address_space(3))) void lvoid;
  __builtin_amdgcn_global_load_lds((gvoid *)src, (lvoid *)(lds_bytes() + lds_dst_byte), 16, 0, (FOTG_RES_DBG & 2) ? 0 : 16);
}

template <int NOC, int RD, int RCW>
__global__ __launch_bounds__(1024) void vr_resident_kernel(VrArgs a, ResArgs g)
{
  constexpr int M = FOTG_RES_M, LI = FOTG_RES_LI, NDW = FOTG_RES_NDW, U = 32, UT = 8, DS = 8;
  constexpr int DB = RD * 8, CB = RCW * 16, CSLOT = 2 * CB;        // bytes per ring slot: (du,dv) row, one system plane, both planes
  static_assert(RCW > 64 && RCW <= 128, "two direct loads per system plane and diagonal");
  static_assert(U % M == 0 && UT % M == 0, "barrier phase is a compile-time property of the unrolled step");
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int w = a.w, h = a.h, S = a.S, RP = a.RP;
  constexpr int RQ = RD;                                           // pitch of the skewed planes and (du,dv) arrays (host: RPD = RD)
  const ResGeom q = res_geom(w, h, a.nsweeps);
  const int E = q.E, NIT = g.inner, NCH = q.NCH;
  char *lds = lds_bytes();
#ifdef FOTG_RES_STATS
  long long st_wait = 0, st_spin = 0, st_vm = 0;
  const long long st_t0 = clock64();
#define RES_BAR_PLAIN() do { const long long b0_ = clock64(); asm volatile("s_barrier" ::: "memory"); st_wait += clock64() - b0_; } while (0)
#define RES_BAR_LGKM() do { const long long b0_ = clock64(); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); st_wait += clock64() - b0_; } while (0)
#define RES_STATS_OUT(role_) do { if (g.stats && pair == 0 && lane == 0) { long long *o_ = g.stats + ((role_) * 16 + wv) * 4; o_[0] = st_wait; o_[1] = clock64() - st_t0; o_[2] = st_vm; o_[3] = st_spin; } } while (0)
#else
#define RES_BAR_PLAIN() asm volatile("s_barrier" ::: "memory")
#define RES_BAR_LGKM() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define RES_STATS_OUT(role_) do { } while (0)
#endif
  // ---- role: ticket -> (pair, role); the 1 + NDW workgroups of a pair hold consecutive tickets
  if (threadIdx.x == 0) *reinterpret_cast<int *>(lds) = atomicAdd(&g.sync[0], 1);
  __syncthreads();
  const int ticket = *reinterpret_cast<const int *>(lds);
  __syncthreads();
  const int pair = ticket / (1 + NDW), role = ticket - pair * (1 + NDW);
  if (pair >= g.npairs) return;
  int *const sprog = g.sync + 32 * (1 + pair * 3);                // finished (du,dv) diagonals, counted over the iterations: it * S + rows
  int *const dprog = sprog + 32;                                  // + W: chunks finished by data worker W
  char *const Dg = reinterpret_cast<char *>(a.Dp(pair));          // D_it = Dg + it * d_it_bytes
  char *const Cg = reinterpret_cast<char *>(a.Cp(pair));          // C_it = Cg + it * c_it_bytes
  const size_t c_it_bytes = (size_t)g.c_it_stride * 16, d_it_bytes = (size_t)g.d_it_stride * 8;
  auto timed_out = [&]() {
    if (lane == 0) { atomicAdd(&g.sync[1], 1); __hip_atomic_store(g.stall_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  };

  // =====================================================================================================================
  // DATA workgroups: every wave is a worker W = nww (role - 1) + wave, taking chunks W, W + NWK, ... of the NIT * NCH chunks of
  // the level (chunk cg: iteration cg / NCH, diagonals M (cg % NCH) ..).  Lane slots e = lane + 64 q enumerate the cells of a
  // row group densely: cell (diagonal dr of the group, row r = e - dr h).
  // =====================================================================================================================
  if (role > 0) {
    const int NWK = g.nww * NDW;
    if (wv >= g.nww) return;
    const int W = g.nww * (role - 1) + wv;
    // The level's skewed planes and (du,dv) arrays have the pitch RD (the host sets RPD = RD for levels of this pipeline), and so
    // has the slab: a row group is one contiguous run of cells in memory and in LDS, lane slot e = lane + 64 q <-> cell
    // (diagonal e / RD of the group, row e % RD), and a cell's neighbours (s-1, r), (s+1, r), (s-1, r-1), (s+1, r+1) sit at
    // e - RD, e + RD, e - RD - 1, e + RD + 1.
    char *const slab = lds + (size_t)wv * q.slab;
    float2 *const uu = reinterpret_cast<float2 *>(slab);           // (uu,vv) = (wx + du, wy + dv), diagonals d0-2 .. d0+M+1
    float2 *const dd = uu + (M + 4) * RD;                          // (du,dv), diagonals d0 .. d0+M-1
    float *const sm = reinterpret_cast<float *>(dd + M * RD);      // smoothness weights, diagonals d0-1 .. d0+M
    // every plane of the pair's skewed workspace through ONE buffer resource: address = resource + plane offset (SGPR) + cell
    // offset (one VGPR) -- no 64-bit vector address arithmetic, no pointer per plane
    const unsigned plb = (unsigned)a.pl * 4u;                      // bytes per plane
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void *)(a.base + (size_t)pair * a.pair_stride), 0,
                                                                         (unsigned)(P_NSINGLE + C_NCOLOR * NOC + FOTG_VR_NEXTRA) * plb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsDall = __builtin_amdgcn_make_buffer_rsrc((void *)Dg, 0, (unsigned)((NIT + 1) * E + 2) * (unsigned)(RD * 8), 0x00020000);
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    auto ld_plane = [&](int plane, unsigned cell_byte) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsP, cell_byte, (unsigned)plane * plb, 0)); };
    constexpr int PL_COLOR = P_NSINGLE, PL_EXTRA = P_NSINGLE + C_NCOLOR * NOC;
    constexpr int N1 = (M + 4) * RD, N2 = (M + 2) * RD, N3 = M * RD;
    constexpr int Q1 = (N1 + 63) / 64, Q2 = (N2 + 63) / 64, Q3 = (N3 + 63) / 64;
    int done = 0, seen = 0;
    for (int cg = W; cg < NIT * NCH; cg += NWK) {
      const int it = cg / NCH, c = cg - it * NCH, d0 = c * M;
      if (d0 < S && !(FOTG_RES_DBG & 4)) {
        const unsigned dprev_off = (unsigned)it * (unsigned)d_it_bytes;                                 // (du,dv) array `it` (array 0: zeros, never loaded)
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)(Cg + (size_t)it * c_it_bytes), 0, (unsigned)(E * RP * 32), 0x00020000);
        if (it > 0) {
          // the final (du,dv) of iteration it-1 up to diagonal d0 + M + 1
          const int last = d0 + M + 2 < S ? d0 + M + 2 : S;
          const int need = (it - 1) * S + last;
          if (seen < need) {
            int spins = 0;
#ifdef FOTG_RES_STATS
            const long long b0_ = clock64();
#endif
            do {
              seen = __hip_atomic_load(sprog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (seen >= need) break;
              __builtin_amdgcn_s_sleep(8);
            } while (++spins < (1 << 20));
#ifdef FOTG_RES_STATS
            st_wait += clock64() - b0_; st_spin += spins;
#endif
            if (seen < need) { timed_out(); seen = 0x3fffffff; }
          }
        }
        // (uu,vv) of diagonals d0-2 .. d0+M+1 and (du,dv) of the chunk's own diagonals -> slab (refine_variational.cpp:208-214)
        const int gbase = (d0 - 2) * RD, gend = S * RD;
        constexpr int QB = 5;                                      // passes whose loads are in flight together
#pragma unroll
        for (int q0 = 0; q0 < Q1; q0 += QB) {
          float2 wq[QB], dv[QB];
#pragma unroll
          for (int qq = 0; qq < QB; ++qq) {
            const int e = lane + 64 * (q0 + qq), gi = gbase + e;
            const bool ok = q0 + qq < Q1 && e < N1 && gi >= 0 && gi < gend;
            const unsigned cb = ok ? (unsigned)gi * 8u : 0u;
            wq[qq] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsP, cb, (unsigned)(PL_EXTRA + 8) * plb, 0));
            dv[qq] = it > 0 ? __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsDall, cb, dprev_off, 16)) : make_float2(0.f, 0.f);   // aux 16 = sc1
            if (!ok) { wq[qq] = make_float2(0.f, 0.f); dv[qq] = make_float2(0.f, 0.f); }
          }
#pragma unroll
          for (int qq = 0; qq < QB; ++qq) {
            const int e = lane + 64 * (q0 + qq);
            if (q0 + qq < Q1 && e < N1) {
              uu[e] = make_float2(wq[qq].x + dv[qq].x, wq[qq].y + dv[qq].y);
              if (e >= 2 * RD && e < (2 + M) * RD) dd[e - 2 * RD] = dv[qq];
            }
          }
        }
        asm volatile("" ::: "memory");                             // (compiler barrier: the wave's LDS accesses execute in order)
        // compute_smoothness first half (opticalflow_aux.c:126-139) on diagonals d0-1 .. d0+M; replicate at the image border like
        // the 3-tap filters (image.c:436-464)
        for (int qq = 0; qq < Q2; ++qq) {
          const int e = lane + 64 * qq;
          const int dr = e / RD, r = e - dr * RD, srow = d0 - 1 + dr, ii = srow - r;
          if (e >= N2 || r >= h || srow < 0 || srow >= S || ii < 0 || ii >= w) continue;
          const bool al = ii > 0, ar = ii < w - 1, at = r > 0, ab = r < h - 1;
          const int uc_ = e + RD;
          const float2 uc = uu[uc_], ul = uu[al ? uc_ - RD : uc_], ur = uu[ar ? uc_ + RD : uc_], ut = uu[at ? uc_ - RD - 1 : uc_], ub = uu[ab ? uc_ + RD + 1 : uc_];
          const int jj = at ? (ab ? 1 : h - 1) : 0;                // smooth_w only tests j == 0 / j == h-1
          sm[e] = smooth_w(ul, uc, ur, ut, ub, jj, h, g.qa);
        }
        asm volatile("" ::: "memory");
        // pair sums (:141-163), data term (:310-438), sub_laplacian (:172-199), block inverse (solver.c:115-120) -> system cell
        const unsigned gcell = (unsigned)(d0 * RD);
        for (int qq = 0; qq < Q3; ++qq) {
          const int e = lane + 64 * qq;
          const int cd = e / RD, cr = e - cd * RD, s_ = d0 + cd, i = s_ - cr;
          if (e >= N3 || cr >= h || s_ >= S || i < 0 || i >= w) continue;
          PixDiff<NOC> pl;
          const unsigned qB = (gcell + (unsigned)e) * 4u;
#pragma unroll
          for (int ch = 0; ch < NOC; ++ch) {
            pl.Ix[ch] = ld_plane(PL_COLOR + C_IX * NOC + ch, qB); pl.Iy[ch] = ld_plane(PL_COLOR + C_IY * NOC + ch, qB); pl.Iz[ch] = ld_plane(PL_COLOR + C_IZ * NOC + ch, qB);
            pl.Ixx[ch] = ld_plane(PL_COLOR + C_IXX * NOC + ch, qB); pl.Ixy[ch] = ld_plane(PL_COLOR + C_IXY * NOC + ch, qB); pl.Iyy[ch] = ld_plane(PL_COLOR + C_IYY * NOC + ch, qB);
            pl.Ixz[ch] = ld_plane(PL_COLOR + C_IXZ * NOC + ch, qB); pl.Iyz[ch] = ld_plane(PL_COLOR + C_IYZ * NOC + ch, qB);
          }
          pl.m = ld_plane(P_MASK, qB);
#pragma unroll
          for (int x = 0; x < 8; ++x) pl.d[x] = ld_plane(PL_EXTRA + x, qB);
          const bool fl_ = i > 0, fr_ = i < w - 1, ft_ = cr > 0, fb_ = cr < h - 1;
          const int sc = e + RD;
          const float s_o = sm[sc], s_r = sm[fr_ ? sc + RD : sc], s_l = sm[fl_ ? sc - RD : sc], s_b = sm[fb_ ? sc + RD + 1 : sc], s_t = sm[ft_ ? sc - RD - 1 : sc];
          const float hr = fr_ ? s_o + s_r : 0.0f, hl = fl_ ? s_l + s_o : 0.0f, vb = fb_ ? s_o + s_b : 0.0f, vt = ft_ ? s_t + s_o : 0.0f;
          const float2 duv = dd[e];
          // data_term_cell only tests i > 0, i < w-1, j > 0, j < h-1: hand it border-equivalent coordinates
          const int ii = fl_ ? (fr_ ? 1 : w - 1) : 0, jj = ft_ ? (fb_ ? 1 : h - 1) : 0;
          float4 c0, c1;
          data_term_cell<NOC, NoMid, PixDiff<NOC>>(a, ii, jj, pl, hr, hl, vb, vt, duv.x, duv.y, g.hd, g.hg, c0, c1);
          const unsigned co = (unsigned)(s_ * RP + cr) * 32u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, c0), rsC, co, 0, 16);            // aux 16 = sc1
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, c1), rsC, co + 16u, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the storing wave drains, then publishes (it signals for itself only)
      }
      ++done;
      if (lane == 0) __hip_atomic_store(dprog + W, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    RES_STATS_OUT(role);
    return;
  }

  // =====================================================================================================================
  // SOLVER workgroup
  // =====================================================================================================================
  const int omax = q.omax, WO = q.WO, RDN = q.RDN, RCN = q.RCN;
  const int NI = NIT * (E / M) + omax / M + 1;                      // barrier intervals every wave goes through
  const int nsolver = a.nsweeps;
  // waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically); the solver waves are 0, 1, 2
#ifndef FOTG_RES_WAVES
#define FOTG_RES_WAVES 4, 5, 6, 3      // loaders beside the solver waves (a wave issues at most every other slot of its SIMD), writer alone
#endif
  constexpr int helper_waves[4] = {FOTG_RES_WAVES};
  constexpr int wvL0 = helper_waves[0], wvL1 = helper_waves[1], wvL2 = helper_waves[2], wvW = helper_waves[3];
  const unsigned CRING = (unsigned)RCN * CSLOT, DRING = (unsigned)RDN * DB;
  const unsigned DBASE = CRING, DUMP = CRING + DRING, POLL = DUMP + DB;   // C ring | D ring | one spare D row | 3 x 64 poll words
  auto ld_f2 = [&](unsigned off) { return *reinterpret_cast<const float2 *>(lds + off); };
  auto ld_f4 = [&](unsigned off) { return *reinterpret_cast<const float4 *>(lds + off); };
  const char *const zrow = reinterpret_cast<const char *>(g.zero);

  // ---------------- loaders ----------------
  // Interval I: barrier, issue the direct loads of chunk I + LI (M diagonals), wait until only that newest chunk is in flight.
  //   loader 0: the (du,dv) row + system plane 0 rows 0..63     loader 1: plane 1 rows 0..63 + plane 0 rows 64..RP
  //   loader 2: plane 1 rows 64..RP
  // The arrays of consecutive iterations are contiguous and every one has exactly E rows (rows S .. E-1 all zero; (du,dv) array 0
  // is all zero = the state before the first iteration), so a loader's source pointers just advance by one row per diagonal for
  // the whole level.  Before a loader issues system chunk cg, data chunk cg must be published: `frontier` = all chunks below it
  // are.  While the data workers are only a few chunks ahead, every interval issues a poll whose value is consumed behind the
  // counted wait (no extra stall); a loader that has caught up with them polls in a loop.
  auto loader = [&](auto role_tag, auto half_tag) {
    // NH > 1: a loader role is shared by NH waves, wave HALF takes the diagonals kq % NH == HALF of every chunk
    constexpr int ROLE = decltype(role_tag)::value, HALF = decltype(half_tag)::value, NH = FOTG_RES_NH;
    constexpr int NG = (M / NH) * (ROLE == 2 ? 1 : 2);
    const int NWK = g.nww * NDW, nchunks = NIT * NCH;
    const bool inD = lane < RQ / 2, inC2 = lane + 64 <= RP;
    const char *pD = Dg + lane * 16;
    const char *pC = Cg + lane * 32;
    size_t dstep = (size_t)RQ * 8, cstep = (size_t)RP * 32;
    unsigned ldslot = 0, lcslot = 0;
    int lchunk = 0, frontier = 0;
    const int *const pword = dprog + (lane < NWK ? lane : NWK - 1);
    // The progress words are polled through the LDS-DMA path too (one 4-byte direct load per lane into this loader's 64 poll
    // words, read back from LDS behind the counted wait): a load that returns into a VGPR inside this loop would make the
    // compiler drain the vector-memory counter at the loop header -- a full memory round trip per interval.
    const unsigned pollb = POLL + (unsigned)(ROLE * NH + HALF) * 256u;
    auto poll_issue = [&]() {
      typedef __attribute__((address_space(1))) const void gvoid;
      typedef __attribute__((address_space(3))) void lvoid;
      __builtin_amdgcn_global_load_lds((gvoid *)pword, (lvoid *)(lds + pollb), 4, 0, 16);
    };
    // worker W has finished chunks W, W + NWK, ..: its first unfinished one is W + NWK * count; the frontier is their minimum
    auto frontier_read = [&]() {
      const int v = *reinterpret_cast<const volatile int *>(lds + pollb + lane * 4);
      int x = lane < NWK ? lane + v * NWK : 0x3fffffff;
#pragma unroll
      for (int sh = 32; sh >= 1; sh >>= 1) { const int y = __shfl_xor(x, sh, 64); x = y < x ? y : x; }
      return __builtin_amdgcn_readfirstlane(x);
    };
    auto issue_chunk = [&]() {
      if (lchunk < nchunks) {
        if (frontier <= lchunk && !(FOTG_RES_DBG & 128)) {          // caught up with the data workers: poll in a loop
          int spins = 0;
#ifdef FOTG_RES_STATS
          const long long b0_ = clock64();
#endif
          do {
            poll_issue();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            frontier = frontier_read();
            if (frontier > lchunk) break;
            __builtin_amdgcn_s_sleep(1);
          } while (++spins < (1 << 21));
#ifdef FOTG_RES_STATS
          st_spin += clock64() - b0_;
#endif
          if (frontier <= lchunk) { timed_out(); frontier = 0x3fffffff; }
        }
      } else if (lchunk == nchunks) {                               // the chunks the schedule loads past the level's end: zero rows
        pD = zrow + lane * 16; pC = zrow + lane * 32; dstep = 0; cstep = 0;
      }
#pragma unroll
      for (int kq = 0; kq < M; ++kq) {
        if (!(FOTG_RES_DBG & 32) && kq % NH == HALF) {
        if (ROLE == 0) { if (inD) glds16_sc1(pD, DBASE + ldslot); glds16_sc1(pC, lcslot); }
        if (ROLE == 1) { glds16_sc1(pC + 16, lcslot + CB); if (inC2) glds16_sc1(pC + 64 * 32, lcslot + 1024); }
        if (ROLE == 2) { if (inC2) glds16_sc1(pC + 64 * 32 + 16, lcslot + CB + 1024); }
        }
        pD += dstep; pC += cstep;
        ldslot += DB; if (ldslot == DRING) ldslot = 0;
        lcslot += CSLOT; if (lcslot == CRING) lcslot = 0;
      }
      ++lchunk;
    };
    for (int c = 0; c < LI; ++c) issue_chunk();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int I = 0; I < NI; ++I) {
      RES_BAR_PLAIN();
      const bool poll = lchunk < nchunks && frontier < lchunk + 12 && !(FOTG_RES_DBG & 128);
      if (poll) poll_issue();                                       // lands before the chunk issued behind it: read behind the counted wait
      issue_chunk();
#ifdef FOTG_RES_STATS
      const long long v0_ = clock64();
#endif
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(FOTG_RES_LQ * NG) : "memory");
#ifdef FOTG_RES_STATS
      st_vm += clock64() - v0_;
#endif     // (a poll is older than LQ chunks only LQ intervals later: its
      if (poll) { const int f = frontier_read(); frontier = f > frontier ? f : frontier; }   // word is read then at the latest -- polls are monotone)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // nothing may land after the workgroup's LDS is released
    RES_STATS_OUT(0);
  };
  if (wv == wvL0) { loader(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); return; }
  if (wv == wvL1) { loader(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); return; }
  if (wv == wvL2) { loader(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}); return; }
  if (FOTG_RES_NH == 2) {                                           // the second wave of every loader role, on the SIMD of the first
    if (wv == wvL0 + 4) { loader(std::integral_constant<int, 0>{}, std::integral_constant<int, FOTG_RES_NH - 1>{}); return; }
    if (wv == wvL1 + 4) { loader(std::integral_constant<int, 1>{}, std::integral_constant<int, FOTG_RES_NH - 1>{}); return; }
    if (wv == wvL2 + 4) { loader(std::integral_constant<int, 2>{}, std::integral_constant<int, FOTG_RES_NH - 1>{}); return; }
  }
  if (threadIdx.x < RD) *reinterpret_cast<float2 *>(lds + DUMP + threadIdx.x * 8) = make_float2(0.f, 0.f);
  __syncthreads();

  // ---------------- writer ----------------
  // Interval J: the (du,dv) rows of chunk J - WO (the last sweep relaxed them in interval J - 1 at the latest) go from the ring
  // to the (du,dv) array the iteration writes (array it + 1), written through; their count is published WQ intervals later,
  // when "all but the newest (WQ - 1)(M + 1) stores have completed" says they are in memory (a write-through store is acknowledged
  // after a memory round trip: microseconds).  Exactly M row stores + 1 progress store per interval (rows that do not exist --
  // intervals before WO, diagonals >= S -- go to a spare row behind the arrays).  In the LAST iteration the rows become
  // flow = (wx + du, wy + dv) (refine_variational.cpp:208-221) instead: cell (diagonal d, row r) is pixel (d - r, r); the (wx,wy)
  // of a chunk are loaded one interval ahead.
  if (wv == wvW) {
    constexpr int WQ = FOTG_RES_WQ;
    int wit = 0, ws = 0;
    unsigned wslot = 0;
    int pubq[WQ];
#pragma unroll
    for (int j = 0; j < WQ; ++j) pubq[j] = 0;
    int count = 0;
    bool flushed = false, have_wq = false;
    const unsigned lo = (unsigned)(lane < RQ / 2 ? lane : 0) * 16u;
    const bool act = lane < RQ / 2;
    const unsigned rowb = (unsigned)RQ * 8u;
    unsigned woff = (unsigned)E * rowb + lo;                        // array 1, row 0
    const unsigned spare_off = (unsigned)(NIT + 1) * (unsigned)E * rowb + lo;
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)Dg, 0, ((unsigned)(NIT + 1) * (unsigned)E + 2u) * rowb, 0x00020000);
    const float2 *const wq2 = reinterpret_cast<const float2 *>(a.extra(pair, 8));
    float *const fl = g.flow + (size_t)pair * g.flow_stride;
    float2 wqn[M][2];
    auto load_wq = [&](int d0_) {                                  // (wx,wy) of the cells this lane holds of diagonals d0_ .. d0_ + M - 1
#pragma unroll
      for (int kq = 0; kq < M; ++kq)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int d = d0_ + kq, r = 2 * lane + half, i = d - r;
          const bool ok = act && d < S && r < h && i >= 0 && i < w;
          wqn[kq][half] = wq2[ok ? d * RQ + r : 0];
        }
    };
    if (FOTG_RES_DBG & 64) { for (int I = 0; I < NI; ++I) asm volatile("s_barrier" ::: "memory"); return; }
    for (int I = 0; I < NI; ++I) {
      RES_BAR_LGKM();
      const bool lastit = wit >= NIT - 1 && I >= WO;
      if (!lastit) {
#ifdef FOTG_RES_STATS
        const long long v0_ = clock64();
#endif
        if (!(FOTG_RES_DBG & 8)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((WQ - 1) * (M + 1)) : "memory");
#ifdef FOTG_RES_STATS
        st_vm += clock64() - v0_;
#endif
        __hip_atomic_store(sprog, pubq[WQ - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = WQ - 1; j > 0; --j) pubq[j] = pubq[j - 1];
        const bool on = I >= WO;
#pragma unroll
        for (int kq = 0; kq < M; ++kq) {
          const float4 v = ld_f4(DBASE + wslot + lo);
          const bool rl = on && ws + kq < S;
          if (!(FOTG_RES_DBG & 1)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rsD, rl ? woff : spare_off, 0, 16);
          if (on) { wslot += DB; if (wslot == DRING) wslot = 0; woff += rowb; }
        }
        if (on) {
          if (ws < S) count = wit * S + (ws + M < S ? ws + M : S);
          ws += M;
          if (ws == E) { ws = 0; ++wit; }
        }
        pubq[0] = count;
        continue;
      }
      if (!flushed) {                                               // the last rows of iteration NIT - 2 (nothing to flush when NIT == 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(sprog, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flushed = true;
      }
      if (wit >= NIT) continue;
      if (!have_wq) { load_wq(ws); have_wq = true; }
      float2 wqc[M][2];
#pragma unroll
      for (int kq = 0; kq < M; ++kq) { wqc[kq][0] = wqn[kq][0]; wqc[kq][1] = wqn[kq][1]; }
      load_wq(ws + M);                                              // next interval's chunk (diagonals >= S: nothing)
#pragma unroll
      for (int kq = 0; kq < M; ++kq) {
        const int d = ws + kq;
        const float4 v = ld_f4(DBASE + wslot + lo);
        wslot += DB; if (wslot == DRING) wslot = 0;
        if (d >= S || !act) continue;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int r = 2 * lane + half, i = d - r;
          const float2 x = half ? make_float2(v.z, v.w) : make_float2(v.x, v.y);
          if (r < h && i >= 0 && i < w)
            *reinterpret_cast<float2 *>(fl + 2 * (size_t)(r * w + i)) = make_float2(wqc[kq][half].x + x.x, wqc[kq][half].y + x.y);
        }
        if (a.taps) *reinterpret_cast<float4 *>(Dg + ((size_t)NIT * E + d) * rowb + lane * 16) = v;      // test taps: the final (du,dv) = array NIT
      }
      ws += M;
      if (ws == E) { ws = 0; ++wit; }
    }
    RES_STATS_OUT(0);
    return;
  }
  if (wv >= nsolver) return;                                        // spare waves leave (a barrier counts the waves that have not ended)

  // ---------------- solver wave of sweep n = wv: rows 2L and 2L+1 per lane, packed f32 (vr_sor_stream_kernel's step) ----------------
  {
    const int off = wv * DS;
    const int nl = (h + 1) >> 1;
    const float om0 = g.omega, om1 = (2 * lane + 1 < h) ? g.omega : 0.f;
    const unsigned vD = DBASE + (unsigned)lane * 16, vC = (unsigned)lane * 32;
    for (int t = 0; t < off / M; ++t) RES_BAR_PLAIN();
    if (FOTG_RES_DBG & 16) {
      for (int t = 0; t < NIT * (E / M); ++t) asm volatile("s_barrier" ::: "memory");
    } else
    if (lane < nl) {
      unsigned d0 = 0, d1 = DB, d2 = 2 * DB, c1o = CSLOT;
      float4 ow = ld_f4(d0 + vD);                                 // own values of rows 2L, 2L+1 (diagonal s)
      float4 nx = ld_f4(d1 + vD);                                 // diagonal s+1: right of both rows, bottom of row 2L
      float2 nb = ld_f2(d1 + vD + 16);                            // diagonal s+1, row 2L+2: bottom of row 2L+1
      float4 ca0 = ld_f4(vC), cb0 = ld_f4(vC + 16), ca1 = ld_f4(vC + CB), cb1 = ld_f4(vC + CB + 16);
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
      // Every step of every round is the same code: the surplus diagonals S .. E-1 of a round are zero rows in both rings (zero
      // system cell, zero (du,dv)): the update leaves them zero, and the carried left value / left weight are zero again when the
      // next round starts at diagonal 0.
      auto step2 = [&](int u) {
        const v2f top0 = {dpp_wave_shr1(p1.x), dpp_wave_shr1(p1.y)};
        if (u % M == 0) RES_BAR_LGKM();
        const float4 nnx = ld_f4(d2 + vD);                        // diagonal s+2
        const float2 nnb = ld_f2(d2 + vD + 16);
        const float4 na0 = ld_f4(c1o + vC), nb0 = ld_f4(c1o + vC + 16), na1 = ld_f4(c1o + vC + CB), nb1 = ld_f4(c1o + vC + CB + 16);
        const v2f own0 = {ow.x, ow.y}, own1 = {ow.z, ow.w}, r0 = {nx.x, nx.y}, r1 = {nx.z, nx.w}, bt1 = {nb.x, nb.y};
        const v2f q0 = relax(own0, ca0, ca1, hl0, p0, top0, r0, r1, om0);
        const v2f q1 = relax(own1, cb0, cb1, hl1, p1, p0, r1, bt1, om1);      // its top (s-1, 2L) is this lane's previous row-0 result
        *reinterpret_cast<float4 *>(lds + d0 + vD) = make_float4(q0.x, q0.y, q1.x, q1.y);
        hl0 = ca1.y; hl1 = cb1.y;
        p0 = q0; p1 = q1; ow = nx; nx = nnx; nb = nnb; ca0 = na0; cb0 = nb0; ca1 = na1; cb1 = nb1;
        d0 = d1; d1 = d2; d2 += DB; if (d2 == DRING) d2 = 0;
        c1o += CSLOT; if (c1o == CRING) c1o = 0;
      };
      for (int it = 0; it < NIT; ++it) {
        int t0 = 0;
        for (; t0 + U <= E; t0 += U) {
#pragma unroll
          for (int u = 0; u < U; ++u) step2(u);
        }
        for (; t0 < E; t0 += UT) {
#pragma unroll
          for (int u = 0; u < UT; ++u) step2(u);
        }
      }
    }                                                             // (the wave executes the loop's barriers once, whatever its exec mask)
    for (int t = off / M; t < omax / M + 1; ++t) RES_BAR_PLAIN();
    RES_STATS_OUT(0);
  }
}

}  // namespace fotg
