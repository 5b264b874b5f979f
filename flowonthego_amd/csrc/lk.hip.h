// lk.hip.h -- patch grid: template extraction + Hessian, initialisation from the coarser flow and the
// whole inverse-compositional Lucas-Kanade loop, ONE ROW OF 16 LANES PER PATCH (four patches per wave64).
//
// Reference semantics (kroeger/):
//   PatGridClass::InitializeGrid -> PatClass::InitializePatch   patchgrid.cpp:98-116, patch.cpp:57-88,287-332
//   PatGridClass::InitializeFromCoarserOF                        patchgrid.cpp:195-211
//   PatGridClass::Optimize -> PatClass::OptimizeIter            patchgrid.cpp:134-141, patch.cpp:120-212
//   OptimizeComputeErrImg / getPatchStaticBil / Loss (L2)       patch.cpp:264-284, 335-402, 223-236
//
// Mapping: a DPP row (16 lanes) owns one patch.  Pixel q of the PSxPS patch lives in lane q%16 of the row, slot q/16
// (4x4: one pixel per lane, 8x8: 4, 12x12: 9, 16x16: 16 -- every patch size fills its lanes exactly).  Template, both
// gradients and the residual stay in VGPRs for the whole loop.  Everything that is a per-patch SCALAR in the reference -- the
// 2x2 Cholesky solve, the position update, the outlier / border tests, the termination tests: most of an iteration's
// instructions, four IEEE divisions are ~45 -- is carried redundantly by the 16 lanes of the row, so one instruction
// stream serves four patches and the pixel work reads its patch's bilinear weights from its own lane: nothing is
// broadcast, nothing is packed or read back.  The reductions of an iteration (query mean; two projections and the L1
// residual) are per-lane sums over the slots followed by four row_ror DPP adds, in the order of the oracle's dis_sum();
// they leave the sum in every lane of the row, i.e. already where the scalar code wants it.  A patch that has converged
// is masked out (its row's lanes skip the evaluation), the loop ends when the wave's four patches are done.  The part of
// I1 a patch can reach -- it may move at most ps/2 from its start before it is reset (patch.cpp:199) -- is staged once
// into a row-private LDS window of (2ps+4)^2 pixels, so the loop reads no global memory.
#pragma once
#include "common.h"
#include "fdiv_hoist.h"

namespace fotg {

struct LkArgs {
  const float *I0, *I0x, *I0y, *I1;   // padded level images, `img_stride` floats between pairs
  long img_stride;
  const float *flow_prev;             // (h/2 x w/2 x 2) per pair or nullptr
  long flow_prev_stride;
  float *p_iter;                      // [n][nop][2]
  float *pweight;                     // [n][nop][nv]
  float *tmpl, *tdx, *tdy;            // optional taps [n][nop][nv] (nullptr in production)
  float *hes;                         // optional [n][nop][3]
  int *cnt;                           // optional [n][nop]
  float *trace;                       // optional [nop][(max_iter+1)][4], pair 0 only
  LevelGeom g;
  int camlr;                          // depth mode: 0 displacement <= 0 (forward grid), 1 displacement >= 0 (oflow.cpp:153,157)
  int max_iter, min_iter, patnorm, costfct;
  int nwg;                            // > 0: XCD-banded placement of the workgroups (xcd_banded_x, common.h): launches whose pair count is not a multiple of 8
  int shw_test;                       // test tap of the shared-window kernels (FOTG_LK_SHW=2 / 3 with FOTG_TEST_TAPS=1): 1 = rows 1, 3 read global memory, 2 = all rows
  float dp_thresh_sq, dr_thresh, res_thresh, outlier, outlier_sq, huber_bsq, huber_2bsq;
};

// DEPTH: stereo depth mode (SELECTMODE 2): the parameter is ONE horizontal displacement per patch -- scalar Hessian
// sum(Tx^2) (patch.cpp:83-87), one projection (:181), sign clamp after the update (:188-193), pt_iter.y stays at the
// reference row (:218-220); flow_prev has one channel (patchgrid.cpp:207-208).  p_iter keeps two slots (second = 0).

// x / NV for the element counts NV = ps*ps*noc of the supported patch sizes, correctly rounded in three instructions:
// q0 = x * RN(1/NV), q = fma(fma(-q0, NV, x), RN(1/NV), q0).  tools/div_const_check.c compares this with x / NV for EVERY
// finite float x and each of 16, 48, 64, 192, 144, 432, 256, 768 (equal for |x| >= 2^-100 -- below that the remainder can
// underflow, and the IEEE division is used).
template <int NV>
__device__ __forceinline__ float div_nv(float x)
{
  constexpr float b = (float)NV, r = 1.0f / b;
  const float q0 = x * r;
  float q = __builtin_fmaf(__builtin_fmaf(-q0, b, x), r, q0);
  const bool small = !(__builtin_fabsf(x) >= 0x1p-100f);
  if (__builtin_amdgcn_ballot_w64(small) != 0) q = small ? x / b : q;      // (wave-uniform branch, taken for all-zero patches)
  return q;
}

// L2: the cost function is the L2 one of every operating point (costfct 0), known at compile time -- the per-element switch
// of the general kernel and the code of the two other cost functions are gone from the loop.
//
// SHW: ONE LDS area for the wave's four windows.  The four patches of a wave are consecutive ids = vertical neighbours `steps`
// pixels apart (patchgrid.cpp:57-66), so their reachable windows overlap by 3/4 and more; private windows cost 4 (2ps+4)^2 floats
// per wave (12.5 KB at ps 12: three waves per SIMD), the union of four neighbours (2ps+4) x (2ps+4 + 3 steps).  The windows are
// packed greedily, in row order, into at most two groups of overlapping windows (a wave whose ids wrap from the bottom of one
// grid column to the top of the next, or that straddles a motion boundary, has two) inside SWW x RB pixels.  A window that
// does not fit (starts further apart than the slack in x, or a third group) is not staged: that patch reads its taps from
// global memory with the same clamping -- same values, only slower, and rare.
// waves per SIMD the register allocation of the shared-window kernels must leave room for (what their LDS area allows)
constexpr int lk_min_waves(int nv, bool shw, int lpp)
{
  return lpp == 8 ? (nv <= 64 ? 4 : nv <= 144 ? 3 : 1) : !shw ? 1 : nv == 48 ? 5 : nv <= 64 ? 6 : nv <= 144 ? 5 : nv <= 192 ? 3 : 1;
}

// the tree of dis_sum() below the 16 partials, inside a group of LPP lanes; the result in every lane of the group.
// LPP = 16: xor 8, 4, 2, 1 as row rotations (see common.h).  LPP = 8 (the lane already holds partial i + partial i + 8): the
// true xor-4 partner through two bank-masked row shifts, xor 2 and xor 1 as quad permutations.
template <int LPP>
__device__ __forceinline__ float group_allsum(float v)
{
  if constexpr (LPP == 16) return row_allsum(v);
  else {
#define FOTG_DPPM(old_, x, ctrl, bank) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old_), __builtin_bit_cast(int, x), ctrl, 0xF, bank, false))
    const float lo = FOTG_DPPM(0.f, v, 0x104, 0x5);      // row_shl:4 -> lanes 0-3, 8-11 read lane + 4
    const float pr = FOTG_DPPM(lo, v, 0x114, 0xA);       // row_shr:4 -> lanes 4-7, 12-15 read lane - 4
    v = v + pr;
    v = v + FOTG_DPPM(0.f, v, 0x4E, 0xF);                // quad_perm [2,3,0,1]: xor 2
    v = v + FOTG_DPPM(0.f, v, 0xB1, 0xF);                // quad_perm [1,0,3,2]: xor 1
#undef FOTG_DPPM
    return v;
  }
}

// LPP: lanes per patch.  16 (default): four patches per wave.  8: eight patches per wave, every lane carries the pixels of two
// of the 16 virtual lanes of dis_sum() (j and j + 8: their partials are summed in the lane = the tree's first level), so the
// per-patch scalar code -- half of an iteration's instructions -- serves twice as many patches; needs the shared LDS area and
// ~1.6x the registers: for launches with enough waves to stay throughput-bound at three waves per SIMD.
template <int PS, int NOC, bool DEPTH = false, bool L2 = false, bool SHW = false, int LPP = 16>
__global__ __launch_bounds__(64, lk_min_waves(PS * PS * NOC, SHW, LPP)) void lk_kernel(LkArgs a)
{
  static_assert(LPP == 16 || (LPP == 8 && SHW), "eight lanes per patch only with the shared LDS area");
  constexpr int PPW = 64 / LPP;                      // patches per wave
  constexpr int NVL = 16 / LPP;                      // virtual lanes (of the 16 of dis_sum()) per lane
  constexpr int NPIX = PS * PS;
  constexpr int NSL = NPIX / 16;                     // pixels per virtual lane
  constexpr int NE1 = NSL * NOC;                     // elements per virtual lane
  constexpr int NE = NE1 * NVL;                      // elements per lane: [virtual lane][slot][channel]
  constexpr int NV = NPIX * NOC;
  constexpr int PAD = PS;
  constexpr int WIN = 2 * PS + 4;                    // window edge, see the column bound below
  constexpr int SWW = WIN + 4;                       // SHW: width of the shared area (slack for starts that differ in x)
  constexpr int RB = 2 * WIN + 8 + (PPW - 4) * (PS / 2 + 2);   // SHW: rows of the shared area (two groups + their vertical spread)
  constexpr int WS = SHW ? SWW : WIN;                // row stride of a patch's window, pixels
  static_assert(NPIX % 16 == 0, "a patch fills the 16 lanes of its row");
  __shared__ float win_all[SHW ? SWW * RB * NOC : PPW * WIN * WIN * NOC];
  const int lane = threadIdx.x & 63, row = lane / LPP, j = lane % LPP;
  WgId wg = xcd_local_wg();                          // all patches of a pair on the XCD of its refinement workgroup
  // a launch for 1..7 pairs (a single 4K pair): consecutive waves hold neighbouring patches, whose templates and windows overlap (a pixel is
  // in (ps / steps)^2 patches); dealt round robin, every XCD's L2 fetched the whole level for itself (4K level 2: 110 MB per launch for 45
  // MB of inputs + outputs).  Every XCD gets a contiguous run of patch columns instead.
  if (a.nwg > 0) { wg.x = xcd_banded_x(a.nwg); if (wg.x < 0) return; }
  const int ipw = wg.x * PPW;                        // first patch of this wave
  const int pair = wg.y;
  const int tw = a.g.tw;
  const float *I0 = a.I0 + (size_t)pair * a.img_stride;
  const float *I0x = a.I0x + (size_t)pair * a.img_stride;
  const float *I0y = a.I0y + (size_t)pair * a.img_stride;
  const float *I1 = a.I1 + (size_t)pair * a.img_stride;
  const bool VALID = ipw + row < a.g.nop;            // surplus rows of the last wave shadow the last patch
  const int IP = VALID ? ipw + row : a.g.nop - 1;

  // per-lane sums over the slots (element order = the oracle's) and the row tree
  // (with two virtual lanes per lane: one partial each, in element order, then their sum = the tree's xor-8 level)
  auto lane_sum = [&](const float *v) {
    float acc[NVL];
#pragma unroll
    for (int vl = 0; vl < NVL; ++vl) {
      acc[vl] = v[vl * NE1];
#pragma unroll
      for (int k = 1; k < NE1; ++k) acc[vl] = acc[vl] + v[vl * NE1 + k];
    }
    return NVL == 2 ? acc[0] + acc[NVL - 1] : acc[0];
  };
  auto lane_sum_abs = [&](const float *v) {
    float acc[NVL];
#pragma unroll
    for (int vl = 0; vl < NVL; ++vl) {
      acc[vl] = fabsf(v[vl * NE1]);
#pragma unroll
      for (int k = 1; k < NE1; ++k) acc[vl] = acc[vl] + fabsf(v[vl * NE1 + k]);
    }
    return NVL == 2 ? acc[0] + acc[NVL - 1] : acc[0];
  };
  auto lane_dot = [&](const float *x, const float *y) {
    float acc[NVL];
#pragma unroll
    for (int vl = 0; vl < NVL; ++vl) {
      acc[vl] = x[vl * NE1] * y[vl * NE1];
#pragma unroll
      for (int k = 1; k < NE1; ++k) acc[vl] = acc[vl] + x[vl * NE1 + k] * y[vl * NE1 + k];
    }
    return NVL == 2 ? acc[0] + acc[NVL - 1] : acc[0];
  };

  // ---- template + gradients at round(pt_ref)+pad (patch.cpp:287-332), Hessian sums (:74-77), starting flow
  //      (patchgrid.cpp:195-211), I1 window -> LDS
  // patch id -> reference position (patchgrid.cpp:57-66: i = x*noph + y)
  const int gx = IP / a.g.noph, gy = IP % a.g.noph;
  const float RX = (float)(gx * a.g.steps + a.g.offw), RY = (float)(gy * a.g.steps + a.g.offh);
  float T[NE], Tx[NE], Ty[NE], r[NE];                 // (the patch weights written at the end are |r|)
  // window offset of the lane's pixels relative to the patch centre: pixel q = 16 s + j sits 16 PER / PS rows below pixel
  // q - 16 PER in the same column (PER = slots after which 16 s is a multiple of the patch width), so PER per-lane offsets and a
  // compile-time row term (an immediate of the LDS read) cover all slots
  constexpr int PER = PS == 12 ? 3 : 1, PROWS = 16 * PER / PS;
  static_assert((16 * PER) % PS == 0 && NSL % PER == 0, "slot period");
  int pu[NVL * PER];
  {
    const int px = (int)RX + PAD, py = (int)RY + PAD;              // pt_ref is integer valued: round() is exact
#pragma unroll
    for (int vl = 0; vl < NVL; ++vl)
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int q = s * 16 + j + vl * LPP;
      const int offy = q / PS - PS / 2, offx = q % PS - PS / 2;
      if (s < PER) pu[vl * PER + s] = ((offy - 1) * WS + offx - 1) * NOC;      // the upper left tap: the lowest address of the four
      const size_t idx = ((size_t)(px + offx) + (size_t)(py + offy) * tw) * NOC;
#pragma unroll
      for (int c = 0; c < NOC; ++c) {
        T[vl * NE1 + s * NOC + c] = I0[idx + c];
        Tx[vl * NE1 + s * NOC + c] = I0x[idx + c];
        Ty[vl * NE1 + s * NOC + c] = I0y[idx + c];
        r[vl * NE1 + s * NOC + c] = 0.f;
      }
    }
  }
  float PIN0 = 0.f, PIN1 = 0.f;
  if (a.flow_prev) {
    int fx = (int)floorf(RX / 2), fy = (int)floorf(RY / 2);
    // oracle definition D5: a level of odd size has patches whose half coordinate is one past the coarser array (only
    // reachable with `initflow`); the reference reads out of bounds there, here the index is clamped
    fx = fx > a.g.w / 2 - 1 ? a.g.w / 2 - 1 : fx;
    fy = fy > a.g.h / 2 - 1 ? a.g.h / 2 - 1 : fy;
    const float *fp = a.flow_prev + (size_t)pair * a.flow_prev_stride + (DEPTH ? 1 : 2) * (size_t)(fy * (a.g.w / 2) + fx);
    PIN0 = fp[0] * 2;
    if constexpr (!DEPTH) PIN1 = fp[1] * 2;
  }
  // OptimizeStart (patch.cpp:120-156)
  float P0 = PIN0, P1 = PIN1, PTX = RX + P0, PTY = RY + P1;
  const float STX = PTX, STY = PTY;
  const bool START_OK = VALID && !(PTX < a.g.lb || PTY < a.g.lb || PTX > a.g.ubw || PTY > a.g.ubh);
  // Stage the reachable window of I1 in LDS.  Every evaluated position is within ps/2 of the start in x and y, so the
  // bilinear taps span columns floor(stx)-ps-1 .. floor(stx)+ps+2 (padded coordinates: + PAD); rows likewise.
  const int WX0 = (int)floorf(STX) + PAD - PS - 1, WY0 = (int)floorf(STY) + PAD - PS - 1;
  const float *win = win_all + row * (WIN * WIN * NOC);
  bool USEG = false;                                     // SHW: this row's window is not staged, its taps come from global memory
  if constexpr (SHW) {
    // wave-uniform plan (all scalar): greedy packing of the starting rows' windows, in row order, into <= 2 groups A, B
    unsigned need;
    {
      const unsigned long long b = __builtin_amdgcn_ballot_w64(START_OK);
      need = 0;
#pragma unroll
      for (int r4 = 0; r4 < PPW; ++r4) need |= (unsigned)((b >> (LPP * r4)) & 1) << r4;
      if (a.shw_test == 1) need &= 0x55u;                // test tap: the odd rows take the global-memory path
      if (a.shw_test == 2) need = 0;                     // test tap: every row takes it
    }
    int xa0 = 0, xa1 = 0, ya0 = 0, ya1 = 0, xb0 = 0, xb1 = 0, yb0 = 0, yb1 = 0, ng = 0;
    unsigned staged = 0, grp1 = 0;                       // bit r: row r is staged / sits in group B
#pragma unroll
    for (int r4 = 0; r4 < PPW; ++r4) {
      const int wxr = __builtin_amdgcn_readlane(WX0, LPP * r4), wyr = __builtin_amdgcn_readlane(WY0, LPP * r4);
      if (!((need >> r4) & 1)) continue;
      const bool inB = ng == 2;
      const int cx0 = inB ? xb0 : xa0, cx1 = inB ? xb1 : xa1, cy0 = inB ? yb0 : ya0, cy1 = inB ? yb1 : ya1;
      const int nx0 = min(cx0, wxr), nx1 = max(cx1, wxr + WIN), ny0 = min(cy0, wyr), ny1 = max(cy1, wyr + WIN);
      const int ha = ya1 - ya0;
      const bool fit = ng > 0 && nx1 - nx0 <= SWW && (inB ? ha : 0) + (ny1 - ny0) <= RB;
      const bool fresh = !fit && (ng == 0 || (ng == 1 && ha + WIN <= RB));
      if (fit) {
        if (inB) { xb0 = nx0; xb1 = nx1; yb0 = ny0; yb1 = ny1; } else { xa0 = nx0; xa1 = nx1; ya0 = ny0; ya1 = ny1; }
      } else if (fresh) {
        if (ng == 0) { xa0 = wxr; xa1 = wxr + WIN; ya0 = wyr; ya1 = wyr + WIN; } else { xb0 = wxr; xb1 = wxr + WIN; yb0 = wyr; yb1 = wyr + WIN; }
        ++ng;
      }
      if (fit || fresh) { staged |= 1u << r4; if (ng == 2) grp1 |= 1u << r4; }
    }
    const int base1 = ya1 - ya0;                         // first LDS row of group B
    int woff = 0;
#pragma unroll
    for (int r4 = 0; r4 < PPW; ++r4) {
      const int wxr = __builtin_amdgcn_readlane(WX0, LPP * r4), wyr = __builtin_amdgcn_readlane(WY0, LPP * r4);
      const bool b = (grp1 >> r4) & 1;
      const int o = (((b ? base1 : 0) + wyr - (b ? yb0 : ya0)) * SWW + (wxr - (b ? xb0 : xa0))) * NOC;
      woff = row == r4 ? o : woff;
    }
    win = win_all + woff;
    USEG = START_OK && !((staged >> row) & 1);
    for (int g = 0; g < ng; ++g) {
      const int fy0 = g ? yb0 : ya0, fx0 = g ? xb0 : xa0;
      const int cells = (g ? yb1 - yb0 : base1) * SWW, lbase = (g ? base1 : 0) * SWW;
      for (int t = lane; t < cells; t += 64) {
        const int sy = t / SWW, sx = t - sy * SWW;
        const size_t src = ((size_t)clampi(fy0 + sy, a.g.th) * tw + clampi(fx0 + sx, tw)) * NOC;
#pragma unroll
        for (int c = 0; c < NOC; ++c) win_all[(lbase + t) * NOC + c] = I1[src + c];
      }
    }
  } else if (START_OK) {
    float *const wdst = win_all + row * (WIN * WIN * NOC);
    for (int t = j; t < WIN * WIN; t += LPP) {
      const int wy = t / WIN, wx = t - wy * WIN;
      const size_t src = ((size_t)clampi(WY0 + wy, a.g.th) * tw + clampi(WX0 + wx, tw)) * NOC;
#pragma unroll
      for (int c = 0; c < NOC; ++c) wdst[t * NOC + c] = I1[src + c];
    }
  }
  // the window is filled by the lanes of the row and read by all of them: keep the compiler (which reasons per thread) from
  // moving reads above the fill; the hardware executes a wave's LDS accesses in order
  asm volatile("" ::: "memory");

  // template mean (patch.cpp:330-331) and Hessian sums (:74-77, depth :84)
  if (a.patnorm > 0) {
    const float m = div_nv<NV>(group_allsum<LPP>(lane_sum(T)));
#pragma unroll
    for (int e = 0; e < NE; ++e) T[e] -= m;
  }
  float H00 = group_allsum<LPP>(lane_dot(Tx, Tx));
  const float H01 = DEPTH ? 0.f : group_allsum<LPP>(lane_dot(Tx, Ty));
  float H11 = DEPTH ? 0.f : group_allsum<LPP>(lane_dot(Ty, Ty));
  if constexpr (DEPTH) {
    if (H00 == 0.f) H00 = (float)((double)H00 + 1e-10);  // :85-86
  } else if (H00 * H11 - H01 * H01 == 0.f) {             // :78-82  (float += 1e-10 in double, like the reference)
    H00 = (float)((double)H00 + 1e-10);
    H11 = (float)((double)H11 + 1e-10);
  }
  if (a.hes && VALID && j == 0) {                        // (test tap; written here so that the three sums do not live through the loop)
    float *hp = a.hes + ((size_t)pair * a.g.nop + IP) * 3;
    hp[0] = H00; hp[1] = H01; hp[2] = H11;
  }
  // Cholesky factor of the (constant) Hessian, hoisted out of the loop: same values every iteration
  const float L00 = sqrtf(H00);
  const float L10 = DEPTH ? 0.f : H01 / L00;
  const float L11 = DEPTH ? 1.f : sqrtf(H11 - L10 * L10);
  // The solve divides by L00 and L11 four times per iteration: the reciprocal refinement of those divisions is hoisted
  // (fdiv_hoist.h); a quotient outside the guarded range sends the whole wave through the compiler's divisions instead.
  const InvDiv iL00 = make_invdiv(L00), iL11 = make_invdiv(L11);
  const bool DEN_OK = iL00.ok && iL11.ok;
  bool CONV = !START_OK;                                 // :135-141; pweight stays 0 (oracle definition D2)
  int CNT = 0;
  float DP0 = 0.f, DP1 = 0.f, DPN_INIT = 1e-10f, MARES = 1e5f, MARES_OLD = 1e20f;
  const int trow = (a.max_iter + 1) * 4;
  float *trace = (a.trace && pair == 0) ? a.trace : nullptr;
  if (trace && VALID && j < 4)
    for (int t = 0; t <= a.max_iter; ++t) trace[(size_t)IP * trow + t * 4 + j] = 0.f;

  // ---- OptimizeComputeErrImg (:264-284) for the rows that are still running: bilinear query patch (:335-402), mean,
  //      residual, and the three sums the patch needs next -- the two projections on the steepest-descent images
  //      (:178-179, used by the NEXT update) and the L1 residual (:278)
  // RC: the residual does not live across the loop; it is evaluated once more at the final position for the patch weights
  // (the same position gives the same bits).  One evaluation in max_iter + 1 more, nine registers and their copies at the loop
  // head less: for the 128-iteration operating points' 12 x 12 patches.
  constexpr bool RC = L2 && NSL * NVL >= 9 && (!SHW || LPP == 8);
  float B0 = 0.f, B1 = 0.f;
  auto residual = [&](float (&rr)[NE]) {
    const int pos2 = (int)floorf(PTX), pos3 = (int)floorf(PTY);
    const int pos0 = (int)ceilf(PTX + .00001f) + PAD - WX0, pos1 = (int)ceilf(PTY + .00001f) + PAD - WY0;   // window coordinates
    const float r0 = PTX - (float)pos2, r1 = PTY - (float)pos3;
    const float we0 = r0 * r1, we1 = (1 - r0) * r1, we2 = r0 * (1 - r1), we3 = (1 - r0) * (1 - r1);
    const int iab = (pos1 * WS + pos0) * NOC;
    float q[NE];
    if (SHW && USEG) {
      // this row's window is not in LDS (see SHW above): the same four taps from the level image, clamped like the staged copy
#pragma unroll 1
      for (int sv = 0; sv < NSL * NVL; ++sv) {
        const int qq = (sv % NSL) * 16 + j + (sv / NSL) * LPP;
        const int ax = pos0 + WX0 + qq % PS - PS / 2, ay = pos1 + WY0 + qq / PS - PS / 2;
        const int x1 = clampi(ax, tw), x0 = clampi(ax - 1, tw), y1 = clampi(ay, a.g.th), y0 = clampi(ay - 1, a.g.th);
#pragma unroll
        for (int c = 0; c < NOC; ++c) {
          const float va = I1[((size_t)y1 * tw + x1) * NOC + c], vb = I1[((size_t)y1 * tw + x0) * NOC + c];
          const float vc = I1[((size_t)y0 * tw + x1) * NOC + c], vd = I1[((size_t)y0 * tw + x0) * NOC + c];
          const float v = we0 * va + we1 * vb + we2 * vc + we3 * vd;
#pragma unroll
          for (int s2 = 0; s2 < NSL * NVL; ++s2) if (s2 == sv) q[s2 * NOC + c] = v;      // (register array: constant indices only)
        }
      }
    } else {
      // (the bases are made opaque: the compiler otherwise folds the window-centre constants into every tap's address and then
      // needs one address add per tap -- 18 at ps 12 -- instead of the 8-bit offset fields of the LDS reads)
      const float *tps[NVL * PER];
#pragma unroll
      for (int u = 0; u < NVL * PER; ++u) { int ib = iab + pu[u]; asm volatile("" : "+v"(ib)); tps[u] = win + ib; }
#pragma unroll
      for (int vl = 0; vl < NVL; ++vl)
#pragma unroll
      for (int s = 0; s < NSL; ++s) {
        // taps at non-negative compile-time offsets from one per-lane base per slot class: immediates of the LDS reads
        const float *tp = tps[vl * PER + s % PER];
        const int o = (s / PER) * (PROWS * WS * NOC);
#pragma unroll
        for (int c = 0; c < NOC; ++c) {
          const float vd = tp[o + c], vc = tp[o + NOC + c], vb = tp[o + WS * NOC + c], va = tp[o + WS * NOC + NOC + c];
          q[vl * NE1 + s * NOC + c] = we0 * va + we1 * vb + we2 * vc + we3 * vd;
        }
      }
    }
    if (a.patnorm > 0) {
      const float m = div_nv<NV>(group_allsum<LPP>(lane_sum(q)));
#pragma unroll
      for (int e = 0; e < NE; ++e) q[e] -= m;
    }
    // (the cost-function switch stays inside the element loop: with the switch hoisted around three loops hipcc 7.2 produced
    // different bits for costfct 1 -- test_patch_cost_functions -- and the two scalar branches per element cost ~2%)
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      float d = q[e] - T[e];                               // :230-236 L2: the difference image itself
      if constexpr (L2) { rr[e] = d; continue; }
      if (a.costfct == 1) d = copysignf(sqrtf(fabsf(d)), d);                                      // :238-246 L1
      else if (a.costfct == 2) d = copysignf(sqrtf((sqrtf(1.0f + (d * d) / a.huber_bsq) - 1.0f) * a.huber_2bsq), d);   // :247-261
      rr[e] = d;
    }
  };
  auto eval = [&]() {
    float rl[NE];
    float (&rr)[NE] = RC ? rl : r;
    residual(rr);
    B0 = group_allsum<LPP>(lane_dot(Tx, rr));
    if constexpr (!DEPTH) B1 = group_allsum<LPP>(lane_dot(Ty, rr));
    const float dpn = DP0 * DP0 + DP1 * DP1;             // :272
    if (CNT == 1) DPN_INIT = dpn;
    MARES_OLD = MARES;
    MARES = div_nv<NV>(group_allsum<LPP>(lane_sum_abs(rr)));    // :278
    // :279-282 (the two rate tests only matter once cnt >= min_iter)
    bool go = (CNT < a.max_iter) & (MARES > a.res_thresh);
    if (go && CNT >= a.min_iter) go = (dpn / DPN_INIT >= a.dp_thresh_sq) & (MARES / MARES_OLD <= a.dr_thresh);
    if (!go) CONV = true;
    if (trace && j == 0 && CNT <= a.max_iter) {
      float *tr = trace + (size_t)IP * trow + CNT * 4;
      tr[0] = P0; tr[1] = P1; tr[2] = MARES; tr[3] = (float)CNT;
    }
  };

  if (!CONV) eval();                                     // OptimizeStart's first error image (:154)
  while (__builtin_amdgcn_ballot_w64(!CONV) != 0) {
    if (!CONV) {                                         // rows whose patch runs this iteration (all 16 lanes of a row agree)
      // 2x2 LLT solve (:184); depth mode: the 1x1 system, L = sqrt(H)
      float y0 = fdiv_fast(B0, iL00);
      float x0, x1;
      bool QOK;
      if constexpr (DEPTH) { x0 = fdiv_fast(y0, iL00); x1 = 0.f; QOK = DEN_OK & fdiv_in_range(y0) & fdiv_in_range(x0); }
      else {
        const float y1 = fdiv_fast(B1 - L10 * y0, iL11);
        x1 = fdiv_fast(y1, iL11);
        x0 = fdiv_fast(y0 - L10 * x1, iL00);
        const float hi = fmaxf(fmaxf(fabsf(y0), fabsf(y1)), fmaxf(fabsf(x1), fabsf(x0)));
        const float lo = fminf(fminf(fabsf(y0), fabsf(y1)), fminf(fabsf(x1), fabsf(x0)));
        QOK = DEN_OK & (lo >= 0x1p-40f) & (hi <= 0x1p40f);       // (a NaN escapes fmaxf / fminf but ends in x0, which is tested below)
      }
      if (__builtin_amdgcn_ballot_w64(!QOK) != 0) {              // zero / tiny / huge / non-finite operands: the IEEE divisions (wave-uniform)
        y0 = B0 / L00;
        if constexpr (DEPTH) { x0 = y0 / L00; x1 = 0.f; }
        else {
          const float y1 = (B1 - L10 * y0) / L11;
          x1 = y1 / L11;
          x0 = (y0 - L10 * x1) / L00;
        }
      }
      float nP0 = P0 - x0, nP1 = P1 - x1;                // :186
      if constexpr (DEPTH) {                             // :188-193 std::min / std::max with 0
        if (a.camlr == 0) nP0 = (0.0f < nP0) ? 0.0f : nP0;
        else nP0 = (nP0 < 0.0f) ? 0.0f : nP0;
      }
      float nPTX = RX + nP0, nPTY = RY + nP1;
      const float ddx = STX - nPTX, ddy = STY - nPTY;
      const bool bad = !(isfinite(x0) && isfinite(x1));  // oracle definition D3
      // :199-208; norm > outlier as a test on the squared norm: a.outlier_sq is the largest float whose correctly rounded
      // square root is <= outlier (found on the host), and sqrtf is monotonic
      const bool reset = bad || ddx * ddx + ddy * ddy > a.outlier_sq ||
                         nPTX < a.g.lb || nPTY < a.g.lb || nPTX > a.g.ubw || nPTY > a.g.ubh;
      if (reset) { nP0 = PIN0; nP1 = PIN1; nPTX = RX + PIN0; nPTY = RY + PIN1; }
      CNT++;
      DP0 = (reset && bad) ? 0.f : x0; DP1 = (reset && bad) ? 0.f : x1;
      P0 = nP0; P1 = nP1; PTX = nPTX; PTY = nPTY;
      if (reset) CONV = true;
      eval();
    }
  }

  // ---- results ----
  if constexpr (RC) { if (START_OK) residual(r); }
  if (!VALID) return;
  size_t pb = (size_t)pair * a.g.nop + IP;
  int jj = j;
  asm volatile("" : "+v"(pb), "+v"(jj));                 // (keeps the store addresses / element indices from being formed before the loop and held across it)
  if (jj == 0) {
    a.p_iter[pb * 2] = P0;
    a.p_iter[pb * 2 + 1] = P1;
    if (a.cnt) a.cnt[pb] = CNT;
  }
#pragma unroll
  for (int vl = 0; vl < NVL; ++vl)
#pragma unroll
  for (int s = 0; s < NSL; ++s) {
    const size_t e = pb * NV + (size_t)(s * 16 + jj + vl * LPP) * NOC;
    const int k = vl * NE1 + s * NOC;
#pragma unroll
    for (int c = 0; c < NOC; ++c) {
      a.pweight[e + c] = fabsf(r[k + c]);
      if (a.tmpl) { a.tmpl[e + c] = T[k + c]; a.tdx[e + c] = Tx[k + c]; a.tdy[e + c] = Ty[k + c]; }
    }
  }
}

}  // namespace fotg
