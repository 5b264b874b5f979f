// lk.hip.h -- patch grid: template extraction + Hessian, initialisation from the coarser flow and the
// whole inverse-compositional Lucas-Kanade loop, ONE WAVE64 PER PATCH.
//
// Reference semantics (kroeger/):
//   PatGridClass::InitializeGrid -> PatClass::InitializePatch   patchgrid.cpp:98-116, patch.cpp:57-88,287-332
//   PatGridClass::InitializeFromCoarserOF                        patchgrid.cpp:195-211
//   PatGridClass::Optimize -> PatClass::OptimizeIter            patchgrid.cpp:134-141, patch.cpp:120-212
//   OptimizeComputeErrImg / getPatchStaticBil / Loss (L2)       patch.cpp:264-284, 335-402, 223-236
//
// Mapping: pixel q of the PSxPS patch lives in lane q%64, slot q/64 (8x8: exactly one pixel per lane;
// 12x12: 3 slots, the last one 16 lanes wide).  Template, both gradients and the residual stay in
// VGPRs for the whole loop; the three reductions per iteration (two projections, query mean, L1
// residual) are wave butterflies (wave_sum) whose order is the oracle's dis_sum().  All lanes carry
// the same scalar state, so the 2x2 Cholesky solve and the termination tests are computed redundantly
// and the loop branch is wave-uniform.  The part of I1 the patch can reach -- it may move at most ps/2 from its start
// before it is reset (patch.cpp:199) -- is staged once into a wave-private LDS window of (2ps+4)^2 pixels, so the
// loop reads no global memory.
#pragma once
#include "common.h"

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
  int ps_unused;
  int max_iter, min_iter, patnorm;
  float dp_thresh_sq, dr_thresh, res_thresh, outlier;
};

template <int PS, int NOC>
__global__ __launch_bounds__(256) void lk_kernel(LkArgs a)
{
  constexpr int NPIX = PS * PS;
  constexpr int NSLOT = (NPIX + 63) / 64;
  constexpr int NV = NPIX * NOC;
  constexpr int PAD = PS;
  constexpr int WIN = 2 * PS + 4;                    // window edge, see the column bound below
  __shared__ float win_all[4][WIN * WIN * NOC];
  float *win = win_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int ip = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ip >= a.g.nop) return;                         // wave-uniform
  const int pair = blockIdx.y;
  const int tw = a.g.tw;
  const float *I0 = a.I0 + (size_t)pair * a.img_stride;
  const float *I0x = a.I0x + (size_t)pair * a.img_stride;
  const float *I0y = a.I0y + (size_t)pair * a.img_stride;
  const float *I1 = a.I1 + (size_t)pair * a.img_stride;

  // patch id -> reference position (patchgrid.cpp:57-66: i = x*noph + y)
  const int gx = ip / a.g.noph, gy = ip % a.g.noph;
  const float rx = (float)(gx * a.g.steps + a.g.offw), ry = (float)(gy * a.g.steps + a.g.offh);

  // per-lane pixel offsets inside the patch
  int offx[NSLOT], offy[NSLOT];
  bool have[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int q = s * 64 + lane;
    have[s] = q < NPIX;
    offy[s] = q / PS - PS / 2;
    offx[s] = q % PS - PS / 2;
  }

  // ---- template + gradients at round(pt_ref)+pad (patch.cpp:287-332) ----
  float T[NSLOT * NOC], Tx[NSLOT * NOC], Ty[NSLOT * NOC];
  {
    const int px = (int)rx + PAD, py = (int)ry + PAD;    // pt_ref is integer valued: round() is exact
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const size_t idx = ((size_t)(px + offx[s]) + (size_t)(py + offy[s]) * tw) * NOC;
#pragma unroll
      for (int c = 0; c < NOC; ++c) {
        T[s * NOC + c] = have[s] ? I0[idx + c] : 0.f;
        Tx[s * NOC + c] = have[s] ? I0x[idx + c] : 0.f;
        Ty[s * NOC + c] = have[s] ? I0y[idx + c] : 0.f;
      }
    }
  }
  auto lane_sum = [&](const float *v) {
    float acc = v[0];
#pragma unroll
    for (int k = 1; k < NSLOT * NOC; ++k)
      if (have[k / NOC]) acc = acc + v[k];
    return acc;
  };
  auto lane_dot = [&](const float *x, const float *y) {
    float acc = x[0] * y[0];
#pragma unroll
    for (int k = 1; k < NSLOT * NOC; ++k)
      if (have[k / NOC]) acc = acc + x[k] * y[k];
    return acc;
  };
  if (a.patnorm > 0) {                                   // patch.cpp:330-331
    const float m = wave_sum(lane_sum(T)) / (float)NV;
#pragma unroll
    for (int k = 0; k < NSLOT * NOC; ++k) T[k] -= m;
  }
  float h00 = wave_sum(lane_dot(Tx, Tx));                // patch.cpp:74-77
  float h01 = wave_sum(lane_dot(Tx, Ty));
  float h11 = wave_sum(lane_dot(Ty, Ty));
  if (h00 * h11 - h01 * h01 == 0.f) {                    // :78-82  (float += 1e-10 in double, like the reference)
    h00 = (float)((double)h00 + 1e-10);
    h11 = (float)((double)h11 + 1e-10);
  }

  // Cholesky factor of the (constant) Hessian, hoisted out of the loop: same values every iteration
  const float l00 = sqrtf(h00);
  const float l10 = h01 / l00;
  const float l11 = sqrtf(h11 - l10 * l10);

  // ---- starting flow (patchgrid.cpp:195-211): nearest neighbour of the coarser flow, x2 ----
  float pin0 = 0.f, pin1 = 0.f;
  if (a.flow_prev) {
    const int fx = (int)floorf(rx / 2), fy = (int)floorf(ry / 2);
    const float *fp = a.flow_prev + (size_t)pair * a.flow_prev_stride + 2 * (size_t)(fy * (a.g.w / 2) + fx);
    pin0 = fp[0] * 2;
    pin1 = fp[1] * 2;
  }

  // ---- OptimizeStart (patch.cpp:120-156) ----
  float p0 = pin0, p1 = pin1;
  float ptx = rx + p0, pty = ry + p1;
  const float stx = ptx, sty = pty;
  bool conv = false;
  int cnt = 0;
  float dp0 = 0.f, dp1 = 0.f, dpn_init = 1e-10f, mares = 1e5f, mares_old = 1e20f;
  float r[NSLOT * NOC], wabs[NSLOT * NOC];
#pragma unroll
  for (int k = 0; k < NSLOT * NOC; ++k) { r[k] = 0.f; wabs[k] = 0.f; }
  float *trace = (a.trace && pair == 0) ? a.trace + (size_t)ip * (a.max_iter + 1) * 4 : nullptr;
  if (trace && lane < 4)
    for (int t = 0; t <= a.max_iter; ++t) trace[t * 4 + lane] = 0.f;

  const bool start_ok = !(ptx < a.g.lb || pty < a.g.lb || ptx > a.g.ubw || pty > a.g.ubh);
  if (!start_ok) conv = true;                            // :135-141; pweight stays 0 (oracle definition D2)

  // ---- stage the reachable window of I1 in LDS.  Every evaluated position is within ps/2 of the start in x and y,
  // so the bilinear taps span columns floor(stx)-ps-1 .. floor(stx)+ps+2 (padded coordinates: + PAD); rows likewise.
  const int wx0 = (int)floorf(stx) + PAD - PS - 1, wy0 = (int)floorf(sty) + PAD - PS - 1;
  if (start_ok) {
    for (int k = lane; k < WIN * WIN; k += 64) {
      const int wy = k / WIN, wx = k % WIN;
      const size_t src = ((size_t)clampi(wy0 + wy, a.g.th) * tw + clampi(wx0 + wx, tw)) * NOC;
#pragma unroll
      for (int c = 0; c < NOC; ++c) win[k * NOC + c] = I1[src + c];
    }
  }

  bool first = start_ok;
  while (first || !conv) {
    if (!first) {
      cnt++;
      // projection on the steepest-descent images (:178-179) and 2x2 LLT solve (:184)
      float b0 = wave_sum(lane_dot(Tx, r));
      float b1 = wave_sum(lane_dot(Ty, r));
      const float y0 = b0 / l00;
      const float y1 = (b1 - l10 * y0) / l11;
      const float x1 = y1 / l11;
      const float x0 = (y0 - l10 * x1) / l00;
      dp0 = x0; dp1 = x1;
      p0 -= dp0; p1 -= dp1;                              // :186
      ptx = rx + p0; pty = ry + p1;
      const float ddx = stx - ptx, ddy = sty - pty;
      const bool bad = !(isfinite(dp0) && isfinite(dp1));  // oracle definition D3
      if (bad || sqrtf(ddx * ddx + ddy * ddy) > a.outlier ||   // :199-208
          ptx < a.g.lb || pty < a.g.lb || ptx > a.g.ubw || pty > a.g.ubh) {
        p0 = pin0; p1 = pin1; ptx = rx + p0; pty = ry + p1;
        conv = true;
        if (bad) { dp0 = 0.f; dp1 = 0.f; }
      }
    }
    first = false;
    // ---- OptimizeComputeErrImg (:264-284): bilinear query patch (:335-402), mean, residual ----
    {
      int pos0 = (int)ceilf(ptx + .00001f), pos1 = (int)ceilf(pty + .00001f);
      const int pos2 = (int)floorf(ptx), pos3 = (int)floorf(pty);
      const float r0 = ptx - (float)pos2, r1 = pty - (float)pos3;
      const float we0 = r0 * r1, we1 = (1 - r0) * r1, we2 = r0 * (1 - r1), we3 = (1 - r0) * (1 - r1);
      pos0 += PAD - wx0; pos1 += PAD - wy0;             // window coordinates
      float q[NSLOT * NOC];
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
        const int ia = ((pos1 + offy[s]) * WIN + (pos0 + offx[s])) * NOC;
        const int ic = ia - WIN * NOC;
#pragma unroll
        for (int c = 0; c < NOC; ++c) {
          if (have[s]) {
            const float va = win[ia + c], vb = win[ia - NOC + c], vc = win[ic + c], vd = win[ic - NOC + c];
            q[s * NOC + c] = we0 * va + we1 * vb + we2 * vc + we3 * vd;
          } else q[s * NOC + c] = 0.f;
        }
      }
      if (a.patnorm > 0) {
        const float m = wave_sum(lane_sum(q)) / (float)NV;
#pragma unroll
        for (int k = 0; k < NSLOT * NOC; ++k) q[k] -= m;
      }
#pragma unroll
      for (int k = 0; k < NSLOT * NOC; ++k) { r[k] = q[k] - T[k]; wabs[k] = fabsf(r[k]); }   // :230-236
      const float dpn = dp0 * dp0 + dp1 * dp1;           // :272
      if (cnt == 1) dpn_init = dpn;
      mares_old = mares;
      mares = wave_sum(lane_sum(wabs)) / (float)NV;      // :278
      // :279-282 (the two rate tests only matter once cnt >= min_iter; skip their divisions before that)
      bool go = (cnt < a.max_iter) & (mares > a.res_thresh);
      if (go && cnt >= a.min_iter) go = (dpn / dpn_init >= a.dp_thresh_sq) & (mares / mares_old <= a.dr_thresh);
      if (!go) conv = true;
      if (trace && lane == 0 && cnt <= a.max_iter) {
        trace[cnt * 4 + 0] = p0; trace[cnt * 4 + 1] = p1; trace[cnt * 4 + 2] = mares; trace[cnt * 4 + 3] = (float)cnt;
      }
    }
  }

  // ---- results ----
  const size_t pbase = (size_t)pair * a.g.nop + ip;
  if (lane == 0) {
    a.p_iter[pbase * 2] = p0;
    a.p_iter[pbase * 2 + 1] = p1;
    if (a.cnt) a.cnt[pbase] = cnt;
    if (a.hes) { a.hes[pbase * 3] = h00; a.hes[pbase * 3 + 1] = h01; a.hes[pbase * 3 + 2] = h11; }
  }
#pragma unroll
  for (int s = 0; s < NSLOT; ++s)
    if (have[s]) {
      const size_t e = pbase * NV + (size_t)(s * 64 + lane) * NOC;
#pragma unroll
      for (int c = 0; c < NOC; ++c) {
        a.pweight[e + c] = wabs[s * NOC + c];
        if (a.tmpl) { a.tmpl[e + c] = T[s * NOC + c]; a.tdx[e + c] = Tx[s * NOC + c]; a.tdy[e + c] = Ty[s * NOC + c]; }
      }
    }
}

}  // namespace fotg
