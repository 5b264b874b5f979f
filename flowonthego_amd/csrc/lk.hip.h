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
// VGPRs for the whole loop; the reductions of an iteration (query mean; two projections and the L1
// residual) are packed wave butterflies (wave_sum_multi) whose order is the oracle's dis_sum().  All lanes carry
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
  int camlr;                          // depth mode: 0 displacement <= 0 (forward grid), 1 displacement >= 0 (oflow.cpp:153,157)
  int max_iter, min_iter, patnorm, costfct;
  float dp_thresh_sq, dr_thresh, res_thresh, outlier, huber_bsq, huber_2bsq;
};

// NP patches per wave.  The per-pixel work (template, bilinear query patch, residual, wave reductions) runs patch
// after patch with one pixel per lane as described above.  Everything that is a per-patch SCALAR in the reference --
// the 2x2 Cholesky solve, the position update, the outlier / border tests, the termination tests -- would cost the
// same wave instructions whether one lane or 64 need the result, and it is most of the loop (four IEEE divisions and
// a square root are ~60 instructions).  So the scalars of the NP patches are PACKED: lanes [k*64/NP, (k+1)*64/NP)
// carry the state of patch k, the scalar code runs once per iteration for all NP patches, and only the five values
// the pixel work needs (window index, four bilinear weights) are read back per patch with v_readlane.
//
// DEPTH: stereo depth mode (SELECTMODE 2): the parameter is ONE horizontal displacement per patch -- scalar Hessian
// sum(Tx^2) (patch.cpp:83-87), one projection (:181), sign clamp after the update (:188-193), pt_iter.y stays at the
// reference row (:218-220); flow_prev has one channel (patchgrid.cpp:207-208).  p_iter keeps two slots (second = 0).
template <int PS, int NOC, int NP, bool DEPTH = false>
__global__ __launch_bounds__(256) void lk_kernel(LkArgs a)
{
  constexpr int NPIX = PS * PS;
  constexpr int NSLOT = (NPIX + 63) / 64;
  constexpr int NE = NSLOT * NOC;
  constexpr int NV = NPIX * NOC;
  constexpr int PAD = PS;
  constexpr int WIN = 2 * PS + 4;                    // window edge, see the column bound below
  constexpr int G = 64 / NP;                         // lanes per patch in the packed domain
  __shared__ float win_all[4][NP][WIN * WIN * NOC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gid = lane / G;
  const WgId wg = xcd_local_wg();                    // all patches of a pair on the XCD of its refinement workgroup
  const int ipw = (wg.x * 4 + wave) * NP;            // first patch of this wave
  if (ipw >= a.g.nop) return;                        // wave-uniform
  const int pair = wg.y;
  const int tw = a.g.tw;
  const float *I0 = a.I0 + (size_t)pair * a.img_stride;
  const float *I0x = a.I0x + (size_t)pair * a.img_stride;
  const float *I0y = a.I0y + (size_t)pair * a.img_stride;
  const float *I1 = a.I1 + (size_t)pair * a.img_stride;

  // per-lane pixel offsets inside the patch (the same for every patch)
  int offx[NSLOT], offy[NSLOT], poff[NSLOT];
  bool have[NSLOT];
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) {
    const int q = s * 64 + lane;
    have[s] = q < NPIX;
    offy[s] = q / PS - PS / 2;
    offx[s] = q % PS - PS / 2;
    poff[s] = (offy[s] * WIN + offx[s]) * NOC;
  }
  auto lane_sum = [&](const float *v) {
    float acc = v[0];
#pragma unroll
    for (int k = 1; k < NE; ++k)
      if (have[k / NOC]) acc = acc + v[k];
    return acc;
  };
  auto lane_dot = [&](const float *x, const float *y) {
    float acc = x[0] * y[0];
#pragma unroll
    for (int k = 1; k < NE; ++k)
      if (have[k / NOC]) acc = acc + x[k] * y[k];
    return acc;
  };
  // (the empty asm keeps every step a v_cndmask: left alone, the compiler turns the select chain of NP = 4 into a load from a
  // private array indexed by gid -- scratch stores and loads in every iteration, 13x the kernel's algorithmic HBM traffic)
  auto packf = [&](const float (&v)[NP]) {
    float x = v[0];
#pragma unroll
    for (int k = 1; k < NP; ++k) { x = gid == k ? v[k] : x; asm volatile("" : "+v"(x)); }
    return x;
  };
  auto packi = [&](const int (&v)[NP]) {
    int x = v[0];
#pragma unroll
    for (int k = 1; k < NP; ++k) { x = gid == k ? v[k] : x; asm volatile("" : "+v"(x)); }
    return x;
  };
  auto getf = [&](float x, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), k * G)); };
  auto geti = [&](int x, int k) { return __builtin_amdgcn_readlane(x, k * G); };

  // ---- per patch: template + gradients at round(pt_ref)+pad (patch.cpp:287-332), Hessian sums (:74-77),
  //      starting flow (patchgrid.cpp:195-211), I1 window -> LDS
  float T[NP][NE], Tx[NP][NE], Ty[NP][NE], r[NP][NE], wabs[NP][NE];
  float h00u[NP], h01u[NP], h11u[NP], rxu[NP], ryu[NP], pin0u[NP], pin1u[NP];
  int validu[NP], wx0u[NP], wy0u[NP], ipu[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int ip = (ipw + k < a.g.nop) ? ipw + k : a.g.nop - 1;     // surplus slots of the last wave shadow the last patch
    validu[k] = ipw + k < a.g.nop;
    ipu[k] = ip;
    // patch id -> reference position (patchgrid.cpp:57-66: i = x*noph + y)
    const int gx = ip / a.g.noph, gy = ip % a.g.noph;
    const float rx = (float)(gx * a.g.steps + a.g.offw), ry = (float)(gy * a.g.steps + a.g.offh);
    rxu[k] = rx; ryu[k] = ry;
    const int px = (int)rx + PAD, py = (int)ry + PAD;              // pt_ref is integer valued: round() is exact
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const size_t idx = ((size_t)(px + offx[s]) + (size_t)(py + offy[s]) * tw) * NOC;
#pragma unroll
      for (int c = 0; c < NOC; ++c) {
        T[k][s * NOC + c] = have[s] ? I0[idx + c] : 0.f;
        Tx[k][s * NOC + c] = have[s] ? I0x[idx + c] : 0.f;
        Ty[k][s * NOC + c] = have[s] ? I0y[idx + c] : 0.f;
        r[k][s * NOC + c] = 0.f; wabs[k][s * NOC + c] = 0.f;
      }
    }
    float pin0 = 0.f, pin1 = 0.f;
    if (a.flow_prev) {
      int fx = (int)floorf(rx / 2), fy = (int)floorf(ry / 2);
      // oracle definition D5: a level of odd size has patches whose half coordinate is one past the coarser array (only
      // reachable with `initflow`); the reference reads out of bounds there, here the index is clamped
      fx = fx > a.g.w / 2 - 1 ? a.g.w / 2 - 1 : fx;
      fy = fy > a.g.h / 2 - 1 ? a.g.h / 2 - 1 : fy;
      const float *fp = a.flow_prev + (size_t)pair * a.flow_prev_stride + (DEPTH ? 1 : 2) * (size_t)(fy * (a.g.w / 2) + fx);
      pin0 = fp[0] * 2;
      if constexpr (!DEPTH) pin1 = fp[1] * 2;
    }
    pin0u[k] = pin0; pin1u[k] = pin1;
    // Stage the reachable window of I1 in LDS.  Every evaluated position is within ps/2 of the start in x and y, so
    // the bilinear taps span columns floor(stx)-ps-1 .. floor(stx)+ps+2 (padded coordinates: + PAD); rows likewise.
    const float stx = rx + pin0, sty = ry + pin1;
    const bool ok = !(stx < a.g.lb || sty < a.g.lb || stx > a.g.ubw || sty > a.g.ubh);
    wx0u[k] = (int)floorf(stx) + PAD - PS - 1; wy0u[k] = (int)floorf(sty) + PAD - PS - 1;
    if (ok) {
      float *win = win_all[wave][k];
      for (int t = lane; t < WIN * WIN; t += 64) {
        const int wy = t / WIN, wx = t % WIN;
        const size_t src = ((size_t)clampi(wy0u[k] + wy, a.g.th) * tw + clampi(wx0u[k] + wx, tw)) * NOC;
#pragma unroll
        for (int c = 0; c < NOC; ++c) win[t * NOC + c] = I1[src + c];
      }
    }
  }
  // the window is filled by all lanes of the wave and read by all of them: keep the compiler (which reasons per thread) from
  // moving reads above the fill; the hardware executes a wave's LDS accesses in order
  asm volatile("" ::: "memory");

  // template mean (patch.cpp:330-331) and Hessian sums (:74-77, depth :84) of all NP patches, reduced together
  if (a.patnorm > 0) {
    float ms[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) ms[k] = lane_sum(T[k]);
    wave_sum_multi<NP>(ms);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const float m = ms[k] / (float)NV;
#pragma unroll
      for (int e = 0; e < NE; ++e) T[k][e] -= m;
    }
  }
  {
    constexpr int NH = DEPTH ? 1 : 3;
    float hs[NH * NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      hs[NH * k] = lane_dot(Tx[k], Tx[k]);
      if constexpr (!DEPTH) { hs[NH * k + 1] = lane_dot(Tx[k], Ty[k]); hs[NH * k + 2] = lane_dot(Ty[k], Ty[k]); }
    }
    wave_sum_multi<NH * NP>(hs);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      h00u[k] = hs[NH * k];
      h01u[k] = DEPTH ? 0.f : hs[NH * k + (DEPTH ? 0 : 1)];
      h11u[k] = DEPTH ? 0.f : hs[NH * k + (DEPTH ? 0 : 2)];
    }
  }

  // ---- packed per-patch state ----
  float H00 = packf(h00u), H11 = packf(h11u);
  const float H01 = packf(h01u);
  if constexpr (DEPTH) {
    if (H00 == 0.f) H00 = (float)((double)H00 + 1e-10);  // :85-86
  } else if (H00 * H11 - H01 * H01 == 0.f) {             // :78-82  (float += 1e-10 in double, like the reference)
    H00 = (float)((double)H00 + 1e-10);
    H11 = (float)((double)H11 + 1e-10);
  }
  // Cholesky factor of the (constant) Hessian, hoisted out of the loop: same values every iteration
  const float L00 = sqrtf(H00);
  const float L10 = DEPTH ? 0.f : H01 / L00;
  const float L11 = DEPTH ? 1.f : sqrtf(H11 - L10 * L10);
  const float RX = packf(rxu), RY = packf(ryu), PIN0 = packf(pin0u), PIN1 = packf(pin1u);
  const int WX0 = packi(wx0u), WY0 = packi(wy0u), IP = packi(ipu);
  const bool VALID = packi(validu) != 0;
  // OptimizeStart (patch.cpp:120-156)
  float P0 = PIN0, P1 = PIN1, PTX = RX + P0, PTY = RY + P1;
  const float STX = PTX, STY = PTY;
  const bool START_OK = VALID && !(PTX < a.g.lb || PTY < a.g.lb || PTX > a.g.ubw || PTY > a.g.ubh);
  bool CONV = !START_OK;                                 // :135-141; pweight stays 0 (oracle definition D2)
  int CNT = 0;
  float DP0 = 0.f, DP1 = 0.f, DPN_INIT = 1e-10f, MARES = 1e5f, MARES_OLD = 1e20f;
  const int trow = (a.max_iter + 1) * 4;
  float *trace = (a.trace && pair == 0) ? a.trace : nullptr;
  if (trace && VALID && (lane % G) < 4)
    for (int t = 0; t <= a.max_iter; ++t) trace[(size_t)IP * trow + t * 4 + (lane % G)] = 0.f;

  // ---- OptimizeComputeErrImg (:264-284) for the patches in ACT: bilinear query patch (:335-402), mean, residual, and the
  //      three sums every patch needs next -- the two projections on the steepest-descent images (:178-179, used by the NEXT
  //      update) and the L1 residual (:278) -- reduced together for all NP patches (wave_sum_multi)
  float B0 = 0.f, B1 = 0.f;
  auto eval = [&](const bool ACT) {
    const int pos2 = (int)floorf(PTX), pos3 = (int)floorf(PTY);
    const int pos0 = (int)ceilf(PTX + .00001f) + PAD - WX0, pos1 = (int)ceilf(PTY + .00001f) + PAD - WY0;   // window coordinates
    const float r0 = PTX - (float)pos2, r1 = PTY - (float)pos3;
    const float WE0 = r0 * r1, WE1 = (1 - r0) * r1, WE2 = r0 * (1 - r1), WE3 = (1 - r0) * (1 - r1);
    const int IA = (pos1 * WIN + pos0) * NOC;
    const int ACTI = ACT ? 1 : 0;
    float q[NP][NE], ms[NP];
    bool act[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      act[k] = geti(ACTI, k) != 0;                         // wave-uniform
#pragma unroll
      for (int e = 0; e < NE; ++e) q[k][e] = 0.f;
      if (act[k]) {
        const float *win = win_all[wave][k];
        const int iab = geti(IA, k);
        const float we0 = getf(WE0, k), we1 = getf(WE1, k), we2 = getf(WE2, k), we3 = getf(WE3, k);
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
          const int ia = iab + poff[s];
          const int ic = ia - WIN * NOC;
#pragma unroll
          for (int c = 0; c < NOC; ++c) {
            if (have[s]) {
              const float va = win[ia + c], vb = win[ia - NOC + c], vc = win[ic + c], vd = win[ic - NOC + c];
              q[k][s * NOC + c] = we0 * va + we1 * vb + we2 * vc + we3 * vd;
            }
          }
        }
      }
      ms[k] = lane_sum(q[k]);
    }
    if (a.patnorm > 0) {
      wave_sum_multi<NP>(ms);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const float m = ms[k] / (float)NV;
#pragma unroll
        for (int e = 0; e < NE; ++e) q[k][e] -= m;
      }
    }
    constexpr int NR = DEPTH ? 2 : 3;
    float red[NR * NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      if (act[k]) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          float d = q[k][e] - T[k][e];                           // :230-236 L2: the difference image itself
          if (a.costfct == 1) d = copysignf(sqrtf(fabsf(d)), d);                                      // :238-246 L1
          else if (a.costfct == 2) d = copysignf(sqrtf((sqrtf(1.0f + (d * d) / a.huber_bsq) - 1.0f) * a.huber_2bsq), d);   // :247-261
          r[k][e] = d; wabs[k][e] = fabsf(d);
        }
      }
      red[NR * k] = lane_dot(Tx[k], r[k]);
      if constexpr (!DEPTH) red[NR * k + 1] = lane_dot(Ty[k], r[k]);
      red[NR * k + NR - 1] = lane_sum(wabs[k]);
    }
    wave_sum_multi<NR * NP>(red);
    float b0u[NP], b1u[NP], maresu[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      b0u[k] = red[NR * k];
      b1u[k] = DEPTH ? 0.f : red[NR * k + (DEPTH ? 0 : 1)];
      maresu[k] = act[k] ? red[NR * k + NR - 1] / (float)NV : 0.f;   // :278
    }
    B0 = packf(b0u); B1 = packf(b1u);
    const float MN = packf(maresu);
    if (ACT) {
      const float dpn = DP0 * DP0 + DP1 * DP1;           // :272
      if (CNT == 1) DPN_INIT = dpn;
      MARES_OLD = MARES;
      MARES = MN;
      // :279-282 (the two rate tests only matter once cnt >= min_iter)
      bool go = (CNT < a.max_iter) & (MARES > a.res_thresh);
      if (go && CNT >= a.min_iter) go = (dpn / DPN_INIT >= a.dp_thresh_sq) & (MARES / MARES_OLD <= a.dr_thresh);
      if (!go) CONV = true;
      if (trace && (lane % G) == 0 && CNT <= a.max_iter) {
        float *tr = trace + (size_t)IP * trow + CNT * 4;
        tr[0] = P0; tr[1] = P1; tr[2] = MARES; tr[3] = (float)CNT;
      }
    }
  };

  eval(START_OK);                                        // OptimizeStart's first error image (:154)
  while (__builtin_amdgcn_ballot_w64(!CONV) != 0) {
    const bool ACT = !CONV;                              // patches that run this iteration
    {
      // 2x2 LLT solve (:184), packed; depth mode: the 1x1 system, L = sqrt(H)
      const float y0 = B0 / L00;
      float x0, x1;
      if constexpr (DEPTH) { x0 = y0 / L00; x1 = 0.f; }
      else {
        const float y1 = (B1 - L10 * y0) / L11;
        x1 = y1 / L11;
        x0 = (y0 - L10 * x1) / L00;
      }
      float nP0 = P0 - x0, nP1 = P1 - x1;                // :186
      if constexpr (DEPTH) {                             // :188-193 std::min / std::max with 0
        if (a.camlr == 0) nP0 = (0.0f < nP0) ? 0.0f : nP0;
        else nP0 = (nP0 < 0.0f) ? 0.0f : nP0;
      }
      float nPTX = RX + nP0, nPTY = RY + nP1;
      const float ddx = STX - nPTX, ddy = STY - nPTY;
      const bool bad = !(isfinite(x0) && isfinite(x1));  // oracle definition D3
      const bool reset = bad || sqrtf(ddx * ddx + ddy * ddy) > a.outlier ||   // :199-208
                         nPTX < a.g.lb || nPTY < a.g.lb || nPTX > a.g.ubw || nPTY > a.g.ubh;
      if (reset) { nP0 = PIN0; nP1 = PIN1; nPTX = RX + PIN0; nPTY = RY + PIN1; }
      if (ACT) {
        CNT++;
        DP0 = (reset && bad) ? 0.f : x0; DP1 = (reset && bad) ? 0.f : x1;
        P0 = nP0; P1 = nP1; PTX = nPTX; PTY = nPTY;
        if (reset) CONV = true;
      }
    }
    eval(ACT);
  }

  // ---- results ----
  if (VALID && (lane % G) == 0) {
    const size_t pb = (size_t)pair * a.g.nop + IP;
    a.p_iter[pb * 2] = P0;
    a.p_iter[pb * 2 + 1] = P1;
    if (a.cnt) a.cnt[pb] = CNT;
    if (a.hes) { a.hes[pb * 3] = H00; a.hes[pb * 3 + 1] = H01; a.hes[pb * 3 + 2] = H11; }
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (!validu[k]) continue;
    const size_t pbase = (size_t)pair * a.g.nop + ipu[k];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s)
      if (have[s]) {
        const size_t e = pbase * NV + (size_t)(s * 64 + lane) * NOC;
#pragma unroll
        for (int c = 0; c < NOC; ++c) {
          a.pweight[e + c] = wabs[k][s * NOC + c];
          if (a.tmpl) { a.tmpl[e + c] = T[k][s * NOC + c]; a.tdx[e + c] = Tx[k][s * NOC + c]; a.tdy[e + c] = Ty[k][s * NOC + c]; }
        }
      }
  }
}

}  // namespace fotg
