// lk_fast.hip.h -- the tolerance mode of the patch loop (fotg_params::fast_math): the same inverse-compositional Lucas-Kanade
// iteration as lk.hip.h (kroeger/patch.cpp:120-212, 264-284, 335-402), evaluated with a third of the instructions.  NOT
// bit-identical to the oracle: algebraically the same update, rounded differently (measured against the oracle by the tests;
// the north star's bound is a mean endpoint error of 1e-3 px).  The default (parity) mode never runs this file's kernel.
//
// What changes against lk.hip.h (one row of LPP lanes per patch there and here; everything per-patch is carried redundantly by
// the patch's lanes, the reductions leave their result in every lane):
//  * the projections do not need the residual.  With r = (q - mean q) - (T - mean T)  (patch.cpp:230-236, 278, 330-331):
//        sum Tx r  =  sum (Tx - mean Tx) q  -  sum (Tx - mean Tx) (T - mean T)
//    -- the second term is a constant of the patch and the first needs neither the mean of the query patch nor the template:
//    one bilinear sample and one fused multiply-add per pixel and projection, two reductions per iteration instead of four,
//    no subtraction pass.  The L1 residual (`mares`) only feeds termination tests that cannot fire when min_iter == max_iter and
//    res_thresh <= 0 (patch.cpp:279-282; every operating point): it is not evaluated in the loop.  The residual itself -- the patch
//    weights of the densification are |r| -- is evaluated ONCE, at the final position.  (Configurations with min_iter < max_iter,
//    res_thresh > 0, another cost function or depth mode run the exact kernel also with fast_math set.)
//  * the 2x2 Cholesky solve (patch.cpp:184) becomes a multiplication with the inverse H^-1 = L^-T L^-1 computed once from the same
//    factor, the constant term folded in: dp = H^-1 S + k, four fused multiply-adds, no division in the loop.
//  * a lane owns a BH x BW BLOCK of the patch (3 x 3 at ps 12, 2 x 2 at ps 8: sixteen lanes per patch) instead of every 16th
//    pixel: the block's bilinear taps are a (BH+1) x (BW+1) window of the staged I1 -- 16 LDS values for 9 pixels instead of 36,
//    all at compile-time offsets from ONE per-lane address.
//  * the window in LDS holds I1 - mean T: the constant does not change the projections (the centred gradients sum to zero) and
//    keeps the products small (less cancellation against the constant term).
//  * fused multiply-adds throughout, the two projections as one packed-f32 accumulator, reductions in DPP order.
// Same semantics kept exactly: start test, window placement, the ceil(x + 1e-5) / floor corner pair, outlier / border / non-finite
// reset (patch.cpp:199-208, oracle definition D3), iteration count, p_iter / pweight layout.
#pragma once
#include "lk.hip.h"

namespace fotg {

typedef float lkf_v2f __attribute__((ext_vector_type(2)));

template <int LPP>
__device__ __forceinline__ float lkf_allsum(float v)
{
  if constexpr (LPP == 4) {
#define FOTG_DPPQ(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, false))
    v = v + FOTG_DPPQ(v, 0x4E);                          // quad_perm [2,3,0,1]
    v = v + FOTG_DPPQ(v, 0xB1);                          // quad_perm [1,0,3,2]
#undef FOTG_DPPQ
    return v;
  } else return group_allsum<LPP>(v);
}

// two sums at once, each left in every lane of its group of LPP lanes.  Hand-placed DPP adds (v = v + rotate(v)): the compiler
// turns two interleaved reduction chains into packed adds fed by v_mov_dpp copies, 20 instructions instead of 8.  A DPP read of a
// register a VALU instruction has just written needs two wait states (the assembler does not see into this block: s_nop).
template <int LPP>
__device__ __forceinline__ void lkf_allsum2(float &a, float &b)
{
  if constexpr (LPP == 16) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf"
        : "+v"(a), "+v"(b));
  } else {
    a = lkf_allsum<LPP>(a);
    asm volatile("" : "+v"(b));
    b = lkf_allsum<LPP>(b);
  }
}

constexpr int lkf_win(int ps, int r) { return ps + 2 * (r > 0 ? r : ps / 2 + 1) + 2; }
constexpr int lkf_min_waves(int ps, int noc, int lpp, int r)
{
  // what the private windows leave room for in the CU's 160 KB of LDS (waves per SIMD)
  const int bytes = (64 / lpp) * lkf_win(ps, r) * lkf_win(ps, r) * noc * 4;
  const int per_cu = 160 * 1024 / bytes;
  return per_cu >= 16 ? 4 : per_cu >= 12 ? 3 : per_cu >= 8 ? 2 : 1;
}

// R: radius of the staged window.  0: the whole reachable region (a patch is reset when it has moved more than ps/2, so every
// evaluation is inside: (2ps+4)^2 values per patch).  R > 0: positions within R pixels of the start -- where patches started from
// the coarser level's flow end up -- come out of a (ps + 2R + 2)^2 window (44 % of the LDS at ps 12 and R = 2: twice the patches
// per wave at the same occupancy, and less to stage); an iteration in which any running patch of the wave is outside reads its
// taps from the level image instead (same clamped coordinates, same values: the two paths are bit-identical, tested by forcing
// the second one).
template <int PS, int NOC, int LPP, int R = 0>
__global__ __launch_bounds__(64, lkf_min_waves(PS, NOC, LPP, R)) void lk_fast_kernel(LkArgs a)
{
  constexpr int PPW = 64 / LPP;                      // patches per wave
  constexpr int LC = LPP == 16 ? 4 : 2, LR = LPP / LC;   // lanes of a patch as LR x LC blocks: 16 = 4 x 4, 8 = 4 x 2, 4 = 2 x 2
  constexpr int BH = PS / LR, BW = PS / LC;          // block of one lane
  static_assert(PS % LR == 0 && PS % LC == 0, "blocks tile the patch");
  constexpr int NE = BH * BW * NOC;
  constexpr int NV = PS * PS * NOC;
  constexpr int PAD = PS;
  constexpr int RR = R > 0 ? R : PS / 2 + 1;         // positions up to RR from the start are inside the window
  constexpr int WIN = lkf_win(PS, R);                // window edge: ps + 2 RR + 2 (lk.hip.h: 2 ps + 4)
  constexpr int WROW = WIN * NOC;                    // floats per window row
  __shared__ float win_all[PPW * WIN * WIN * NOC];
  const int lane = threadIdx.x & 63, row = lane / LPP, j = lane % LPP;
  const int by = (j / LC) * BH, bx = (j % LC) * BW;  // the lane's block inside the patch
  WgId wg = xcd_local_wg();
  if (a.nwg > 0) { wg.x = xcd_banded_x(a.nwg); if (wg.x < 0) return; }      // (see lk_kernel)
  const int ipw = wg.x * PPW;
  const int pair = wg.y;
  const int tw = a.g.tw;
  const float *I0 = a.I0 + (size_t)pair * a.img_stride;
  const float *I0x = a.I0x + (size_t)pair * a.img_stride;
  const float *I0y = a.I0y + (size_t)pair * a.img_stride;
  const float *I1 = a.I1 + (size_t)pair * a.img_stride;
  const bool VALID = ipw + row < a.g.nop;
  const int IP = VALID ? ipw + row : a.g.nop - 1;
  const int gx = IP / a.g.noph, gy = IP % a.g.noph;    // patchgrid.cpp:57-66
  const float RX = (float)(gx * a.g.steps + a.g.offw), RY = (float)(gy * a.g.steps + a.g.offh);
  const bool PN = a.patnorm > 0;

  // ---- template block: centred gradients (Tx - mean Tx, Ty - mean Ty) as pairs, Hessian of the raw gradients (patch.cpp:74-77),
  //      constant terms C = sum (Tx - mean Tx)(T - mean T), inverse Hessian, K = -H^-1 C
  const size_t tbase = ((size_t)((int)RX + PAD + bx - PS / 2) + (size_t)((int)RY + PAD + by - PS / 2) * tw) * NOC;
  lkf_v2f G[NE];
  float MT = 0.f, IH00, IH01, IH11, K0, K1;
  {
    float t[NE];
    float sT = 0.f, sX = 0.f, sY = 0.f, hxx = 0.f, hxy = 0.f, hyy = 0.f;
#pragma unroll
    for (int dy = 0; dy < BH; ++dy)
#pragma unroll
      for (int k = 0; k < BW * NOC; ++k) {
        const size_t idx = tbase + (size_t)dy * tw * NOC + k;
        const int e = dy * BW * NOC + k;
        const float tv = I0[idx], gxv = I0x[idx], gyv = I0y[idx];
        t[e] = tv; G[e].x = gxv; G[e].y = gyv;
        sT += tv; sX += gxv; sY += gyv;
        hxx = __builtin_fmaf(gxv, gxv, hxx); hxy = __builtin_fmaf(gxv, gyv, hxy); hyy = __builtin_fmaf(gyv, gyv, hyy);
      }
    float H00 = lkf_allsum<LPP>(hxx), H01 = lkf_allsum<LPP>(hxy), H11 = lkf_allsum<LPP>(hyy);
    float mX = 0.f, mY = 0.f;
    if (PN) {
      constexpr float inv = 1.0f / (float)NV;
      MT = lkf_allsum<LPP>(sT) * inv; mX = lkf_allsum<LPP>(sX) * inv; mY = lkf_allsum<LPP>(sY) * inv;
    }
    float c0 = 0.f, c1 = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      G[e].x -= mX; G[e].y -= mY;
      const float tc = t[e] - MT;
      c0 = __builtin_fmaf(G[e].x, tc, c0); c1 = __builtin_fmaf(G[e].y, tc, c1);
    }
    const float C0 = lkf_allsum<LPP>(c0), C1 = lkf_allsum<LPP>(c1);
    // :78-82  (float += 1e-10 in double, like the reference and lk.hip.h)
    if (H00 * H11 - H01 * H01 == 0.f) { H00 = (float)((double)H00 + 1e-10); H11 = (float)((double)H11 + 1e-10); }
    if (a.hes && VALID && j == 0) {
      float *hp = a.hes + ((size_t)pair * a.g.nop + IP) * 3;
      hp[0] = H00; hp[1] = H01; hp[2] = H11;
    }
    // H^-1 = L^-T L^-1 from the Cholesky factor the reference solves with (patch.cpp:184); a singular / indefinite H gives
    // non-finite entries, i.e. a non-finite update: the patch is reset at its first iteration, as in the exact kernel
    const float L00 = sqrtf(H00), L10 = H01 / L00, L11 = sqrtf(H11 - L10 * L10);
    const float ia = 1.0f / L00, ib = 1.0f / L11, m = -L10 * ia * ib;
    IH00 = ia * ia + m * m; IH01 = m * ib; IH11 = ib * ib;
    K0 = -(IH00 * C0 + IH01 * C1); K1 = -(IH01 * C0 + IH11 * C1);          // dp = H^-1 (S - C) = H^-1 S + K
  }

  // ---- starting flow (patchgrid.cpp:195-211), OptimizeStart (patch.cpp:120-156): as lk.hip.h
  float PIN0 = 0.f, PIN1 = 0.f;
  if (a.flow_prev) {
    int fx = (int)floorf(RX / 2), fy = (int)floorf(RY / 2);
    fx = fx > a.g.w / 2 - 1 ? a.g.w / 2 - 1 : fx;        // oracle definition D5
    fy = fy > a.g.h / 2 - 1 ? a.g.h / 2 - 1 : fy;
    const float *fp = a.flow_prev + (size_t)pair * a.flow_prev_stride + 2 * (size_t)(fy * (a.g.w / 2) + fx);
    PIN0 = fp[0] * 2; PIN1 = fp[1] * 2;
  }
  float P0 = PIN0, P1 = PIN1, PTX = RX + P0, PTY = RY + P1;
  const float STX = PTX, STY = PTY;
  const bool START_OK = VALID && !(PTX < a.g.lb || PTY < a.g.lb || PTX > a.g.ubw || PTY > a.g.ubh);
  // window origin (padded coordinates): the leftmost tap of a position RR left of the start, floor(stx) - RR - ps/2
  const float FSX = floorf(STX), FSY = floorf(STY);
  const int WX0 = (int)FSX + PAD - RR - PS / 2, WY0 = (int)FSY + PAD - RR - PS / 2;
  if (START_OK) {
    // the window of I1 minus the template mean: a lane takes columns j, j + LPP, .. of every row (clamped like the level's border)
    float *const wdst = win_all + row * (WIN * WIN * NOC);
    constexpr int NCOL = (WIN + LPP - 1) / LPP;
    int cofs[NCOL];
#pragma unroll
    for (int k = 0; k < NCOL; ++k) cofs[k] = clampi(WX0 + j + k * LPP, tw) * NOC;
    for (int wy = 0; wy < WIN; ++wy) {
      const float *rowp = I1 + (size_t)clampi(WY0 + wy, a.g.th) * tw * NOC;
#pragma unroll
      for (int k = 0; k < NCOL; ++k)
        if (j + k * LPP < WIN)
#pragma unroll
          for (int c = 0; c < NOC; ++c) wdst[(wy * WIN + j + k * LPP) * NOC + c] = rowp[cofs[k] + c] - MT;
    }
  }
  asm volatile("" ::: "memory");      // (the window is filled by the lanes of the row and read by all of them; LDS accesses of a wave execute in order)
  // per-lane window offset of the block's first upper-left tap at position (0, 0): pixel (by, bx) sits at patch offset
  // (by - PS/2, bx - PS/2), its upper-left tap one up and one left; window coordinates = padded coordinates - (WX0, WY0)
  const int laneoff = row * (WIN * WIN * NOC) + ((by - PS / 2 - 1 + PAD - WY0) * WIN + (bx - PS / 2 - 1 + PAD - WX0)) * NOC;

  // bilinear query block at (PTX, PTY) (patch.cpp:335-402), centred by mean T.  Two neighbouring values of a block row as one
  // packed-f32 operation: the pair (W[k], W[k+1]) and the pair one pixel to the right come from LDS as two 8-byte reads (every
  // value is read twice, no register shuffling), the four weights are broadcast operands.
  constexpr int RW = BW * NOC;                           // values per block row
  constexpr int NPR = RW / 2;                            // pairs per block row (+ one single value when RW is odd)
  typedef float v2u __attribute__((ext_vector_type(2), aligned(4)));
  // the four bilinear weights at (PTX, PTY) and the integer corner (ceil(x + 1e-5), ceil(y + 1e-5)) as floats
  // (plain scalars, not a struct: a struct handed through the lambdas below ended up in private memory)
#define FOTG_LKF_TAPS                                                                                                          \
  const float r0 = __builtin_amdgcn_fractf(PTX), r1 = __builtin_amdgcn_fractf(PTY); /* x - floor(x) (:344-345) in one instruction */ \
  /* ceil(x + 1e-5) keeps the reference's corner pair (:340-343); the window index is formed in floating point (exact) */     \
  const float cx = ceilf(PTX + .00001f), cy = ceilf(PTY + .00001f);                                                           \
  const float we0 = r0 * r1, we1 = __builtin_fmaf(-r0, r1, r1), we2 = __builtin_fmaf(-r0, r1, r0), we3 = (1.0f - r0) - we1
  // the block's (BH+1) x (BW+1) window through a pair loader (values k, k+1 of window row dy) and a single-value loader, then the
  // four-tap blend; one body for both sources, everything in registers (arrays local to this lambda: nothing goes to scratch)
  auto sample_with = [&](float we0, float we1, float we2, float we3, auto ld2, auto ld1, float (&q)[NE]) {
    const lkf_v2f w0 = {we0, we0}, w1 = {we1, we1}, w2 = {we2, we2}, w3 = {we3, we3};
    lkf_v2f A[BH + 1][NPR > 0 ? NPR : 1], Cc[BH + 1][NPR > 0 ? NPR : 1];
    float sA[BH + 1], sC[BH + 1];
#pragma unroll
    for (int dy = 0; dy <= BH; ++dy) {
#pragma unroll
      for (int m = 0; m < NPR; ++m) { A[dy][m] = ld2(dy, 2 * m); Cc[dy][m] = ld2(dy, 2 * m + NOC); }
      if constexpr (RW & 1) { sA[dy] = ld1(dy, RW - 1); sC[dy] = ld1(dy, RW - 1 + NOC); }
    }
#pragma unroll
    for (int dy = 0; dy < BH; ++dy) {
#pragma unroll
      for (int m = 0; m < NPR; ++m) {
        const lkf_v2f v = __builtin_elementwise_fma(w0, Cc[dy + 1][m], __builtin_elementwise_fma(w1, A[dy + 1][m], __builtin_elementwise_fma(w2, Cc[dy][m], w3 * A[dy][m])));
        q[dy * RW + 2 * m] = v.x; q[dy * RW + 2 * m + 1] = v.y;
      }
      if constexpr (RW & 1)
        q[dy * RW + RW - 1] = __builtin_fmaf(we0, sC[dy + 1], __builtin_fmaf(we1, sA[dy + 1], __builtin_fmaf(we2, sC[dy], we3 * sA[dy])));
    }
  };
  // from the staged window
  auto sample = [&](float (&q)[NE]) {
    FOTG_LKF_TAPS;
    int ib = (int)__builtin_fmaf(cy, (float)WROW, cx * (float)NOC) + laneoff;
    asm volatile("" : "+v"(ib));                          // (one address register, the taps are immediates of the LDS reads)
    const float *tp = win_all + ib;
    sample_with(we0, we1, we2, we3, [&](int dy, int k) -> lkf_v2f { return *reinterpret_cast<const v2u *>(tp + dy * WROW + k); },
                [&](int dy, int k) { return tp[dy * WROW + k]; }, q);
  };
  // (R > 0, rare, wave-uniform) the same taps from the level image, clamped like the staged copy
  auto sample_image = [&](float (&q)[NE]) {
    FOTG_LKF_TAPS;
    const int gx0 = (int)cx - 1 + PAD - PS / 2 + bx, gy0 = (int)cy - 1 + PAD - PS / 2 + by;
    auto px = [&](int dy, int k) { return I1[((size_t)clampi(gy0 + dy, a.g.th) * tw + clampi(gx0 + k / NOC, tw)) * NOC + k % NOC] - MT; };
    sample_with(we0, we1, we2, we3, [&](int dy, int k) -> lkf_v2f { return lkf_v2f{px(dy, k), px(dy, k + 1)}; }, px, q);
  };
#undef FOTG_LKF_TAPS
  // S = sum (Tx - mean Tx, Ty - mean Ty) q over the patch
  float S0 = 0.f, S1 = 0.f;
  auto accumulate = [&](const float (&q)[NE]) {
    lkf_v2f sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const lkf_v2f qq = {q[e], q[e]};
      if (e & 1) sb = __builtin_elementwise_fma(G[e], qq, sb);
      else sa = __builtin_elementwise_fma(G[e], qq, sa);
    }
    if constexpr (NE > 1) sa = sa + sb;
    S0 = sa.x; S1 = sa.y;
    lkf_allsum2<LPP>(S0, S1);
  };
  // far: a running patch of this lane's row sits outside the staged window (R > 0 only; the whole wave then reads the level image:
  // two complete code paths, so that nothing but the two sums is merged behind them)
  const bool FORCE_IMAGE = R > 0 && a.shw_test == 2;       // test tap
  auto project = [&](bool far) {
    float q[NE];
    if (R > 0 && __builtin_amdgcn_ballot_w64(far | FORCE_IMAGE) != 0) { sample_image(q); accumulate(q); }
    else { sample(q); accumulate(q); }
  };

  // The loop is free of divergent branches: every lane evaluates every iteration (a masked-out row costs the wave the same
  // instructions), the rows that are still running take the results; a row that never started or has been reset reads
  // whatever its position selects in LDS (out-of-range LDS reads return zeros) and drops it.
  bool ACT = START_OK && a.max_iter > 0;
  int CNT = 0;
  project(false);                                        // OptimizeStart's first error image (:154): at the start position
  for (int it = 1; it <= a.max_iter; ++it) {
    if (__builtin_amdgcn_ballot_w64(ACT) == 0) break;
    const float x0 = __builtin_fmaf(IH00, S0, __builtin_fmaf(IH01, S1, K0));
    const float x1 = __builtin_fmaf(IH01, S0, __builtin_fmaf(IH11, S1, K1));
    const float nP0 = P0 - x0, nP1 = P1 - x1;            // :186
    const float nPTX = RX + nP0, nPTY = RY + nP1;
    const float ddx = PIN0 - nP0, ddy = PIN1 - nP1;      // = pt_st - pt_iter (:199) without waiting for the new position
    // :199-208 and oracle definition D3 (a non-finite update resets the patch): the comparisons are false for NaN, so `keep`
    // is false for every non-finite position as well
    const bool keep = (__builtin_fmaf(ddx, ddx, ddy * ddy) <= a.outlier_sq) & (nPTX >= a.g.lb) & (nPTY >= a.g.lb) & (nPTX <= a.g.ubw) & (nPTY <= a.g.ubh);
    // |position - start| <= R on both axes is inside the staged window (a reset patch is back at the start)
    const bool far = R > 0 && keep && fmaxf(fabsf(ddx), fabsf(ddy)) > (float)R;
    P0 = ACT ? (keep ? nP0 : PIN0) : P0;
    P1 = ACT ? (keep ? nP1 : PIN1) : P1;
    CNT += ACT ? 1 : 0;
    ACT = ACT & keep & (it < a.max_iter);
    PTX = RX + P0; PTY = RY + P1;
    if (it < a.max_iter) project(ACT & far);             // (wave-uniform)
  }

  // ---- results: the residual at the final position = the patch weights of the densification (patchgrid.cpp:213-275)
  float r[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) r[e] = 0.f;               // patches that never started: oracle definition D2
  if (START_OK) {
    float q[NE];
    if (R > 0 && __builtin_amdgcn_ballot_w64((fmaxf(fabsf(PIN0 - P0), fabsf(PIN1 - P1)) > (float)R) | FORCE_IMAGE) != 0) sample_image(q);
    else sample(q);
    float mq = 0.f;
    if (PN) {
      float s = q[0];
#pragma unroll
      for (int e = 1; e < NE; ++e) s += q[e];
      mq = lkf_allsum<LPP>(s) * (1.0f / (float)NV);
    }
#pragma unroll
    for (int dy = 0; dy < BH; ++dy)
#pragma unroll
      for (int k = 0; k < BW * NOC; ++k) {
        const int e = dy * BW * NOC + k;
        r[e] = (q[e] - mq) - (I0[tbase + (size_t)dy * tw * NOC + k] - MT);
      }
  }
  if (!VALID) return;
  const size_t pb = (size_t)pair * a.g.nop + IP;
  if (j == 0) {
    a.p_iter[pb * 2] = P0;
    a.p_iter[pb * 2 + 1] = P1;
    if (a.cnt) a.cnt[pb] = CNT;
  }
#pragma unroll
  for (int dy = 0; dy < BH; ++dy)
#pragma unroll
    for (int k = 0; k < BW * NOC; ++k)
      a.pweight[pb * NV + (size_t)((by + dy) * PS + bx) * NOC + k] = fabsf(r[dy * BW * NOC + k]);
}

}  // namespace fotg
