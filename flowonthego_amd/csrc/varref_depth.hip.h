// varref_depth.hip.h -- variational refinement of one level in STEREO DEPTH mode (SELECTMODE 2):
// VarRefClass::RefLevelDE (kroeger/refine_variational.cpp:243-330) driving compute_smoothness,
// compute_data_DE (FDF1.0.1/opticalflow_aux.c:446-540), sub_laplacian and
// sor_coupled_slow_but_readable_DE (FDF1.0.1/solver.c:428-466).
//
// One displacement component: the system is scalar per pixel (a11, b1), the solver is the plain point-SOR update
//   du = (1-omega) du + omega (b1 - sigma) / (a11 + sum psi)
// in row-major order (the reference's default build has no OpenMP, so its row loop is serial).  The set-up stages
// (warp with wy = 0, derivatives) are the optical-flow ones (vr_setup_kernel<NOC, 1>).  Extra planes, FDF image_t
// layout, appended to the workspace of varref.hip.h:
//   du  uu  s (smoothness weight)  a11  b1  sh  sv
// Kernels per inner iteration: smooth -> data -> sor (+ clamped update uu = min/max(wx + du, 0), :299-314).
#pragma once
#include "varref.hip.h"

namespace fotg {

enum VrDePlane { DE_DU = 0, DE_UU, DE_S, DE_A11, DE_B1, DE_SH, DE_SV, DE_NPLANE };

__host__ __device__ inline float *de_plane(const VrArgs &a, int pair, int k)
{
  return a.base + (size_t)pair * a.pair_stride + (size_t)(P_NSINGLE + C_NCOLOR * a.noc + k) * a.pl;
}

// image_erase(du); uu = wx (refine_variational.cpp:277-280)
__global__ __launch_bounds__(256) void vr_de_init_kernel(VrArgs a)
{
  const WgId wg = xcd_local_wg();
  const int idx = wg.x * blockDim.x + threadIdx.x;
  if (idx >= a.w * a.h) return;
  const int pair = wg.y, o = (idx / a.w) * a.st + idx % a.w;
  de_plane(a, pair, DE_DU)[o] = 0.f;
  de_plane(a, pair, DE_UU)[o] = a.single(pair, P_WX)[o];
}

// compute_smoothness first half (opticalflow_aux.c:126-139) with vv = wy_dummy = 0 (refine_variational.cpp:288)
__global__ __launch_bounds__(256) void vr_de_smooth_kernel(VrArgs a, float quarter_alpha)
{
  const WgId wg = xcd_local_wg();
  const int idx = wg.x * blockDim.x + threadIdx.x;
  if (idx >= a.w * a.h) return;
  const int pair = wg.y, i = idx % a.w, j = idx / a.w, st = a.st, o = j * st + i;
  const float *uu = de_plane(a, pair, DE_UU);
  auto at = [&](int q) { return make_float2(uu[q], 0.f); };
  const int ol = i > 0 ? o - 1 : o, orr = i < a.w - 1 ? o + 1 : o, ot = j > 0 ? o - st : o, ob = j < a.h - 1 ? o + st : o;
  de_plane(a, pair, DE_S)[o] = smooth_w(at(ol), at(o), at(orr), at(ot), at(ob), j, a.h, quarter_alpha);
}

// compute_data_DE (opticalflow_aux.c:446-540) + sub_laplacian(b1, wx) (:172-199) for pixel (i, j): the scalar system (a11, b1)
// given the increment u = du and the four smoothness pair sums
template <int NOC>
__device__ __forceinline__ void de_data_pixel(const PixIn<NOC> &p, float u, float hr, float hl, float vb, float vt, int i, int j, int w, int h,
                                              float half_delta_over3, float half_gamma_over3, float &A11_out, float &B1_out)
{
  const float dnorm = 0.1f * 0.1f, epsc = 0.001f * 0.001f, epsg = 0.001f * 0.001f;
  const float m = p.m;
  float A11 = 0, B1 = 0;
  if constexpr (NOC == 1) {
    const float Ix = p.Ix[0], Iy = p.Iy[0], Iz = p.Iz[0], Ixx = p.Ixx[0], Ixy = p.Ixy[0], Iyy = p.Iyy[0], Ixz = p.Ixz[0], Iyz = p.Iyz[0];
    float tmp, tmp2, n1, n2;
    if (half_delta_over3) {
      tmp = Iz + Ix * u;
      n1 = Ix * Ix + Iy * Iy + dnorm;
      tmp = m * half_delta_over3 / sqrtf(3 * tmp * tmp / n1 + epsc);
      tmp /= n1;
      A11 += tmp * Ix * Ix;
      B1 -= tmp * Iz * Ix;
    }
    n1 = Ixx * Ixx + Ixy * Ixy + dnorm;
    n2 = Iyy * Iyy + Ixy * Ixy + dnorm;
    tmp = Ixz + Ixx * u;
    tmp2 = Iyz + Ixy * u;
    tmp = m * half_gamma_over3 / sqrtf(3 * tmp * tmp / n1 + 3 * tmp2 * tmp2 / n2 + epsg);
    tmp2 = tmp / n2; tmp /= n1;
    A11 += tmp * Ixx * Ixx + tmp2 * Ixy * Ixy;
    B1 -= tmp * Ixx * Ixz + tmp2 * Ixy * Iyz;
    A11 *= 3; B1 *= 3;                                     // :537-540
  } else {
    if (half_delta_over3) {
      float t[3], n[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { t[c] = p.Iz[c] + p.Ix[c] * u; n[c] = p.Ix[c] * p.Ix[c] + p.Iy[c] * p.Iy[c] + dnorm; }
      const float tmp = m * half_delta_over3 / sqrtf(t[0] * t[0] / n[0] + t[1] * t[1] / n[1] + t[2] * t[2] / n[2] + epsc);
      const float k[3] = {tmp / n[0], tmp / n[1], tmp / n[2]};
#pragma unroll
      for (int c = 0; c < 3; ++c) { A11 += k[c] * p.Ix[c] * p.Ix[c]; B1 -= k[c] * p.Iz[c] * p.Ix[c]; }
    }
    float n1[3], n2[3], t1[3], t2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      n1[c] = p.Ixx[c] * p.Ixx[c] + p.Ixy[c] * p.Ixy[c] + dnorm; n2[c] = p.Iyy[c] * p.Iyy[c] + p.Ixy[c] * p.Ixy[c] + dnorm;
      t1[c] = p.Ixz[c] + p.Ixx[c] * u;                           t2[c] = p.Iyz[c] + p.Ixy[c] * u;
    }
    const float tmp = m * half_gamma_over3 / sqrtf(t1[0] * t1[0] / n1[0] + t2[0] * t2[0] / n2[0] + t1[1] * t1[1] / n1[1] +
                                                   t2[1] * t2[1] / n2[1] + t1[2] * t1[2] / n1[2] + t2[2] * t2[2] / n2[2] + epsg);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float ka = tmp / n1[c], kb = tmp / n2[c];
      A11 += ka * p.Ixx[c] * p.Ixx[c] + kb * p.Ixy[c] * p.Ixy[c];
      B1 -= ka * p.Ixx[c] * p.Ixz[c] + kb * p.Ixy[c] * p.Iyz[c];
    }
  }
  // sub_laplacian(b1, wx): -left, +right, -top, +bottom
  if (i > 0)     B1 -= hl * (p.wxc - p.wxl);
  if (i < w - 1) B1 += hr * (p.wxr - p.wxc);
  if (j > 0)     B1 -= vt * (p.wxc - p.wxt);
  if (j < h - 1) B1 += vb * (p.wxb - p.wxc);
  A11_out = A11; B1_out = B1;
}

// compute_smoothness second half (:141-163) + de_data_pixel, one thread per pixel
template <int NOC>
__global__ __launch_bounds__(256) void vr_de_data_kernel(VrArgs a, float half_delta_over3, float half_gamma_over3)
{
  const WgId wg = xcd_local_wg();
  const int idx = wg.x * blockDim.x + threadIdx.x;
  if (idx >= a.w * a.h) return;
  const int pair = wg.y, w = a.w, h = a.h, i = idx % w, j = idx / w, st = a.st, o = j * st + i;
  const PixIn<NOC> p = data_load<NOC>(a, pair, i, j);
  const float *s = de_plane(a, pair, DE_S);
  const float s_o = s[o];
  const float hr = (i < w - 1) ? s_o + s[o + 1] : 0.0f;
  const float hl = (i > 0) ? s[o - 1] + s_o : 0.0f;
  const float vb = (j < h - 1) ? s_o + s[o + st] : 0.0f;
  const float vt = (j > 0) ? s[o - st] + s_o : 0.0f;
  float A11, B1;
  de_data_pixel<NOC>(p, de_plane(a, pair, DE_DU)[o], hr, hl, vb, vt, i, j, w, h, half_delta_over3, half_gamma_over3, A11, B1);
  de_plane(a, pair, DE_A11)[o] = A11;
  de_plane(a, pair, DE_B1)[o] = B1;
  de_plane(a, pair, DE_SH)[o] = hr;
  de_plane(a, pair, DE_SV)[o] = vb;
}

// The sweeps of sor_coupled_slow_but_readable_DE as an anti-diagonal wavefront, one workgroup per pair, thread r = image
// row r.  Pixel (i, j) of sweep n runs at step i + j + 2n: after its NEW top/left neighbours (same sweep, one step earlier)
// and with the OLD bottom/right ones (previous sweep, also one step earlier) -- the dependencies of the serial row-major
// loops, hence the same floats.  One barrier per step.  LDS: 2 = du and the four coefficient planes live in LDS (levels
// up to 8192 cells: the finest level of 1080p op-pt 2 is 120x68 = 8160), 1 = only du (coefficients are read from global
// memory one step ahead of their use), 0 = everything in global memory (levels beyond 128 KiB of du).
// Afterwards: uu = min/max(wx + du, 0) (refine_variational.cpp:299-314).
extern __shared__ float fotg_de_lds[];
// SPLIT: the SW sweeps of a step run on SW groups of waves (thread = (sweep, row); blockDim = SW * roundup64(h)) instead
// of one after the other in each thread -- a solver wave is bound by its own instruction stream, so this divides the
// step time by almost SW.  Needs SW * roundup64(h) <= 1024.
template <int SW, int LDS, bool SPLIT = false>
__global__ __launch_bounds__(1024) void vr_de_sor_kernel(VrArgs a, float omega, int camlr)
{
  constexpr bool LDS_DU = LDS >= 1;
  constexpr int NS = SPLIT ? 1 : SW;               // sweeps handled by one thread
  const int rows_pad = SPLIT ? (int)blockDim.x / SW : (int)blockDim.x;
  const int n0 = SPLIT ? (int)threadIdx.x / rows_pad : 0;
  const int pair = blockIdx.x, w = a.w, h = a.h, st = a.st, j = SPLIT ? (int)threadIdx.x % rows_pad : (int)threadIdx.x;
  const float *__restrict__ a11 = de_plane(a, pair, DE_A11), *__restrict__ b1 = de_plane(a, pair, DE_B1);
  const float *__restrict__ sh = de_plane(a, pair, DE_SH), *__restrict__ sv = de_plane(a, pair, DE_SV);
  float *dug = de_plane(a, pair, DE_DU);
  float *du = LDS_DU ? fotg_de_lds : dug;
  if (LDS_DU) {
    const int np = st * h;
    for (int k = threadIdx.x; k < np; k += blockDim.x) du[k] = dug[k];
    if (LDS == 2) {
      float *l = fotg_de_lds + np;
      for (int k = threadIdx.x; k < np; k += blockDim.x) { l[k] = a11[k]; l[np + k] = b1[k]; l[2 * np + k] = sh[k]; l[3 * np + k] = sv[k]; }
      a11 = l; b1 = l + np; sh = l + 2 * np; sv = l + 3 * np;
    }
    __syncthreads();
  }
  // Branch-free step: every lane always loads (indices clamped into the image), the terms of missing neighbours are
  // selected to 0 (x - 0 == x, x + 0 == x: the same floats as the reference's skipped statements) and only the store is
  // predicated -- so the loads of the SW sweeps of a step are all in flight together instead of one dependent
  // load -> divide -> store chain per sweep.  Loads first, stores last: within a step the sweeps touch disjoint
  // diagonals (d, d-2, d-4) and read only d+-1, so this is the sequential order.
  struct Coef { float a, b, h, v, vt; };
  const bool row = j < h;
  const int jc = row ? j : h - 1;
  const bool has_t = jc > 0, has_b = jc < h - 1;
  auto load = [&](int i) {
    const int o = jc * st + clampi(i, w);
    Coef c;
    c.a = a11[o]; c.b = b1[o]; c.h = sh[o]; c.v = sv[o]; c.vt = sv[has_t ? o - st : o];
    return c;
  };
  Coef cur[NS];
  float hl[NS];                                   // sh of the previous pixel of the row (psi towards the left neighbour)
#pragma unroll
  for (int n = 0; n < NS; ++n) { cur[n] = load(0 - j - 2 * (n0 + n)); hl[n] = 0.f; }
  const float om1 = 1.0f - omega;
  const int T = w + h - 1 + 2 * (SW - 1);
  for (int t = 0; t < T; ++t) {
    Coef nxt[NS];
    float own[NS], up[NS], lf[NS], dn[NS], rt[NS], res[NS];
    int oo[NS];
    bool valid[NS], has_l[NS], has_r[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const int i = t - j - 2 * (n0 + n), ic = clampi(i, w), o = jc * st + ic;
      valid[n] = row && i >= 0 && i < w;
      has_l[n] = ic > 0; has_r[n] = ic < w - 1;
      oo[n] = o;
      own[n] = du[o];
      up[n] = du[has_t ? o - st : o];
      lf[n] = du[has_l[n] ? o - 1 : o];
      dn[n] = du[has_b ? o + st : o];
      rt[n] = du[has_r[n] ? o + 1 : o];
      nxt[n] = load(i + 1);
    }
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      const Coef c = cur[n];
      float sigma_u = 0.0f, sum_dpsis = 0.0f;
      sigma_u = sigma_u - (has_t ? c.vt * up[n] : 0.0f);     sum_dpsis = sum_dpsis + (has_t ? c.vt : 0.0f);
      sigma_u = sigma_u - (has_l[n] ? hl[n] * lf[n] : 0.0f); sum_dpsis = sum_dpsis + (has_l[n] ? hl[n] : 0.0f);
      sigma_u = sigma_u - (has_b ? c.v * dn[n] : 0.0f);      sum_dpsis = sum_dpsis + (has_b ? c.v : 0.0f);
      sigma_u = sigma_u - (has_r[n] ? c.h * rt[n] : 0.0f);   sum_dpsis = sum_dpsis + (has_r[n] ? c.h : 0.0f);
      const float A11 = c.a + sum_dpsis;
      const float B1 = c.b - sigma_u;
      res[n] = om1 * own[n] + omega * (B1 / A11);
    }
#pragma unroll
    for (int n = 0; n < NS; ++n) {
      if (valid[n]) { du[oo[n]] = res[n]; hl[n] = cur[n].h; }
      cur[n] = nxt[n];
    }
    __syncthreads();
  }
  const float *wx = a.single(pair, P_WX);
  float *uu = de_plane(a, pair, DE_UU);
  for (int k = threadIdx.x; k < w * h; k += blockDim.x) {
    const int o = (k / w) * st + k % w;
    const float d = du[o];
    if (LDS_DU) dug[o] = d;
    const float s = wx[o] + d;
    uu[o] = camlr == 0 ? (s < 0.0f ? s : 0.0f) : (s > 0.0f ? s : 0.0f);
  }
}

// Levels of more than 1024 rows (a full-resolution level of a large frame: no operating point gets there): ONE sweep of the same
// wavefront per launch, one workgroup per pair, every thread walks the rows tid, tid + 1024, ... of a step.  Nothing is carried
// between steps (the coefficient of the left neighbour is read back: sh of the previous pixel of the row), everything in global
// memory; within a step the cells of diagonal t are written and those of t - 1 / t + 1 read, so the order inside a step is free.
// k launches = k sweeps, bit for bit (the sweeps are sequential passes over du).  clamp: the update uu = min/max(wx + du, 0)
// behind the LAST sweep (refine_variational.cpp:299-314).
__global__ __launch_bounds__(1024) void vr_de_sor_tall_kernel(VrArgs a, float omega, int camlr, int clamp)
{
  const int pair = blockIdx.x, w = a.w, h = a.h, st = a.st;
  const float *__restrict__ a11 = de_plane(a, pair, DE_A11), *__restrict__ b1 = de_plane(a, pair, DE_B1);
  const float *__restrict__ sh = de_plane(a, pair, DE_SH), *__restrict__ sv = de_plane(a, pair, DE_SV);
  float *du = de_plane(a, pair, DE_DU);
  const float om1 = 1.0f - omega;
  const int T = w + h - 1;
  for (int t = 0; t < T; ++t) {
    for (int j = threadIdx.x; j < h; j += blockDim.x) {
      const int i = t - j;
      if (i < 0 || i >= w) continue;
      const int o = j * st + i;
      const bool has_t = j > 0, has_b = j < h - 1, has_l = i > 0, has_r = i < w - 1;
      const float ca = a11[o], cb = b1[o], ch = sh[o], cv = sv[o], cvt = sv[has_t ? o - st : o], hl = has_l ? sh[o - 1] : 0.f;
      const float own = du[o], up = du[has_t ? o - st : o], lf = du[has_l ? o - 1 : o], dn = du[has_b ? o + st : o], rt = du[has_r ? o + 1 : o];
      float sigma_u = 0.0f, sum_dpsis = 0.0f;
      sigma_u = sigma_u - (has_t ? cvt * up : 0.0f);  sum_dpsis = sum_dpsis + (has_t ? cvt : 0.0f);
      sigma_u = sigma_u - (has_l ? hl * lf : 0.0f);   sum_dpsis = sum_dpsis + (has_l ? hl : 0.0f);
      sigma_u = sigma_u - (has_b ? cv * dn : 0.0f);   sum_dpsis = sum_dpsis + (has_b ? cv : 0.0f);
      sigma_u = sigma_u - (has_r ? ch * rt : 0.0f);   sum_dpsis = sum_dpsis + (has_r ? ch : 0.0f);
      const float A11 = ca + sum_dpsis;
      const float B1 = cb - sigma_u;
      du[o] = om1 * own + omega * (B1 / A11);
    }
    __syncthreads();
  }
  if (!clamp) return;
  const float *wx = a.single(pair, P_WX);
  float *uu = de_plane(a, pair, DE_UU);
  for (int k = threadIdx.x; k < w * h; k += blockDim.x) {
    const int o = (k / w) * st + k % w;
    const float s = wx[o] + du[o];
    uu[o] = camlr == 0 ? (s < 0.0f ? s : 0.0f) : (s > 0.0f ? s : 0.0f);
  }
}

// The whole level after the set-up stage in ONE launch, one workgroup (1024 threads) per pair, for levels whose five planes
// du, uu, s, a11, b1 fit in LDS (<= 8192 cells): per inner iteration the smoothness weights, the data term (per-pixel
// phases, all threads), the three sweeps (the wave groups of vr_de_sor_kernel's SPLIT form; the other waves only keep the
// barriers) and the clamped update -- nothing but the per-pixel image terms is read from global memory in between, and
// nothing is written until the final uu.  The solver forms psi from the s plane on the fly (sh = s + s_right etc., the
// same additions compute_smoothness makes), so sh / sv need no planes.  taps: also store du, uu, s, a11, b1, sh, sv of
// the last inner iteration to the workspace planes for fotg_varref_plane.
template <int NOC>
__global__ __launch_bounds__(1024) void vr_de_inner_kernel(VrArgs a, int inner, float quarter_alpha, float half_delta_over3,
                                                           float half_gamma_over3, float omega, int camlr, float *__restrict__ flow,
                                                           long flow_stride, int rows_pad, int taps)
{
  constexpr int SW = 3;
  const int pair = blockIdx.x, w = a.w, h = a.h, st = a.st, np = st * h, tid = threadIdx.x;
  float *du = fotg_de_lds, *uu = du + np, *S = uu + np, *A = S + np, *B = A + np;
  const float *__restrict__ wx = a.single(pair, P_WX);
  for (int k = tid; k < np; k += 1024) { du[k] = 0.f; uu[k] = (k % st) < w ? wx[k] : 0.f; }     // :277-280
  const bool solver = tid < SW * rows_pad;
  const int n0 = tid / rows_pad, j = tid % rows_pad;
  const bool row = solver && j < h;
  const int jc = j < h ? j : h - 1;
  const bool has_t = jc > 0, has_b = jc < h - 1;
  const float om1 = 1.0f - omega;
  const int T = w + h - 1 + 2 * (SW - 1);
  __syncthreads();
  for (int it = 0; it < inner; ++it) {
    // compute_smoothness first half (:126-139) on (uu, 0)
    for (int k = tid; k < w * h; k += 1024) {
      const int i = k % w, jj = k / w, o = jj * st + i;
      auto at = [&](int q) { return make_float2(uu[q], 0.f); };
      const int ol = i > 0 ? o - 1 : o, orr = i < w - 1 ? o + 1 : o, ot = jj > 0 ? o - st : o, ob = jj < h - 1 ? o + st : o;
      S[o] = smooth_w(at(ol), at(o), at(orr), at(ot), at(ob), jj, h, quarter_alpha);
    }
    __syncthreads();
    // second half (:141-163) + compute_data_DE + sub_laplacian
    for (int k = tid; k < w * h; k += 1024) {
      const int i = k % w, jj = k / w, o = jj * st + i;
      const PixIn<NOC> p = data_load<NOC>(a, pair, i, jj);
      const float s_o = S[o];
      const float hr = (i < w - 1) ? s_o + S[o + 1] : 0.0f;
      const float hl = (i > 0) ? S[o - 1] + s_o : 0.0f;
      const float vb = (jj < h - 1) ? s_o + S[o + st] : 0.0f;
      const float vt = (jj > 0) ? S[o - st] + s_o : 0.0f;
      float A11, B1;
      de_data_pixel<NOC>(p, du[o], hr, hl, vb, vt, i, jj, w, h, half_delta_over3, half_gamma_over3, A11, B1);
      A[o] = A11; B[o] = B1;
    }
    __syncthreads();
    // sor_coupled_slow_but_readable_DE: sweep n0 of row j, one diagonal per step (see vr_de_sor_kernel)
    for (int t = 0; t < T; ++t) {
      if (solver) {
        const int i = t - j - 2 * n0, ic = clampi(i, w), o = jc * st + ic;
        const bool valid = row && i >= 0 && i < w, has_l = ic > 0, has_r = ic < w - 1;
        const float own = du[o], up = du[has_t ? o - st : o], lf = du[has_l ? o - 1 : o], dn = du[has_b ? o + st : o], rt = du[has_r ? o + 1 : o];
        const float s_o = S[o], s_t = S[has_t ? o - st : o], s_l = S[has_l ? o - 1 : o], s_b = S[has_b ? o + st : o], s_r = S[has_r ? o + 1 : o];
        const float ca = A[o], cb = B[o];
        const float pvt = s_t + s_o, phl = s_l + s_o, pv = s_o + s_b, ph = s_o + s_r;      // vert[o-st], horiz[o-1], vert[o], horiz[o]
        float sigma_u = 0.0f, sum_dpsis = 0.0f;
        sigma_u = sigma_u - (has_t ? pvt * up : 0.0f); sum_dpsis = sum_dpsis + (has_t ? pvt : 0.0f);
        sigma_u = sigma_u - (has_l ? phl * lf : 0.0f); sum_dpsis = sum_dpsis + (has_l ? phl : 0.0f);
        sigma_u = sigma_u - (has_b ? pv * dn : 0.0f);  sum_dpsis = sum_dpsis + (has_b ? pv : 0.0f);
        sigma_u = sigma_u - (has_r ? ph * rt : 0.0f);  sum_dpsis = sum_dpsis + (has_r ? ph : 0.0f);
        const float A11 = ca + sum_dpsis;
        const float B1 = cb - sigma_u;
        const float res = om1 * own + omega * (B1 / A11);
        if (valid) du[o] = res;
      }
      __syncthreads();
    }
    // uu = min/max(wx + du, 0) (refine_variational.cpp:299-314)
    for (int k = tid; k < w * h; k += 1024) {
      const int o = (k / w) * st + k % w;
      const float sum = wx[o] + du[o];
      uu[o] = camlr == 0 ? (sum < 0.0f ? sum : 0.0f) : (sum > 0.0f ? sum : 0.0f);
    }
    __syncthreads();
  }
  for (int k = tid; k < w * h; k += 1024) {
    const int i = k % w, jj = k / w, o = jj * st + i;
    flow[(size_t)pair * flow_stride + k] = uu[o];                                   // :316-317
    if (taps) {
      de_plane(a, pair, DE_DU)[o] = du[o]; de_plane(a, pair, DE_UU)[o] = uu[o]; de_plane(a, pair, DE_S)[o] = S[o];
      de_plane(a, pair, DE_A11)[o] = A[o]; de_plane(a, pair, DE_B1)[o] = B[o];
      de_plane(a, pair, DE_SH)[o] = (i < w - 1) ? S[o] + S[o + 1] : 0.0f;
      de_plane(a, pair, DE_SV)[o] = (jj < h - 1) ? S[o] + S[o + st] : 0.0f;
    }
  }
}

__global__ __launch_bounds__(256) void vr_de_finish_kernel(VrArgs a, float *__restrict__ flow, long flow_stride)
{
  const WgId wg = xcd_local_wg();
  const int idx = wg.x * blockDim.x + threadIdx.x;
  if (idx >= a.w * a.h) return;
  const int pair = wg.y, o = (idx / a.w) * a.st + idx % a.w;
  flow[(size_t)pair * flow_stride + idx] = de_plane(a, pair, DE_UU)[o];          // :316-317, one channel
}

}  // namespace fotg
