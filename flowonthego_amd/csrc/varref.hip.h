// varref.hip.h -- variational refinement of one pyramid level
// (kroeger/refine_variational.cpp:25-241 driving FDF1.0.1/{opticalflow_aux,solver,image}.c).
//
// Planes use the FDF image_t layout (stride = ceil4(w), FDF1.0.1/image.c:15-31), one set per pair:
//   wx wy mask                                          (3 single planes)
//   avg Iz Ix Iy Ixx Ixy Iyy Ixz Iyz                    (9 x NOC planes, channel-planar like color_image_t)
// plus the solver's skewed arrays C (system) and D (du,dv), see VrArgs.
// Kernels (per inner iteration: data -> sor):
//   vr_prep     de-interleave flow, image_warp + mask (opticalflow_aux.c:18-60), 0.5*(I0+Iw), Iw-I0 (:80-83)
//   vr_deriv1/2 the seven 5-tap derivative images (opticalflow_aux.c:85-92, image.c:401-434,466-502)
//   vr_data     smoothness s = 1/4 alpha / sqrt(|grad uu|^2+|grad vv|^2+eps) with uu = wx+du on the fly
//               (opticalflow_aux.c:123-139) and its pair sums (:141-163), data term (:310-438), sub_laplacian
//               (:172-199, gather form), the 2x2 block inverse of sor_coupled's first sweep (solver.c:115-120);
//               writes the skewed system C
//   vr_sor      the sweeps of sor_coupled (solver.c:77-421) as a register-pipelined single-wave wavefront
//   vr_finish   flow = (wx+du, wy+dv) (refine_variational.cpp:208-221)
#pragma once
#include <type_traits>
#include "common.h"

namespace fotg {

enum VrPlane { P_WX = 0, P_WY, P_MASK, P_NSINGLE };
enum VrCPlane { C_AVG = 0, C_IZ, C_IX, C_IY, C_IXX, C_IXY, C_IYY, C_IXZ, C_IYZ, C_NCOLOR };

// Skewed (anti-diagonal major) arrays used by the solver: cell (i,j) of the image lives at [i+j][j].
//   C[s][r] = {a11, a12 (2x2 block inverse), b1, b2 | a22, psi_right, psi_bottom, psi_top}   8 floats
//             (pairs the solver multiplies as packed f32 sit in even-aligned register pairs)
//   D[s][r] = {du, dv}                                                                     2 floats
// Row s holds exactly the pixels the lexicographic sweep may process together (step s of the wavefront),
// contiguous in the row index r, so one wave reads/writes a step with fully coalesced 16-B / 8-B accesses.
// Cells outside the image stay zero for the life of the context (C) / of a refinement (D); a zero cell is a
// fixed point of the update, so the solver needs no bounds predicates.
struct VrArgs {
  float *base;           // workspace of pair 0
  long pair_stride;      // floats between pairs
  long pl;               // floats per plane (st*h)
  int w, h, st, noc;
  float4 *C;             // [pair][SC][RP][2] float4
  float2 *D;             // [pair][SC+1][RPD] float2
  long c_pair_stride;    // in float4
  long d_pair_stride;    // in float2
  int S, SC, RP, RPD, K, nlanes, nsweeps, nbands, band_rows, band_mode, taps;
  int *zsync;            // sync words of the tile pipeline the NEXT launch (the level's sor_coupled call) expects zeroed: the data-term
  int zsync_n;           // launch clears them on its way (a memset launch per call otherwise: 13 per 4K pair)
  int nwg;               // vr_setup_kernel: > 0 = XCD-banded placement of the tiles (xcd_banded_x; launches of 1..7 pairs that span the chip)
  int redblack;          // FOTG_SOR_REDBLACK: the fused per-level kernel relaxes with red-black half-sweeps instead of the wavefront solver
  int point;             // FOTG_SOR_POINT: cells hold (A11 + sum psi, A12, b1, b2 | A22 + sum psi, psi_r, psi_b, psi_t), no block inverse
  __host__ __device__ int pix(int i, int j) const { return j * st + i; }
  __host__ __device__ float4 *Cp(int pair) const { return C + (size_t)pair * c_pair_stride; }
  __host__ __device__ float2 *Dp(int pair) const { return D + (size_t)pair * d_pair_stride; }
  __host__ __device__ size_t cidx(int i, int j) const { return ((size_t)(i + j) * RP + j) * 2; }
  __host__ __device__ size_t didx(int i, int j) const { return (size_t)(i + j) * RPD + j; }
  __host__ __device__ float *single(int pair, int p) const { return base + (size_t)pair * pair_stride + (size_t)p * pl; }
  __host__ __device__ float *color(int pair, int p, int c) const {
    return base + (size_t)pair * pair_stride + (size_t)(P_NSINGLE + p * noc + c) * pl;
  }
};

// convolution_new(2,{0,-8/12,1/12},0) -> {1/12,-8/12,-0,8/12,-1/12} (image.c:326-349, refine_variational.cpp:45-46)
#define FOTG_D5 const float c0 = 1.0f / 12.0f, c1 = -8.0f / 12.0f, c2 = -0.0f, c3 = -(-8.0f / 12.0f), c4 = -(1.0f / 12.0f)

__device__ __forceinline__ float conv_h5(const float *__restrict__ row, int i, int w)
{
  FOTG_D5;
  return c0 * row[clampi(i - 2, w)] + c1 * row[clampi(i - 1, w)] + c2 * row[i] + c3 * row[clampi(i + 1, w)] + c4 * row[clampi(i + 2, w)];
}
__device__ __forceinline__ float conv_v5(const float *__restrict__ col, int j, int h, int st)
{
  FOTG_D5;
#define S(r) col[(size_t)(r) * st]
  if (j == 0) return (c0 + c1 + c2) * S(0) + c3 * S(1) + c4 * S(2);
  if (j == 1) return (c0 + c1) * S(0) + c2 * S(1) + c3 * S(2) + c4 * S(3);
  if (j == h - 2) return c0 * S(j - 2) + c1 * S(j - 1) + c2 * S(j) + (c3 + c4) * S(j + 1);
  if (j == h - 1) return c0 * S(j - 2) + c1 * S(j - 1) + (c2 + c3 + c4) * S(j);
  return c0 * S(j - 2) + c1 * S(j - 1) + c2 * S(j) + c3 * S(j + 1) + c4 * S(j + 2);
#undef S
}

// per-pixel bodies of the three set-up stages (shared by the wide kernels and the fused per-pair kernel)
template <int NOC>
struct PrepVal { float wx, wy, mask, avg[NOC], iz[NOC]; };

// NCH = 1: stereo depth mode, the flow has one channel and wy = 0 (wy_dummy, refine_variational.cpp:258,273)
template <int NOC, int NCH = 2>
__device__ __forceinline__ PrepVal<NOC> prep_values(const VrArgs &a, int pair, int i, int j, const float *__restrict__ I0, const float *__restrict__ I1,
                                                    long img_stride, int tw, int pad, const float *__restrict__ flow, long flow_stride)
{
  PrepVal<NOC> v;
  const float *f = flow + (size_t)pair * flow_stride + NCH * (size_t)(j * a.w + i);
  const float wx = f[0], wy = NCH == 2 ? f[NCH - 1] : 0.f;
  v.wx = wx; v.wy = wy;
  // image_warp (opticalflow_aux.c:18-60)
  const float xx = i + wx, yy = j + wy;
  const int x = (int)floorf(xx), y = (int)floorf(yy);
  const float dx = xx - x, dy = yy - y;
  v.mask = (xx >= 0 && xx <= a.w - 1 && yy >= 0 && yy <= a.h - 1) ? 1.f : 0.f;
  // (a flow that has diverged -- a relaxation weight outside (0, 2), say -- saturates the conversion: x + 1 must not overflow)
  const int xs = x < -2 ? -2 : (x > a.w ? a.w : x), ys = y < -2 ? -2 : (y > a.h ? a.h : y);
  const int x1 = clampi(xs, a.w), x2 = clampi(xs + 1, a.w), y1 = clampi(ys, a.h), y2 = clampi(ys + 1, a.h);
  const float *s1 = I1 + (size_t)pair * img_stride, *s0 = I0 + (size_t)pair * img_stride;
#pragma unroll
  for (int c = 0; c < NOC; ++c) {
#define SRC(yy_, xx_) s1[((size_t)((yy_) + pad) * tw + ((xx_) + pad)) * NOC + c]
    const float wv = SRC(y1, x1) * (1.0f - dx) * (1.0f - dy) + SRC(y1, x2) * dx * (1.0f - dy) +
                     SRC(y2, x1) * (1.0f - dx) * dy + SRC(y2, x2) * dx * dy;
#undef SRC
    const float i1 = s0[((size_t)(j + pad) * tw + (i + pad)) * NOC + c];
    v.avg[c] = 0.5f * (wv + i1);                          // get_derivatives :81
    v.iz[c] = wv - i1;                                    // :82
  }
  return v;
}

template <int NOC>
__device__ __forceinline__ void prep_store(const VrArgs &a, int pair, int i, int j, const PrepVal<NOC> &v)
{
  const int o = a.pix(i, j);
  a.single(pair, P_WX)[o] = v.wx;
  a.single(pair, P_WY)[o] = v.wy;
  a.single(pair, P_MASK)[o] = v.mask;
#pragma unroll
  for (int c = 0; c < NOC; ++c) { a.color(pair, C_AVG, c)[o] = v.avg[c]; a.color(pair, C_IZ, c)[o] = v.iz[c]; }
}

template <int NOC>
__device__ __forceinline__ void prep_pixel(const VrArgs &a, int pair, int i, int j, const float *__restrict__ I0, const float *__restrict__ I1,
                                           long img_stride, int tw, int pad, const float *__restrict__ flow, long flow_stride)
{
  prep_store<NOC>(a, pair, i, j, prep_values<NOC>(a, pair, i, j, I0, I1, img_stride, tw, pad, flow, flow_stride));
}

template <int NOC>
__device__ __forceinline__ void deriv1_pixel(const VrArgs &a, int pair, int i, int j)
{
  const int o = j * a.st + i;
#pragma unroll
  for (int c = 0; c < NOC; ++c) {
    const float *avg = a.color(pair, C_AVG, c), *iz = a.color(pair, C_IZ, c);
    a.color(pair, C_IX, c)[o] = conv_h5(avg + (size_t)j * a.st, i, a.w);
    a.color(pair, C_IY, c)[o] = conv_v5(avg + i, j, a.h, a.st);
    a.color(pair, C_IXZ, c)[o] = conv_h5(iz + (size_t)j * a.st, i, a.w);
    a.color(pair, C_IYZ, c)[o] = conv_v5(iz + i, j, a.h, a.st);
  }
}

template <int NOC>
__device__ __forceinline__ void deriv2_pixel(const VrArgs &a, int pair, int i, int j)
{
  const int o = j * a.st + i;
#pragma unroll
  for (int c = 0; c < NOC; ++c) {
    const float *ix = a.color(pair, C_IX, c), *iy = a.color(pair, C_IY, c);
    a.color(pair, C_IXX, c)[o] = conv_h5(ix + (size_t)j * a.st, i, a.w);
    a.color(pair, C_IXY, c)[o] = conv_v5(ix + i, j, a.h, a.st);
    a.color(pair, C_IYY, c)[o] = conv_v5(iy + i, j, a.h, a.st);
  }
}

// smoothness weight from the 3x3 cross of (uu,vv): compute_smoothness first half (opticalflow_aux.c:126-139);
// 3-tap {-0.5,-0,0.5} with the border rows of convolve_vert_fast_3 / replicate columns of convolve_horiz_fast_3
// (image.c:376-399,436-464).  l,c,r = left/centre/right, t,b = top/bottom (t or b unused on the border rows).
// The three set-up stages of a level in ONE launch for levels too large for the fused per-pair kernel: one workgroup per 32x8
// tile recomputes what it needs of the neighbouring tiles instead of meeting them in global memory -- warp / average /
// difference on the tile + 4 pixels (the second derivatives reach 2 + 2 pixels), first derivatives on the tile + 2.  Halo
// entries outside the image hold the values of the clamped coordinate, exactly what the reference's replicate indexing
// reads, and the 5-tap helpers index the LDS tiles through pointers biased to global coordinates.
// (defined behind the data term: the first inner iteration's system, computed by the set-up launch itself)
template <int NOC, int NCH, bool FM>
__device__ __forceinline__ void first_data_term(const VrArgs &a, int pair, int tx0, int ty0, const float *__restrict__ flow, long flow_stride,
                                                float quarter_alpha, float half_delta_over3, float half_gamma_over3);

// first_data != 0 (flow mode): the launch also builds the system of the FIRST inner iteration (du = dv = 0), i.e. what
// vr_data_kernel would do next -- one launch and one read of the planes less per level.
template <int NOC, int NCH = 2, bool FM = false>
__global__ __launch_bounds__(256) void vr_setup_kernel(VrArgs a, const float *__restrict__ I0, const float *__restrict__ I1,
                                                       long img_stride, int tw, int pad,
                                                       const float *__restrict__ flow, long flow_stride, int zero_d = 0,
                                                       int first_data = 0, float quarter_alpha = 0.f, float half_delta_over3 = 0.f, float half_gamma_over3 = 0.f)
{
  constexpr int TW_ = 32, TH_ = 8, XW = TW_ + 8, XH = TH_ + 8, YW = TW_ + 4, YH = TH_ + 4;
  __shared__ float Xa[NOC][XH * XW], Xz[NOC][XH * XW], Yx[NOC][YH * YW], Yy[NOC][YH * YW];
  if (first_data && a.zsync_n > 0)
    for (long k = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; k < a.zsync_n; k += (long)gridDim.x * gridDim.y * 256) a.zsync[k] = 0;
  WgId wg = xcd_local_wg();
  int ntiles = (int)gridDim.x;
  // a single pair (4K): neighbouring tiles share their halos (the warp is evaluated on tile + 4, the derivatives on tile + 2); dealt
  // round robin every XCD's L2 fetched the halo rows of all its neighbours' tiles -- bands of tile rows per XCD instead
  if (a.nwg > 0) { wg.x = xcd_banded_x(a.nwg); ntiles = a.nwg; if (wg.x < 0) return; }
  const int pair = wg.y, w = a.w, h = a.h;
  const int tiles_x = (w + TW_ - 1) / TW_;
  if (zero_d == 1) {
    // image_erase(du), image_erase(dv) (refine_variational.cpp:185-186): the pair's tiles share the zeroing of its skewed D
    float2 *D = a.Dp(pair);
    for (long k = (long)wg.x * 256 + threadIdx.x; k < a.d_pair_stride; k += (long)ntiles * 256) D[k] = make_float2(0.f, 0.f);
  }
  const int tx0 = (wg.x % tiles_x) * TW_, ty0 = (wg.x / tiles_x) * TH_;
  // stage A: warp + mask + average / difference at the clamped coordinate of every tile+4 position
  for (int e = threadIdx.x; e < XW * XH; e += 256) {
    const int cx = tx0 - 4 + e % XW, cy = ty0 - 4 + e / XW;
    const int gx = clampi(cx, w), gy = clampi(cy, h);
    const PrepVal<NOC> v = prep_values<NOC, NCH>(a, pair, gx, gy, I0, I1, img_stride, tw, pad, flow, flow_stride);
#pragma unroll
    for (int c = 0; c < NOC; ++c) { Xa[c][e] = v.avg[c]; Xz[c][e] = v.iz[c]; }
    if (cx >= tx0 && cx < tx0 + TW_ && cy >= ty0 && cy < ty0 + TH_ && cx < w && cy < h) prep_store<NOC>(a, pair, cx, cy, v);
  }
  __syncthreads();
  // stage B: first derivatives at the clamped coordinate of every tile+2 position (get_derivatives :84-93)
  for (int e = threadIdx.x; e < YW * YH; e += 256) {
    const int cx = tx0 - 2 + e % YW, cy = ty0 - 2 + e / YW;
    const int gx = clampi(cx, w), gy = clampi(cy, h);
    const bool own = cx >= tx0 && cx < tx0 + TW_ && cy >= ty0 && cy < ty0 + TH_ && cx < w && cy < h;
#pragma unroll
    for (int c = 0; c < NOC; ++c) {
      // row / column pointers biased so that GLOBAL indices address the tile: X position of global (x, y) = (y - ty0 + 4) * XW + x - tx0 + 4
      const float *arow = Xa[c] + (gy - ty0 + 4) * XW - tx0 + 4, *acol = Xa[c] + (4 - ty0) * XW + gx - tx0 + 4;
      const float *zrow = Xz[c] + (gy - ty0 + 4) * XW - tx0 + 4, *zcol = Xz[c] + (4 - ty0) * XW + gx - tx0 + 4;
      const float ix = conv_h5(arow, gx, w), iy = conv_v5(acol, gy, h, XW);
      const float ixz = conv_h5(zrow, gx, w), iyz = conv_v5(zcol, gy, h, XW);
      Yx[c][e] = ix; Yy[c][e] = iy;
      if (own) {
        const int o = a.pix(cx, cy);
        a.color(pair, C_IX, c)[o] = ix; a.color(pair, C_IY, c)[o] = iy; a.color(pair, C_IXZ, c)[o] = ixz; a.color(pair, C_IYZ, c)[o] = iyz;
      }
    }
  }
  __syncthreads();
  // stage C: second derivatives of the tile's own pixels (:95-99)
  const int i = tx0 + (threadIdx.x % TW_), j = ty0 + (threadIdx.x / TW_);
  if (i < w && j < h) {
    const int o = a.pix(i, j);
#pragma unroll
    for (int c = 0; c < NOC; ++c) {
      const float *xrow = Yx[c] + (j - ty0 + 2) * YW - tx0 + 2, *xcol = Yx[c] + (2 - ty0) * YW + i - tx0 + 2, *ycol = Yy[c] + (2 - ty0) * YW + i - tx0 + 2;
      a.color(pair, C_IXX, c)[o] = conv_h5(xrow, i, w);
      a.color(pair, C_IXY, c)[o] = conv_v5(xcol, j, h, YW);
      a.color(pair, C_IYY, c)[o] = conv_v5(ycol, j, h, YW);
    }
  }
  if constexpr (NCH == 2) {
    if (first_data) {
      __syncthreads();                                           // the planes of the tile's pixels are in memory (stores of this workgroup)
      first_data_term<NOC, NCH, FM>(a, pair, tx0, ty0, flow, flow_stride, quarter_alpha, half_delta_over3, half_gamma_over3);
    }
  }
}

// Everything compute_data / sub_laplacian read from global memory for one pixel, gathered up front so callers can
// put the loads of several pixels in flight before computing (the fused kernel batches 4 pixels per thread).
template <int NOC>
struct PixIn {
  float Ix[NOC], Iy[NOC], Iz[NOC], Ixx[NOC], Ixy[NOC], Iyy[NOC], Ixz[NOC], Iyz[NOC];
  float m, wxc, wxl, wxr, wxt, wxb, wyc, wyl, wyr, wyt, wyb;
  // the differences sub_laplacian forms (opticalflow_aux.c:172-199)
  __device__ __forceinline__ float dxl() const { return wxc - wxl; }
  __device__ __forceinline__ float dxr() const { return wxr - wxc; }
  __device__ __forceinline__ float dxt() const { return wxc - wxt; }
  __device__ __forceinline__ float dxb() const { return wxb - wxc; }
  __device__ __forceinline__ float dyl() const { return wyc - wyl; }
  __device__ __forceinline__ float dyr() const { return wyr - wyc; }
  __device__ __forceinline__ float dyt() const { return wyc - wyt; }
  __device__ __forceinline__ float dyb() const { return wyb - wyc; }
};
template <int NOC>
__device__ __forceinline__ PixIn<NOC> data_load(const VrArgs &a, int pair, int i, int j)
{
  const int st = a.st, w = a.w, h = a.h, o = j * st + i;
  const float *wx = a.single(pair, P_WX), *wy = a.single(pair, P_WY);
  PixIn<NOC> p;
#pragma unroll
  for (int c = 0; c < NOC; ++c) {
    p.Ix[c] = a.color(pair, C_IX, c)[o]; p.Iy[c] = a.color(pair, C_IY, c)[o]; p.Iz[c] = a.color(pair, C_IZ, c)[o];
    p.Ixx[c] = a.color(pair, C_IXX, c)[o]; p.Ixy[c] = a.color(pair, C_IXY, c)[o]; p.Iyy[c] = a.color(pair, C_IYY, c)[o];
    p.Ixz[c] = a.color(pair, C_IXZ, c)[o]; p.Iyz[c] = a.color(pair, C_IYZ, c)[o];
  }
  p.m = a.single(pair, P_MASK)[o];
  const int ol = i > 0 ? o - 1 : o, orr = i < w - 1 ? o + 1 : o, ot = j > 0 ? o - st : o, ob = j < h - 1 ? o + st : o;
  p.wxc = wx[o]; p.wxl = wx[ol]; p.wxr = wx[orr]; p.wxt = wx[ot]; p.wxb = wx[ob];
  p.wyc = wy[o]; p.wyl = wy[ol]; p.wyr = wy[orr]; p.wyt = wy[ot]; p.wyb = wy[ob];
  return p;
}

// smooth_w() and data_term_cell() in the two arithmetics of the library (varref_dataterm.inc.h): the reference's operation by
// operation (parity mode) and, for fotg_params::fast_math, with v_rcp / v_rsq instead of IEEE divisions and square roots (tolerance mode)
#define FOTG_DT_FAST 0
#define FOTG_DT_NAME(x) x##_exact
#include "varref_dataterm.inc.h"
#undef FOTG_DT_FAST
#undef FOTG_DT_NAME
#define FOTG_DT_FAST 1
#define FOTG_DT_NAME(x) x##_fast
#include "varref_dataterm.inc.h"
#undef FOTG_DT_FAST
#undef FOTG_DT_NAME
template <bool FM = false>
__device__ __forceinline__ float smooth_w(float2 l, float2 c, float2 r, float2 t, float2 b, int j, int h, float quarter_alpha)
{
  if constexpr (FM) return smooth_w_fast(l, c, r, t, b, j, h, quarter_alpha);
  else return smooth_w_exact(l, c, r, t, b, j, h, quarter_alpha);
}
template <int NOC, bool FM = false>
__device__ __forceinline__ void data_term_cell(const VrArgs &a, int i, int j, const PixIn<NOC> &p, float hr, float hl, float vb, float vt,
                                               float u, float v, float half_delta_over3, float half_gamma_over3, float4 &c0, float4 &c1)
{
  if constexpr (FM) data_term_cell_fast<NOC>(a, i, j, p, hr, hl, vb, vt, u, v, half_delta_over3, half_gamma_over3, c0, c1);
  else data_term_cell_exact<NOC>(a, i, j, p, hr, hl, vb, vt, u, v, half_delta_over3, half_gamma_over3, c0, c1);
}

template <int NOC, bool FM = false>
__device__ __forceinline__ void data_term_compute(const VrArgs &a, int pair, int i, int j, const PixIn<NOC> &p, float hr, float hl,
                                                  float vb, float vt, float u, float v, float half_delta_over3, float half_gamma_over3)
{
  float4 c0, c1;
  data_term_cell<NOC, FM>(a, i, j, p, hr, hl, vb, vt, u, v, half_delta_over3, half_gamma_over3, c0, c1);
  float4 *C = a.Cp(pair) + a.cidx(i, j);
  C[0] = c0;
  C[1] = c1;
}

// One workgroup = one 32x8 pixel tile.  (uu,vv) of the tile + 2-pixel halo and the smoothness weight s of the tile +
// 1-pixel halo are staged in LDS, so each s is computed once (not once per neighbour) and the skewed D is read ~1.7x
// per pixel instead of 13x.
#define FOTG_TW 32
#define FOTG_TH 8
template <int NOC, bool FM = false>
__global__ __launch_bounds__(256) void vr_data_kernel(VrArgs a, float quarter_alpha, float half_delta_over3, float half_gamma_over3)
{
  constexpr int UW = FOTG_TW + 4, UH = FOTG_TH + 4, SW = FOTG_TW + 2, SH = FOTG_TH + 2;
  __shared__ float2 uv[UW * UH];
  __shared__ float sm[SW * SH];
  const int st = a.st, w = a.w, h = a.h;
  const int tiles_x = (w + FOTG_TW - 1) / FOTG_TW;
  if (a.zsync_n > 0)
    for (long k = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; k < a.zsync_n; k += (long)gridDim.x * gridDim.y * 256) a.zsync[k] = 0;
  const WgId wg = xcd_local_wg();                                // all tiles of a pair on the XCD its solver workgroup runs on
  const int pair = wg.y, tile = wg.x;
  const int x0 = (tile % tiles_x) * FOTG_TW, y0 = (tile / tiles_x) * FOTG_TH;
  const int lx = threadIdx.x % FOTG_TW, ly = threadIdx.x / FOTG_TW;
  const int i = x0 + lx, j = y0 + ly, o = j * st + i;
  const float *wx = a.single(pair, P_WX), *wy = a.single(pair, P_WY);
  const float2 *D = a.Dp(pair);
  // this pixel's own inputs first: their global-load latency overlaps the two LDS staging phases below
  const bool inimg = i < w && j < h;
  const int ic0 = inimg ? i : 0, jc0 = inimg ? j : 0;
  const PixIn<NOC> pin = data_load<NOC>(a, pair, ic0, jc0);
  const float2 duv = D[a.didx(ic0, jc0)];
  // (uu,vv) = (wx+du, wy+dv) (refine_variational.cpp:208-214), coordinates clamped like the replicate borders of the
  // 3-tap filters (image.c:436-464); du,dv come from the skewed D
  for (int k = threadIdx.x; k < UW * UH; k += 256) {
    const int jj = clampi(y0 - 2 + k / UW, h), ii = clampi(x0 - 2 + k % UW, w);
    const int q = jj * st + ii;
    const float2 d = D[a.didx(ii, jj)];
    uv[k] = make_float2(wx[q] + d.x, wy[q] + d.y);
  }
  __syncthreads();
  // compute_smoothness first half (opticalflow_aux.c:126-139) for the tile + 1 halo
  for (int k = threadIdx.x; k < SW * SH; k += 256) {
    const int sy = k / SW, sx = k % SW;
    const int c = (sy + 1) * UW + (sx + 1);
    sm[k] = smooth_w<FM>(uv[c - 1], uv[c], uv[c + 1], uv[c - UW], uv[c + UW], y0 - 1 + sy, h, quarter_alpha);
  }
  __syncthreads();
  if (i >= w || j >= h) return;
  // second half (:141-163): horiz(i) = s(i)+s(i+1) (0 in the last column), vert likewise
  const int sc = (ly + 1) * SW + (lx + 1);
  const float s_o = sm[sc];
  const float hr = (i < w - 1) ? s_o + sm[sc + 1] : 0.0f;
  const float hl = (i > 0) ? sm[sc - 1] + s_o : 0.0f;
  const float vb = (j < h - 1) ? s_o + sm[sc + SW] : 0.0f;
  const float vt = (j > 0) ? sm[sc - SW] + s_o : 0.0f;

  data_term_compute<NOC, FM>(a, pair, i, j, pin, hr, hl, vb, vt, duv.x, duv.y, half_delta_over3, half_gamma_over3);
}

// vr_data_kernel's work for the first inner iteration of a level, on the tile of a set-up workgroup (same 32 x 8 tiles): (du,dv) is
// zero, so (uu,vv) = (wx + 0, wy + 0) with (wx,wy) straight from the incoming flow (the neighbours' wx / wy planes belong to other
// workgroups of this launch and may not be written yet; same values), the planes of the tile's own pixels from global memory.
template <int NOC, int NCH, bool FM>
__device__ __forceinline__ void first_data_term(const VrArgs &a, int pair, int x0, int y0, const float *__restrict__ flow, long flow_stride,
                                                float quarter_alpha, float half_delta_over3, float half_gamma_over3)
{
  constexpr int UW = FOTG_TW + 4, UH = FOTG_TH + 4, SW = FOTG_TW + 2, SH = FOTG_TH + 2;
  __shared__ float2 uv[UW * UH];
  __shared__ float sm[SW * SH];
  const int st = a.st, w = a.w, h = a.h;
  const int lx = threadIdx.x % FOTG_TW, ly = threadIdx.x / FOTG_TW;
  const int i = x0 + lx, j = y0 + ly;
  const float *fl = flow + (size_t)pair * flow_stride;
  auto W = [&](int ii, int jj) { const float *f = fl + NCH * (size_t)(jj * w + ii); return make_float2(f[0], f[NCH - 1]); };
  for (int k = threadIdx.x; k < UW * UH; k += 256) {
    const float2 wv = W(clampi(x0 - 2 + k % UW, w), clampi(y0 - 2 + k / UW, h));
    uv[k] = make_float2(wv.x + 0.f, wv.y + 0.f);                // wx + du, wy + dv with du = dv = +0 (refine_variational.cpp:185-186, 208-214)
  }
  __syncthreads();
  for (int k = threadIdx.x; k < SW * SH; k += 256) {
    const int sy = k / SW, sx = k % SW;
    const int c = (sy + 1) * UW + (sx + 1);
    sm[k] = smooth_w<FM>(uv[c - 1], uv[c], uv[c + 1], uv[c - UW], uv[c + UW], y0 - 1 + sy, h, quarter_alpha);
  }
  __syncthreads();
  if (i >= w || j >= h) return;
  const int sc = (ly + 1) * SW + (lx + 1), o = j * st + i;
  const float s_o = sm[sc];
  const float hr = (i < w - 1) ? s_o + sm[sc + 1] : 0.0f;
  const float hl = (i > 0) ? sm[sc - 1] + s_o : 0.0f;
  const float vb = (j < h - 1) ? s_o + sm[sc + SW] : 0.0f;
  const float vt = (j > 0) ? sm[sc - SW] + s_o : 0.0f;
  PixIn<NOC> p;
#pragma unroll
  for (int c = 0; c < NOC; ++c) {
    p.Ix[c] = a.color(pair, C_IX, c)[o]; p.Iy[c] = a.color(pair, C_IY, c)[o]; p.Iz[c] = a.color(pair, C_IZ, c)[o];
    p.Ixx[c] = a.color(pair, C_IXX, c)[o]; p.Ixy[c] = a.color(pair, C_IXY, c)[o]; p.Iyy[c] = a.color(pair, C_IYY, c)[o];
    p.Ixz[c] = a.color(pair, C_IXZ, c)[o]; p.Iyz[c] = a.color(pair, C_IYZ, c)[o];
  }
  p.m = a.single(pair, P_MASK)[o];
  const float2 wc = W(i, j), wl = W(i > 0 ? i - 1 : i, j), wr = W(i < w - 1 ? i + 1 : i, j), wt = W(i, j > 0 ? j - 1 : j), wb = W(i, j < h - 1 ? j + 1 : j);
  p.wxc = wc.x; p.wxl = wl.x; p.wxr = wr.x; p.wxt = wt.x; p.wxb = wb.x;
  p.wyc = wc.y; p.wyl = wl.y; p.wyr = wr.y; p.wyt = wt.y; p.wyb = wb.y;
  data_term_compute<NOC, FM>(a, pair, i, j, p, hr, hl, vb, vt, 0.f, 0.f, half_delta_over3, half_gamma_over3);
}

// one pixel update of sor_coupled (solver.c:122-130 etc.).  du_l/du_t are the NEW left/top values, du_r/du_b the OLD
// right/bottom ones.  The reference skips the terms of missing neighbours; here their weights (psi) are exactly 0 and
// the neighbour values finite, so the skipped terms add +-0 and the sums are the same floats.
__device__ __forceinline__ float2 sor_update(float2 cur, float4 c0, float4 c1, float hl, float2 left, float2 top, float2 right, float2 bottom, float omega)
{
  const float a11 = c0.x, a12 = c0.y, b1 = c0.z, b2 = c0.w, a22 = c1.x, hr = c1.y, vb = c1.z, vt = c1.w;
  float s1 = hr * right.x, s2 = hr * right.y;
  s1 = s1 + vt * top.x;    s2 = s2 + vt * top.y;
  s1 = s1 + vb * bottom.x; s2 = s2 + vb * bottom.y;
  s1 = s1 + b1;            s2 = s2 + b2;
  const float B1 = hl * left.x + s1, B2 = hl * left.y + s2;
  float2 r;
  r.x = cur.x + omega * (a11 * B1 + a12 * B2 - cur.x);
  r.y = cur.y + omega * (a12 * B1 + a22 * B2 - cur.y);
  return r;
}

// one pixel update of sor_coupled_slow_but_readable (solver.c:28-65): sigma accumulated top, left, bottom, right from 0; du from
// the OLD dv, dv from the NEW du; cells as data_term_cell builds them with a.point.  Missing neighbours: psi = 0 and a finite
// value, the skipped terms subtract +-0.  A cell outside the image is all zero (A11 = 0): it stays untouched, like a lane / step
// that runs with omega = 0.
__device__ __forceinline__ float2 sor_update_point(float2 cur, float4 c0, float4 c1, float hl, float2 left, float2 top, float2 right, float2 bottom, float omega)
{
  const float A11 = c0.x, A12 = c0.y, b1 = c0.z, b2 = c0.w, A22 = c1.x, hr = c1.y, vb = c1.z, vt = c1.w;
  float su = 0.0f - vt * top.x, sv = 0.0f - vt * top.y;
  su = su - hl * left.x;   sv = sv - hl * left.y;
  su = su - vb * bottom.x; sv = sv - vb * bottom.y;
  su = su - hr * right.x;  sv = sv - hr * right.y;
  const float B1 = b1 - su, B2 = b2 - sv;
  float2 r;
  r.x = (1.0f - omega) * cur.x + omega / A11 * (B1 - A12 * cur.y);
  r.y = (1.0f - omega) * cur.y + omega / A22 * (B2 - A12 * r.x);
  const bool live = omega != 0.f && A11 != 0.f;
  r.x = live ? r.x : cur.x;
  r.y = live ? r.y : cur.y;
  return r;
}

__device__ __forceinline__ float dpp_wave_shr1(float v)
{
  // lane L reads lane L-1 (wave_shr:1, GFX9 DPP); lane 0 gets 0 (bound_ctrl)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}

// Lexicographic sweeps as an anti-diagonal wavefront: pixel (i,j) runs at step i+j, after its NEW left (i-1,j) and
// top (i,j-1) neighbours (step i+j-1) and before its OLD right/bottom neighbours (step i+j+1) -- the dependency
// order of the row-major loop of solver.c, hence bit-identical.
// ONE WAVE PER PAIR, no barriers: lane L owns rows K*L..K*L+K-1.  The fresh top value of a lane's first row comes
// from lane L-1 by DPP; left values stay in registers; everything else is old data streamed from the skewed arrays
// P steps ahead through a register ring (coalesced 16-B/8-B loads), so the loop runs at the speed of its
// ~10-instruction dependency chain, not at memory latency.
// POINT: the point update of sor_coupled_slow_but_readable (FOTG_SOR_POINT) in the same dependency order.
template <int K, int P, int U, bool POINT = false>
__global__ __launch_bounds__(64) void vr_sor_kernel(VrArgs a, int sweeps, float omega)
{
  const int pair = blockIdx.x, lane = threadIdx.x;
  // lanes beyond the image own K padding rows (columns RP.. of D, all zero) and run with omega = 0: they read
  // zeros, compute zeros and write zeros, so the loop body needs no exec-mask branches at all
  const bool act = lane < a.nlanes;
  const int r0 = act ? lane * K : a.RP;
  const float om_lane = act ? omega : 0.f;
  const float4 *__restrict__ C = a.Cp(pair);
  float2 *D = a.Dp(pair);
  const int S = a.S, RP = a.RP, RPD = a.RPD;
  const int T = sweeps * S;
  struct Stage { float4 c[K][2]; float2 own[K]; float2 nxt[K + 1]; };
  Stage ring[P];
  // wave-uniform row base (SGPR) + constant per-lane byte offset (VGPR): the loads use the saddr form and the
  // per-step address arithmetic is scalar
  const char *Cb = reinterpret_cast<const char *>(C);
  char *Db = reinterpret_cast<char *>(D);
  const unsigned c_lane = (unsigned)r0 * 32u, d_lane = (unsigned)r0 * 8u;
  const unsigned c_row = (unsigned)RP * 32u, d_row = (unsigned)RPD * 8u;
  auto issue = [&](Stage &st, int row) {
    const float4 *cp = reinterpret_cast<const float4 *>(Cb + (size_t)((unsigned)row * c_row) + c_lane);
#pragma unroll
    for (int m = 0; m < K; ++m) { st.c[m][0] = cp[2 * m]; st.c[m][1] = cp[2 * m + 1]; }
    const float2 *d0 = reinterpret_cast<const float2 *>(Db + (size_t)((unsigned)row * d_row) + d_lane);
    const float2 *d1 = reinterpret_cast<const float2 *>(Db + (size_t)((unsigned)(row + 1) * d_row) + d_lane);
#pragma unroll
    for (int m = 0; m < K; ++m) st.own[m] = d0[m];
#pragma unroll
    for (int m = 0; m <= K; ++m) st.nxt[m] = d1[m];
  };
#pragma unroll
  for (int p = 0; p < P; ++p) issue(ring[p], p);          // host guarantees S >= 2P+2
  float2 prev[K];          // results of the previous step (new left / new top of the lane's own rows)
  float hl[K];
#pragma unroll
  for (int m = 0; m < K; ++m) { prev[m] = make_float2(0.f, 0.f); hl[m] = 0.f; }
  int s = 0, sp = P;       // row of the current step, row being prefetched
  // Straight-line body, one back edge: steps past T (T rounded up to a multiple of P) run with omega = 0, i.e.
  // rewrite the values they loaded, so no per-step bounds branch is needed and the compiler's s_waitcnt counting
  // stays exact (each use waits only for its own stage, P steps old).
  // U = steps per loop trip (multiple of P): the compiler drains all loads at the loop header (conservative
  // s_waitcnt merge over the back edge), so a long body amortises that one exposed memory latency
  for (int t0 = 0; t0 < T; t0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      Stage &st = ring[u % P];
      const float om = (t0 + u < T) ? om_lane : 0.f;
      float2 top0;
      top0.x = dpp_wave_shr1(prev[K - 1].x);
      top0.y = dpp_wave_shr1(prev[K - 1].y);
      float2 res[K];
#pragma unroll
      for (int m = 0; m < K; ++m)
        res[m] = POINT ? sor_update_point(st.own[m], st.c[m][0], st.c[m][1], hl[m], prev[m], m == 0 ? top0 : prev[m - 1], st.nxt[m], st.nxt[m + 1], om)
                       : sor_update(st.own[m], st.c[m][0], st.c[m][1], hl[m], prev[m], m == 0 ? top0 : prev[m - 1], st.nxt[m], st.nxt[m + 1], om);
      float2 *dst = reinterpret_cast<float2 *>(Db + (size_t)((unsigned)s * d_row) + d_lane);
#pragma unroll
      for (int m = 0; m < K; ++m) dst[m] = res[m];
#pragma unroll
      for (int m = 0; m < K; ++m) { prev[m] = res[m]; hl[m] = st.c[m][1].y; }
      issue(st, sp);
      s = (s + 1 == S) ? 0 : s + 1;
      sp = (sp + 1 == S) ? 0 : sp + 1;
    }
  }
}

// The same wavefront for levels of MORE rows than the single wave's 64 lanes x 16: one workgroup per pair, ONE sweep per launch,
// every thread walks the rows tid, tid + 1024, ... of a step, a barrier per anti-diagonal.  Nothing is carried between steps: the new
// left / top values are read back from diagonal d - 1, the weight towards the left neighbour from its cell.  Within a step only the
// cells of diagonal d are written and d - 1 / d + 1 read.  Slow and simple: what FOTG_SOR_POINT runs beyond 1024 rows, and what the
// self-healing recompute of a stalled tile pipeline runs there (no inter-workgroup waits).
template <bool POINT>
__global__ __launch_bounds__(1024) void vr_sor_tall_kernel(VrArgs a, float omega)
{
  const int pair = blockIdx.x, w = a.w, h = a.h, S = a.S, RP = a.RP, RPD = a.RPD;
  const float4 *__restrict__ C = a.Cp(pair);
  float2 *D = a.Dp(pair);
  for (int d = 0; d < S; ++d) {
    for (int j = threadIdx.x; j < h; j += blockDim.x) {
      const int i = d - j;
      if (i < 0 || i >= w) continue;
      const float4 c0 = C[((size_t)d * RP + j) * 2], c1 = C[((size_t)d * RP + j) * 2 + 1];
      const float hl = d > 0 ? C[((size_t)(d - 1) * RP + j) * 2 + 1].y : 0.f;        // (outside the image: a zero cell)
      const float2 own = D[(size_t)d * RPD + j];
      const float2 zero = make_float2(0.f, 0.f);
      const float2 left = d > 0 ? D[(size_t)(d - 1) * RPD + j] : zero, top = (d > 0 && j > 0) ? D[(size_t)(d - 1) * RPD + j - 1] : zero;
      const float2 right = D[(size_t)(d + 1) * RPD + j], bottom = D[(size_t)(d + 1) * RPD + j + 1];
      D[(size_t)d * RPD + j] = POINT ? sor_update_point(own, c0, c1, hl, left, top, right, bottom, omega)
                                     : sor_update(own, c0, c1, hl, left, top, right, bottom, omega);
    }
    __syncthreads();
  }
}

// (du,dv) of a level in LDS, skewed like the global arrays (the LDS solvers below).
//
// LDS map (dynamic): u64[0..16) = header (unused), u64[16 ..) = float2 cells of D ((S+2) rows of RPD cells: row S stays zero,
// row S+1 is scratch for the tail steps), then whatever the calling kernel appends.
extern __shared__ unsigned long long fotg_lds64[];
#define FOTG_LDS_HDR 16     // u64 units
#ifndef FOTG_FUSED_NT
#define FOTG_FUSED_NT false
#endif
#ifndef FOTG_SYNC_M
#define FOTG_SYNC_M 4          // solver steps per workgroup barrier of the barrier-stepped solver waves
#endif

__device__ __forceinline__ float2 lds_d_ld(int idx) { return __builtin_bit_cast(float2, fotg_lds64[FOTG_LDS_HDR + idx]); }
__device__ __forceinline__ void lds_d_st(int idx, float2 v) { fotg_lds64[FOTG_LDS_HDR + idx] = __builtin_bit_cast(unsigned long long, v); }


// Barrier-stepped solver waves: every wave of the workgroup executes one s_barrier per M solver steps, so the sweep /
// band stagger is a fixed number of steps instead of a progress-counter handshake:
//   wave (sweep n, band b) relaxes diagonal s at global step t = s + n*DS + b*DB.
// What a wave stores during step t' is only guaranteed visible to loads issued behind the first barrier after t', i.e.
// to loads of step t >= M*floor(t/M) > t'.  The loads of step t fetch, for step t+1,
//   the top of a band's first row (s, rb-1), stored by band b-1 at t - DB                      -> DB >= M
//   right / bottom (s+2, .),         stored by sweep n-1 [band b+1] at t + 2 - DS [+ DB]       -> DS >= M + 2 [+ DB]
// (worst case t = M*I + M - 1), rounded up to multiples of M so that all waves hit the barrier on the same local step.
// Nobody overwrites a value a slower wave still needs: the reader's load and the overwrite are always separated by at
// least DB + 2 >= M steps, hence by a barrier.  The barrier sits a third into its step (after the neighbour products)
// so the latency of the previous step's LDS store and of the loads issued behind the barrier is covered by arithmetic.
// Waves that do not solve (copy helpers, the other waves of the fused kernel) only count barriers.
// CL: the system cells come from an LDS copy `lc` of the pair's skewed C (the fused kernel's data phase writes it there)
// instead of the register-ring prefetch from global memory.
// PK: the (du,dv) pair as packed f32 (v_pk_mul_f32 / v_pk_add_f32 round each half like the scalar op): 15 instead of 28
// arithmetic instructions per step.  Measured: a win when every solver wave has a SIMD to itself (fused levels, 3 waves:
// -13 %), a loss when two solver waves share a SIMD (level 4, 6 waves: +8 %; the packed ops occupy the SIMD twice as long).
// NOB: the caller guarantees a single band (nbands <= 1): the row above the wave's first row does not exist, lane 0's DPP
// source is out of range and reads 0 (bound_ctrl) -- no top value from LDS, no select.
template <int P, int U, bool NT, int M, bool CL = false, bool PK = false, bool NOB = false>
__device__ __forceinline__ void sor_sync_wave(const VrArgs &a, int pair, float omega, int wv, int lane, const float4 *lc = nullptr)
{
  constexpr int UT = CL ? 2 * P : P;                             // unroll of the tail loop (no prefetch ring with CL: longer, fewer back edges)
  static_assert(U % M == 0 && UT % M == 0 && U % P == 0, "barrier phase must be a compile-time property of the unrolled step");
  const int NB = a.nbands > 0 ? a.nbands : 1;
  constexpr int DB = M, DS1 = ((M + 2 + M - 1) / M) * M, DSB = ((2 * M + 2 + M - 1) / M) * M;
  const int DS = NB > 1 ? DSB : DS1;
  const int S = a.S, RP = a.RP, RPD = a.RPD;
  int E = 0;                                                     // steps every solver wave executes (the last ones are no-ops)
  while (E + U + P <= S) E += U;
  while (E < S) E += UT;
  const int omax = a.nsweeps > 0 ? (a.nsweeps - 1) * DS + (NB - 1) * DB : 0;
  if (wv >= a.nsweeps * NB) {
    for (int t = 0; t < (E + omax) / M; ++t) asm volatile("s_barrier" ::: "memory");
    return;
  }
  const int n = wv / NB, b = wv % NB, off = n * DS + b * DB;
  const int rb = b * a.band_rows, nrows = NB > 1 ? (rb + a.band_rows <= a.h ? a.band_rows : a.h - rb) : a.h;
  const bool first_row = lane == 0;
  const bool act = lane < nrows;
  const int r = act ? rb + lane : RP;                            // idle lanes park on the zero padding cells with omega = 0
  const float om_lane = act ? omega : 0.f;
  struct Stage { float4 c0, c1; };
  Stage ring[P];
  typedef float vf4 __attribute__((ext_vector_type(4)));
  const unsigned c_row = (unsigned)RP * 32u;
  const char *cptr = reinterpret_cast<const char *>(a.Cp(pair)) + (size_t)r * 32u;
  const char *cend = cptr + (size_t)S * c_row;
  auto load_c = [&](Stage &st, const char *ptr) {
    const vf4 *cp = reinterpret_cast<const vf4 *>(ptr);
    const vf4 x = NT ? __builtin_nontemporal_load(cp) : cp[0], y = NT ? __builtin_nontemporal_load(cp + 1) : cp[1];
    st.c0 = make_float4(x.x, x.y, x.z, x.w); st.c1 = make_float4(y.x, y.y, y.z, y.w);
  };
  Stage cnx;
  const int lc_pl = a.SC * RP + 1;                               // two planes (c0 | c1) of 16-byte cells: conflict-free ds_read_b128
  if constexpr (CL) {
    lc += r;
    cnx.c0 = lc[0]; cnx.c1 = lc[lc_pl];
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) { load_c(ring[p], cptr); cptr += c_row; }
  }
  for (int t = 0; t < off / M; ++t) asm volatile("s_barrier" ::: "memory");
  int lrow = r;                                                  // LDS cell (diagonal s, my row)
  int trow = b > 0 ? rb - 1 : RP;                                // LDS cell (diagonal s, row above the band) [band 0: a zero cell]
  float2 own = lds_d_ld(lrow), nxr = lds_d_ld(lrow + RPD), nxb = lds_d_ld(lrow + RPD + 1);
  float2 tpl = make_float2(0.f, 0.f), prev = make_float2(0.f, 0.f);
  float hl = 0.f;
  const int rpd2 = 2 * RPD;
  auto step = [&](auto tail_tag, int u, int s) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    Stage &st = ring[CL ? 0 : u % P];
    if constexpr (CL) st = cnx;
    const float om = (!TAIL || s < S) ? om_lane : 0.f;
    // DPP reads must run with all lanes enabled: compute, pin, then select.  (wave_shr:1 with old = tpl and no
    // bound_ctrl would drop the two selects, but ties the DPP to the LDS load of tpl: measured 3 % slower.)
    float dx = dpp_wave_shr1(prev.x), dy = dpp_wave_shr1(prev.y);
    asm volatile("" : "+v"(dx), "+v"(dy));
    const float2 top = NOB ? make_float2(dx, dy) : (first_row ? tpl : make_float2(dx, dy));
    // sor_update()'s expression order
    typedef float v2f __attribute__((ext_vector_type(2)));
    const float a11 = st.c0.x, a12 = st.c0.y, b1 = st.c0.z, b2 = st.c0.w, a22 = st.c1.x, hr = st.c1.y, vb = st.c1.z, vt = st.c1.w;
    float s1, s2;
    if constexpr (PK) {
      const v2f vnxr = {nxr.x, nxr.y}, vnxb = {nxb.x, nxb.y}, vtop = {top.x, top.y};
      v2f sv = hr * vnxr;
      sv = sv + vt * vtop;
      sv = sv + vb * vnxb;
      s1 = sv.x; s2 = sv.y;
    } else {
      s1 = hr * nxr.x;       s2 = hr * nxr.y;
      s1 = s1 + vt * top.x;  s2 = s2 + vt * top.y;
      s1 = s1 + vb * nxb.x;  s2 = s2 + vb * nxb.y;
    }
    if (u % M == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(s1), "+v"(s2) :: "memory");   // tied to (s1,s2): placed here
    float2 nr, nb, tp = make_float2(0.f, 0.f);
    if (!TAIL) {
      nr = lds_d_ld(lrow + rpd2); nb = lds_d_ld(lrow + rpd2 + 1);  // diagonal s+2: right / bottom of step s+1
      if constexpr (!NOB) tp = lds_d_ld(trow);                     // (diagonal s, row above the band): top of step s+1
    } else {
      const int d2 = s + 2 < S + 1 ? s + 2 : S + 1, d0 = s < S ? s : S;
      nr = lds_d_ld(d2 * RPD + r); nb = lds_d_ld(d2 * RPD + r + 1);
      if constexpr (!NOB) tp = lds_d_ld(d0 * RPD + (b > 0 ? rb - 1 : RP));
    }
    if constexpr (CL) {                                          // cells of diagonal s+1 (diagonal S is all zero)
      if (!TAIL || s + 1 <= S) lc += RP;
      cnx.c0 = lc[0]; cnx.c1 = lc[lc_pl];
    }
    float2 res;
    if constexpr (PK) {
      const v2f a1 = {a11, a12}, bb = {b1, b2}, vprev = {prev.x, prev.y}, vown = {own.x, own.y};
      v2f sv = {s1, s2};
      sv = sv + bb;
      const v2f B = hl * vprev + sv;                             // (B1, B2)
      const v2f pa = a1 * B;                                     // (a11 B1, a12 B2)
      v2f t = {pa.x + pa.y, a12 * B.x + a22 * B.y};
      t = t - vown;
      const v2f rs = vown + om * t;
      res = make_float2(rs.x, rs.y);
    } else {
      s1 = s1 + b1;          s2 = s2 + b2;
      const float B1 = hl * prev.x + s1, B2 = hl * prev.y + s2;
      res.x = own.x + om * (a11 * B1 + a12 * B2 - own.x);
      res.y = own.y + om * (a12 * B1 + a22 * B2 - own.y);
    }
    lds_d_st((!TAIL || s < S) ? lrow : (S + 1) * RPD + r, res);
    prev = res; hl = hr; own = nxr; nxr = nr; nxb = nb; tpl = tp;
    if constexpr (!CL) {
      load_c(st, cptr);
      if (!TAIL || cptr < cend) cptr += c_row;
    }
    lrow += RPD; trow += RPD;
  };
  int t0 = 0;
  for (; t0 + U + P <= S; t0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) step(std::false_type{}, u, t0 + u);
  }
  for (; t0 < S; t0 += UT) {
#pragma unroll
    for (int u = 0; u < UT; ++u) step(std::true_type{}, u, t0 + u);
  }
  for (int t = off / M; t < omax / M; ++t) asm volatile("s_barrier" ::: "memory");
}

// stand-alone launch of one sor_coupled call: D global -> LDS, sweeps, LDS -> global
template <int P, int U>
__global__ __launch_bounds__(1024) void vr_sor_pipe_kernel(VrArgs a, float omega)
{
  const int pair = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float2 *Dg = a.Dp(pair);
  const int ncell = (a.S + 1) * a.RPD;
  // global -> LDS, 16 B per lane and 4 loads in flight per lane (RPD is even, so cells pair up)
  {
    const float4 *g4 = reinterpret_cast<const float4 *>(Dg);
    const int n2 = ncell >> 1;
    constexpr int Q = 8;                                          // loads in flight per lane
    for (int k = threadIdx.x; k < n2; k += Q * blockDim.x) {
      float4 v[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) { const int kk = k + q * blockDim.x; v[q] = kk < n2 ? g4[kk] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int kk = k + q * blockDim.x;
        if (kk < n2) { lds_d_st(2 * kk, make_float2(v[q].x, v[q].y)); lds_d_st(2 * kk + 1, make_float2(v[q].z, v[q].w)); }
      }
    }
  }
  __syncthreads();
  sor_sync_wave<P, U, false, FOTG_SYNC_M>(a, pair, omega, wv, lane);
  __syncthreads();
  {
    float4 *g4 = reinterpret_cast<float4 *>(Dg);
    const int n2 = ncell >> 1;
    for (int k = threadIdx.x; k < n2; k += blockDim.x) {
      const float2 a0 = lds_d_ld(2 * k), a1 = lds_d_ld(2 * k + 1);
      g4[k] = make_float4(a0.x, a0.y, a1.x, a1.y);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Streaming launch of one sor_coupled call for levels whose (du,dv) AND system do not fit LDS together (1080p level 4):
// the anti-diagonals travel through two LDS rings, one slot per diagonal: the D ring (RD cells of 8 B per slot) and the C
// ring (two planes of RCW 16-byte cells per slot).  Besides the barrier-stepped solver waves (sor_sync_wave's schedule,
// M steps per barrier) the workgroup has
//   * three LOADER waves (the D row + rows 0..63 of C plane 0 | rows 0..63 of plane 1 + rows 64.. of plane 0 | rows 64.. of
//     plane 1): in barrier interval I they issue the direct-to-LDS loads (global_load_lds_dwordx4, no VGPRs) of the M diagonals
//     of chunk I + LI and then wait (counted vmcnt) until only that newest chunk is in flight.  A chunk is thus complete
//     before barrier I+2 and visible to everybody from interval c - 1 on; the first solver wave reads diagonal
//     s+2 <= M*I + M + 1 in interval I  ->  LI = 3.
//   * a WRITER wave: in interval J it copies the D rows of chunk J - WO back to global memory, WO = omax/M + 1 (the last
//     sweep relaxed them in interval J - 1 at the latest, and its stores landed before barrier J).
// A D slot therefore lives LI + WO + 1 intervals and a C slot LI + omax/M + 1: 36 + 32 slots = 93 KB for 120x68 with
// 3 sweeps of two rows per lane (K = 2).  Compared with vr_sor_pipe_kernel this removes the D copy-in / copy-out phases (they overlap the
// solve) and the two global loads per solver wave and step (the system now comes out of LDS), and the LDS footprint no
// longer grows with the level's width.
// MEASURED (MI355X, 64 x 1080p level 4, XCD-local placement): 29.5 us per call with K = 2 (one solver wave per sweep, two
// rows per lane, packed f32; 115 ns per diagonal) against 35.5 us for vr_sor_pipe_kernel (six solver waves on four SIMDs)
// and 34 us for K = 1 here.  The LDS footprint does not grow with the level's width.
// ------------------------------------------------------------------------------------------------------------------
#ifndef FOTG_STREAM_DBG
#define FOTG_STREAM_DBG 0      // timing-only elimination builds (wrong results): 1 loaders issue nothing in the loop, 2 writer without stores, 4 solver waves without arithmetic
#endif
template <int RD, int RCW>
struct StreamGeom {
  static constexpr int DB = RD * 8, CB = RCW * 16, CSLOT = 2 * CB;         // bytes
  static constexpr int LI = 3;                                             // load lead in barrier intervals (see the loaders)
};

__device__ __forceinline__ char *lds_bytes() { return reinterpret_cast<char *>(fotg_lds64); }

// one 16-byte-per-lane direct load: lane L's 16 bytes at `src` land at lds_dst + 16 L (lds_dst wave-uniform)
__device__ __forceinline__ void glds16(const void *src, unsigned lds_dst_byte)
{
  typedef __attribute__((address_space(1))) const void gvoid;
  typedef __attribute__((address_space(3))) void lvoid;
  __builtin_amdgcn_global_load_lds((gvoid *)src, (lvoid *)(lds_bytes() + lds_dst_byte), 16, 0, 0);
}

// K = 2: one solver wave per sweep, two consecutive rows per lane, packed-f32 arithmetic.  With the system in LDS a
// second row costs no prefetch registers, its top neighbour is the lane's own previous result, the two rows are
// independent within a step (better issue than one dependent chain), and three solver waves have a SIMD each (the
// loaders and the writer sit on the fourth).
// FMA: the cell update with fused multiply-adds (fotg_params::fast_math, the tolerance mode; see vr_sor_tile_kernel)
template <int RD, int RCW, int M, int U, bool FMA = false>
__global__ __launch_bounds__(1024) void vr_sor_stream_kernel(VrArgs a, float omega)
{
  using GEO = StreamGeom<RD, RCW>;
  constexpr int DB = GEO::DB, CB = GEO::CB, CSLOT = GEO::CSLOT, LI = GEO::LI;
  constexpr int UT = 8;
  static_assert(U % M == 0 && UT % M == 0, "barrier phase is a compile-time property of the unrolled step");
  static_assert(RCW > 64 && RCW <= 128, "two direct loads per C plane and diagonal");
  const int pair = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int NB = 1;
  constexpr int DBS = M, DS1 = ((M + 2 + M - 1) / M) * M, DSB = ((2 * M + 2 + M - 1) / M) * M;
  const int DS = NB > 1 ? DSB : DS1;
  const int S = a.S, RP = a.RP, RPD = a.RPD;
  int E = 0;
  while (E + U <= S) E += U;
  while (E < S) E += UT;
  const int omax = a.nsweeps > 0 ? (a.nsweeps - 1) * DS + (NB - 1) * DBS : 0;
  const int WO = omax / M + 1;
  const int RDN = M * (LI + WO + 1), RCN = RDN - M;               // ring slots
  const int NI = (E + omax) / M + 1;                              // barrier intervals every wave goes through
  const int nsolver = a.nsweeps * NB;
  // roles: solver waves 0..2, helpers on SIMD 3 (waves 3, 7, 11, 15)
#ifndef FOTG_STREAM_WV
#define FOTG_STREAM_WV 3, 7, 11, 15
#endif
  constexpr int wv_roles[4] = {FOTG_STREAM_WV};
  const int wvL0 = wv_roles[0], wvL1 = wv_roles[1], wvL2 = wv_roles[2], wvW = wv_roles[3];
  const unsigned CRING = (unsigned)RCN * CSLOT, DRING = (unsigned)RDN * DB;     // bytes
  const unsigned DBASE = CRING, DUMP = CRING + DRING;             // C ring | D ring | one spare D row for the no-op tail steps
  char *Dg = reinterpret_cast<char *>(a.Dp(pair));
  const char *Cg = reinterpret_cast<const char *>(a.Cp(pair));
  auto ld_f2 = [&](unsigned off) { return *reinterpret_cast<const float2 *>(lds_bytes() + off); };
  auto st_f2 = [&](unsigned off, float2 v) { *reinterpret_cast<float2 *>(lds_bytes() + off) = v; };
  auto ld_f4 = [&](unsigned off) { return *reinterpret_cast<const float4 *>(lds_bytes() + off); };

  // Direct loads of the next M diagonals (diagonals past S re-read the all-zero diagonal S), split over three loader waves
  // so that none of them issues more than 2 loads per diagonal (an LDS-DMA load costs its wave ~27 ns of issue):
  //   loader 0: the D row, C plane 0 rows 0..63      loader 1: C plane 1 rows 0..63, C plane 0 rows 64..RP
  //   loader 2: C plane 1 rows 64..RP                (row RP = what the idle lanes read; finite)
  // Every load has at least one active lane (RP >= 64 is a launch condition), so the per-chunk instruction counts are exact
  // and a counted vmcnt can leave the newest chunk in flight across the barrier.
  const int lrole = wv == wvL0 ? 0 : wv == wvL1 ? 1 : 2;
  int ld = 0;
  unsigned ldslot = 0, lcslot = 0;
  const char *pD = Dg + lane * 16;                                // lane's 16 bytes of the D row
  const char *pC = Cg + lane * 32;                                // lane's row of C (both planes interleaved in global memory)
  const size_t dstep = (size_t)RPD * 8, cstep = (size_t)RP * 32;
  const bool inD = lane < RPD / 2, inC2 = lane + 64 <= RP;
  auto issue_chunk = [&](auto role_tag) {
    constexpr int ROLE = decltype(role_tag)::value;
#pragma unroll
    for (int k = 0; k < M; ++k) {
      if (ROLE == 0) { if (inD) glds16(pD, DBASE + ldslot); glds16(pC, lcslot); }
      if (ROLE == 1) { glds16(pC + 16, lcslot + CB); if (inC2) glds16(pC + 64 * 32, lcslot + 1024); }
      if (ROLE == 2) { if (inC2) glds16(pC + 64 * 32 + 16, lcslot + CB + 1024); }
      if (ld < S) { pD += dstep; pC += cstep; }
      ++ld;
      ldslot += DB; if (ldslot == DRING) ldslot = 0;
      lcslot += CSLOT; if (lcslot == CRING) lcslot = 0;
    }
  };
  // Interval I of a loader: barrier I, issue chunk I + LI, wait until only that newest chunk is still in flight.  Chunk c is
  // thus complete before barrier c - LI + 2 and visible to everybody from interval c - LI + 2 = c - 1 on (LI = 3).
  auto loader = [&](auto role_tag) {
    constexpr int ROLE = decltype(role_tag)::value;
    constexpr int NG = M * (ROLE == 2 ? 1 : 2);                   // loads per chunk of this loader
    for (int c = 0; c < LI; ++c) issue_chunk(role_tag);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int I = 0; I < NI; ++I) {
      asm volatile("s_barrier" ::: "memory");
      if (!(FOTG_STREAM_DBG & 1)) issue_chunk(role_tag);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NG) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // nothing may land after the workgroup's LDS is released
  };
  if (threadIdx.x < RD) st_f2(DUMP + threadIdx.x * 8, make_float2(0.f, 0.f));
  if (wv == wvL0) { loader(std::integral_constant<int, 0>{}); return; }
  if (wv == wvL1) { loader(std::integral_constant<int, 1>{}); return; }
  if (wv == wvL2) { loader(std::integral_constant<int, 2>{}); return; }
  __syncthreads();
  (void)lrole;

  if (wv == wvW) {                                                // ---------------- writer ----------------
    int wd = 0;
    unsigned wslot = 0;
    for (int I = 0; I < NI; ++I) {
      asm volatile("s_barrier" ::: "memory");
      if (I >= WO) {
#pragma unroll
        for (int k = 0; k < M; ++k) {
          if (!(FOTG_STREAM_DBG & 2) && wd < S && lane < RPD / 2) *reinterpret_cast<float4 *>(Dg + ((size_t)wd * RPD) * 8 + lane * 16) = ld_f4(DBASE + wslot + lane * 16);
          ++wd; wslot += DB; if (wslot == DRING) wslot = 0;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }
  if (wv >= nsolver) return;      // spare waves leave: a barrier counts the waves that have not ended, and their wave slots (9 of the
                                  // workgroup's 16) go back to the kernels of the other batches in flight

  {
    // ---------------- solver wave of sweep n, rows 2L and 2L+1 per lane ----------------
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int off = wv * DS;
    const int nl = (a.h + 1) >> 1;                                // lanes with rows; the others stay disabled for the whole solve
    const float om0 = omega, om1 = (2 * lane + 1 < a.h) ? omega : 0.f;
    const unsigned vD = DBASE + (unsigned)lane * 16, vC = (unsigned)lane * 32;
    for (int t = 0; t < off / M; ++t) asm volatile("s_barrier" ::: "memory");
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
        // vt * top with vt read in place (the high half of the register pair the cell was loaded into: the compiler copies it to
        // a pair of its own first)
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
        v2f t = {t0, t1};
        t = t - own;
        return own + om * t;
      };
      auto step2 = [&](auto tail_tag, int u, int s) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const float o0 = (!TAIL || s < S) ? om0 : 0.f, o1 = (!TAIL || s < S) ? om1 : 0.f;
        // top of row 2L: row 2L-1's result of the previous step, in lane L-1 (lane 0: no row above, 0)
        const v2f top0 = {dpp_wave_shr1(p1.x), dpp_wave_shr1(p1.y)};
        if (u % M == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const float4 nnx = ld_f4(d2 + vD);                        // diagonal s+2
        const float2 nnb = ld_f2(d2 + vD + 16);
        const float4 na0 = ld_f4(c1o + vC), nb0 = ld_f4(c1o + vC + 16), na1 = ld_f4(c1o + vC + CB), nb1 = ld_f4(c1o + vC + CB + 16);
        const v2f own0 = {ow.x, ow.y}, own1 = {ow.z, ow.w}, r0 = {nx.x, nx.y}, r1 = {nx.z, nx.w}, bt1 = {nb.x, nb.y};
        const v2f q0 = (FOTG_STREAM_DBG & 4) ? own0 + top0 : relax(own0, ca0, ca1, hl0, p0, top0, r0, r1, o0);
        const v2f q1 = (FOTG_STREAM_DBG & 4) ? own1 + p0 : relax(own1, cb0, cb1, hl1, p1, p0, r1, bt1, o1);      // its top (s-1, 2L) is this lane's previous row-0 result
        *reinterpret_cast<float4 *>(lds_bytes() + ((!TAIL || s < S) ? d0 + vD : DUMP + (unsigned)lane * 16)) = make_float4(q0.x, q0.y, q1.x, q1.y);
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
    }
    for (int t = off / M; t < omax / M + 1; ++t) asm volatile("s_barrier" ::: "memory");
    return;
  }
}

// the same half-sweeps on a level whose (du,dv) -- and, with CL, whose system -- sit in LDS (the fused per-level kernel)
template <bool CL>
__device__ __forceinline__ void rb_sweeps_lds(const VrArgs &a, int pair, int sweeps, float omega, const float4 *lc)
{
  const int RP = a.RP, RPD = a.RPD, S = a.S;
  const float4 *Cg = a.Cp(pair);
  const int pl1 = a.SC * RP + 1;
  for (int it = 0; it < sweeps; ++it)
    for (int col = 0; col < 2; ++col) {
      const int nd = (S - col + 1) / 2;
      for (int k = threadIdx.x; k < nd * RP; k += blockDim.x) {
        const int s = col + 2 * (k / RP), r = k - (k / RP) * RP, i = s - r;
        if (r >= a.h || i < 0 || i >= a.w) continue;
        const int c = s * RP + r, d = s * RPD + r;
        float4 c0, c1;
        float hl;
        if constexpr (CL) { c0 = lc[c]; c1 = lc[c + pl1]; hl = i > 0 ? lc[c - RP + pl1].y : 0.f; }
        else { c0 = Cg[2 * (size_t)c]; c1 = Cg[2 * (size_t)c + 1]; hl = i > 0 ? Cg[2 * (size_t)(c - RP) + 1].y : 0.f; }
        const float2 z = make_float2(0.f, 0.f);
        const float2 nl = s > 0 ? lds_d_ld(d - RPD) : z, nt = (s > 0 && r > 0) ? lds_d_ld(d - RPD - 1) : z;
        const float2 nr = lds_d_ld(d + RPD), nb = lds_d_ld(d + RPD + 1);
        lds_d_st(d, sor_update(lds_d_ld(d), c0, c1, hl, nl, nt, nr, nb, omega));
      }
      __syncthreads();
    }
}

// The whole fixed-point loop of one level in ONE launch, one workgroup per pair (refine_variational.cpp:182-221):
//   repeat inner times { smoothness, data term + sub_laplacian + block inverse -> system C ; sor_coupled } ; flow = w + d.
// (du,dv) never leave LDS; the smoothness weights live in an LDS plane; C goes through global memory (L2) to the
// solver waves of the same workgroup.  Replaces 2*inner + 2 launches (and the LDS copy-in/out of D) per level.
// CL (barrier-stepped waves only): the system cells C stay in LDS too -- the data phase writes them there and the solver
// waves read them with ds_read_b128, so nothing but the level's input planes crosses the CU boundary inside the loop.
// FM: the data term in the tolerance mode's arithmetic (fotg_params::fast_math, varref_dataterm.inc.h)
template <int NOC, int P, int U, bool CL, bool RES = false, int NT = 512, bool FM = false>
__global__ __launch_bounds__(NT) void vr_inner_fused_kernel(VrArgs a, int inner, float quarter_alpha, float half_delta_over3,
                                                             float half_gamma_over3, float omega, float *__restrict__ flow, long flow_stride,
                                                             const float *__restrict__ I0, const float *__restrict__ I1, long img_stride, int tw, int pad)
{
  const int pair = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int w = a.w, h = a.h, st = a.st, S = a.S, RPD = a.RPD;
  const int ncell = (S + 2) * RPD;
  float *sm = reinterpret_cast<float *>(fotg_lds64 + FOTG_LDS_HDR + ncell);            // smoothness plane [h][w]
  float4 *lc = reinterpret_cast<float4 *>(fotg_lds64 + FOTG_LDS_HDR + ncell + 2 * ((w * h + 3) / 4));   // CL: skewed C, SC x RP cells (+1)
  const int nlc = CL ? (a.SC * a.RP + 1) * 2 : 0;
  const float *wx = a.single(pair, P_WX), *wy = a.single(pair, P_WY);
  for (int k = threadIdx.x; k < ncell; k += blockDim.x) lds_d_st(k, make_float2(0.f, 0.f));     // image_erase(du), (dv) (:185-186)
  for (int k = threadIdx.x; k < nlc; k += blockDim.x) lc[k] = make_float4(0.f, 0.f, 0.f, 0.f);  // cells outside the image stay zero
  // set-up stages of the level (refine_variational.cpp:182-183): warp + mask + mean/difference, then the derivative planes
  const int npx0 = w * h;
  if (CL && 4 * NOC * npx0 <= 4 * nlc) {
    // the planes the 5-tap filters read (average, difference, Ix, Iy) are staged in the still unused LDS area of the system C,
    // pitch w: no global round trip between the three stages.  Everything is also stored to the global planes (later loads, taps).
    float *sa = reinterpret_cast<float *>(lc);
    __syncthreads();                                             // (the zero fill of lc above is redone below)
    for (int px = threadIdx.x; px < npx0; px += blockDim.x) {
      const int i = px % w, j = px / w;
      const PrepVal<NOC> v = prep_values<NOC>(a, pair, i, j, I0, I1, img_stride, tw, pad, flow, flow_stride);
      prep_store<NOC>(a, pair, i, j, v);
#pragma unroll
      for (int c = 0; c < NOC; ++c) { sa[c * npx0 + px] = v.avg[c]; sa[(NOC + c) * npx0 + px] = v.iz[c]; }
    }
    __syncthreads();
    for (int px = threadIdx.x; px < npx0; px += blockDim.x) {
      const int i = px % w, j = px / w, o = j * st + i;
#pragma unroll
      for (int c = 0; c < NOC; ++c) {
        const float *A = sa + c * npx0, *Z = sa + (NOC + c) * npx0;
        const float ix = conv_h5(A + j * w, i, w), iy = conv_v5(A + i, j, h, w);
        a.color(pair, C_IX, c)[o] = ix; a.color(pair, C_IY, c)[o] = iy;
        a.color(pair, C_IXZ, c)[o] = conv_h5(Z + j * w, i, w); a.color(pair, C_IYZ, c)[o] = conv_v5(Z + i, j, h, w);
        sa[(2 * NOC + c) * npx0 + px] = ix; sa[(3 * NOC + c) * npx0 + px] = iy;
      }
    }
    __syncthreads();
    for (int px = threadIdx.x; px < npx0; px += blockDim.x) {
      const int i = px % w, j = px / w, o = j * st + i;
#pragma unroll
      for (int c = 0; c < NOC; ++c) {
        const float *X = sa + (2 * NOC + c) * npx0, *Y = sa + (3 * NOC + c) * npx0;
        a.color(pair, C_IXX, c)[o] = conv_h5(X + j * w, i, w);
        a.color(pair, C_IXY, c)[o] = conv_v5(X + i, j, h, w);
        a.color(pair, C_IYY, c)[o] = conv_v5(Y + i, j, h, w);
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nlc; k += blockDim.x) lc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
  } else {
    for (int px = threadIdx.x; px < w * h; px += blockDim.x) prep_pixel<NOC>(a, pair, px % w, px / w, I0, I1, img_stride, tw, pad, flow, flow_stride);
    __syncthreads();
    for (int px = threadIdx.x; px < w * h; px += blockDim.x) deriv1_pixel<NOC>(a, pair, px % w, px / w);
    __syncthreads();
    for (int px = threadIdx.x; px < w * h; px += blockDim.x) deriv2_pixel<NOC>(a, pair, px % w, px / w);
    __syncthreads();
  }
  constexpr int B = 2048 / NT;                                   // pixels per thread whose global loads are in flight together
  const int npx = w * h, nth = blockDim.x;
  // Gray levels of at most B pixels per thread: everything the loop reads from global memory (mask, derivative planes,
  // wx / wy with their neighbours) is constant over the inner iterations -> load it once and keep it in registers;
  // inside the loop only LDS is touched.  (RGB has 35 values per pixel: reloaded every iteration.)
  // (template flag RES, set by the host when NOC == 1 and npx <= B * NT = 2048)
  constexpr bool resident = RES;
  PixIn<NOC> rp[RES ? B : 1];
  if constexpr (resident) {
#pragma unroll
    for (int q = 0; q < B; ++q) {
      const int px = threadIdx.x + q * nth < npx ? threadIdx.x + q * nth : npx - 1;
      rp[q] = data_load<NOC>(a, pair, px % w, px / w);
    }
  }
  // resident variant: the pixels' index arithmetic is the same in every inner iteration -- (du,dv) cell, system cell, image
  // index and border flags are computed once (the compiler does not hoist them out of the divergent pixel loops by itself)
  int pdc[RES ? B : 1], plc[RES ? B : 1], ppx[RES ? B : 1];      // D cell of the pixel, C cell, px | flags << 24
  if constexpr (resident) {
#pragma unroll
    for (int q = 0; q < B; ++q) {
      const int px = threadIdx.x + q * nth;
      const bool ok = px < npx;
      const int pc = ok ? px : npx - 1, i = pc % w, j = pc / w;
      pdc[q] = (i + j) * RPD + j;
      plc[q] = (i + j) * a.RP + j;
      ppx[q] = pc | (ok ? 1 << 24 : 0) | (i > 0 ? 1 << 25 : 0) | (i < w - 1 ? 1 << 26 : 0) | (j > 0 ? 1 << 27 : 0) | (j < h - 1 ? 1 << 28 : 0);
    }
  }
  for (int it = 0; it < inner; ++it) {
    if constexpr (resident) {
      // opaque to the optimiser: otherwise every loop-invariant product of compute_data is hoisted and kept live too
#pragma unroll
      for (int q = 0; q < B; ++q) {
        PixIn<NOC> &r = rp[q];
#pragma unroll
        for (int c = 0; c < NOC; ++c)
          asm volatile("" : "+v"(r.Ix[c]), "+v"(r.Iy[c]), "+v"(r.Iz[c]), "+v"(r.Ixx[c]), "+v"(r.Ixy[c]), "+v"(r.Iyy[c]), "+v"(r.Ixz[c]), "+v"(r.Iyz[c]));
        asm volatile("" : "+v"(r.m), "+v"(r.wxc), "+v"(r.wxl), "+v"(r.wxr), "+v"(r.wxt), "+v"(r.wxb), "+v"(r.wyc), "+v"(r.wyl), "+v"(r.wyr), "+v"(r.wyt), "+v"(r.wyb));
      }
    }
    if constexpr (resident) {                                    // compute_smoothness first half (:126-139), indices precomputed
#pragma unroll
      for (int q = 0; q < B; ++q) {
        const int f = ppx[q];
        if (!(f & (1 << 24))) continue;
        const PixIn<NOC> &r = rp[q];
        const int c = pdc[q];
        const int cl = (f & (1 << 25)) ? c - RPD : c, cr = (f & (1 << 26)) ? c + RPD : c;      // clamped neighbours (replicate)
        const int ct = (f & (1 << 27)) ? c - RPD - 1 : c, cb = (f & (1 << 28)) ? c + RPD + 1 : c;
        const float2 d0 = lds_d_ld(c), dl = lds_d_ld(cl), dr = lds_d_ld(cr), dt = lds_d_ld(ct), db = lds_d_ld(cb);
        const int px = f & 0xFFFFFF, j = (f & (1 << 27)) ? ((f & (1 << 28)) ? 1 : h - 1) : 0;   // smooth_w only tests j == 0 / j == h-1
        sm[px] = smooth_w<FM>(make_float2(r.wxl + dl.x, r.wyl + dl.y), make_float2(r.wxc + d0.x, r.wyc + d0.y), make_float2(r.wxr + dr.x, r.wyr + dr.y),
                          make_float2(r.wxt + dt.x, r.wyt + dt.y), make_float2(r.wxb + db.x, r.wyb + db.y), j, h, quarter_alpha);
      }
    } else
    for (int k0 = threadIdx.x; k0 < npx; k0 += B * nth) {        // compute_smoothness first half (:126-139)
      float gx[B][5], gy[B][5];                                  // wx, wy at centre, left, right, top, bottom (clamped)
      int qi[B], qj[B];
#pragma unroll
      for (int q = 0; q < B; ++q) {
        const int px = k0 + q * nth < npx ? k0 + q * nth : npx - 1;
        const int i = px % w, j = px / w;
        qi[q] = i; qj[q] = j;
        if constexpr (resident) {
          const PixIn<NOC> &r = rp[q];
          gx[q][0] = r.wxc; gx[q][1] = r.wxl; gx[q][2] = r.wxr; gx[q][3] = r.wxt; gx[q][4] = r.wxb;
          gy[q][0] = r.wyc; gy[q][1] = r.wyl; gy[q][2] = r.wyr; gy[q][3] = r.wyt; gy[q][4] = r.wyb;
          continue;
        }
        const int jc[5] = {j, j, j, clampi(j - 1, h), clampi(j + 1, h)}, ic[5] = {i, clampi(i - 1, w), clampi(i + 1, w), i, i};
#pragma unroll
        for (int t = 0; t < 5; ++t) { gx[q][t] = wx[jc[t] * st + ic[t]]; gy[q][t] = wy[jc[t] * st + ic[t]]; }
      }
#pragma unroll
      for (int q = 0; q < B; ++q) {
        if (k0 + q * nth >= npx) continue;
        const int i = qi[q], j = qj[q];
        const int jc[5] = {j, j, j, clampi(j - 1, h), clampi(j + 1, h)}, ic[5] = {i, clampi(i - 1, w), clampi(i + 1, w), i, i};
        float2 uvv[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          const float2 d = lds_d_ld((ic[t] + jc[t]) * RPD + jc[t]);
          uvv[t] = make_float2(gx[q][t] + d.x, gy[q][t] + d.y);  // (uu,vv) = (wx+du, wy+dv)
        }
        sm[j * w + i] = smooth_w<FM>(uvv[1], uvv[0], uvv[2], uvv[3], uvv[4], j, h, quarter_alpha);
      }
    }
    __syncthreads();
    if constexpr (resident) {                                    // second half (:141-163) + data term + laplacian + inverse
#pragma unroll
      for (int q = 0; q < B; ++q) {
        const int f = ppx[q];
        if (!(f & (1 << 24))) continue;
        const int px = f & 0xFFFFFF;
        const float s_o = sm[px];
        const float hr = (f & (1 << 26)) ? s_o + sm[px + 1] : 0.0f;
        const float hl = (f & (1 << 25)) ? sm[px - 1] + s_o : 0.0f;
        const float vb = (f & (1 << 28)) ? s_o + sm[px + w] : 0.0f;
        const float vt = (f & (1 << 27)) ? sm[px - w] + s_o : 0.0f;
        const float2 duv = lds_d_ld(pdc[q]);
        // data_term_cell only tests i > 0, i < w-1, j > 0, j < h-1: hand it border-equivalent coordinates
        const int ii = (f & (1 << 25)) ? ((f & (1 << 26)) ? 1 : w - 1) : 0, jj = (f & (1 << 27)) ? ((f & (1 << 28)) ? 1 : h - 1) : 0;
        float4 c0, c1;
        data_term_cell<NOC, FM>(a, ii, jj, rp[q], hr, hl, vb, vt, duv.x, duv.y, half_delta_over3, half_gamma_over3, c0, c1);
        lc[plc[q]] = c0;
        lc[plc[q] + a.SC * a.RP + 1] = c1;
      }
    } else
    for (int k0 = threadIdx.x; k0 < npx; k0 += B * nth) {        // second half (:141-163) + data term + laplacian + inverse
      PixIn<NOC> pin[B];
#pragma unroll
      for (int q = 0; q < B; ++q) {
        const int px = k0 + q * nth < npx ? k0 + q * nth : npx - 1;
        if constexpr (resident) pin[q] = rp[q]; else pin[q] = data_load<NOC>(a, pair, px % w, px / w);
      }
#pragma unroll
      for (int q = 0; q < B; ++q) {
        const int px = k0 + q * nth;
        if (px >= npx) continue;
        const int i = px % w, j = px / w;
        const float s_o = sm[px];
        const float hr = (i < w - 1) ? s_o + sm[px + 1] : 0.0f;
        const float hl = (i > 0) ? sm[px - 1] + s_o : 0.0f;
        const float vb = (j < h - 1) ? s_o + sm[px + w] : 0.0f;
        const float vt = (j > 0) ? sm[px - w] + s_o : 0.0f;
        const float2 duv = lds_d_ld((i + j) * RPD + j);
        if constexpr (CL) {
          float4 c0, c1;
          data_term_cell<NOC, FM>(a, i, j, pin[q], hr, hl, vb, vt, duv.x, duv.y, half_delta_over3, half_gamma_over3, c0, c1);
          lc[(i + j) * a.RP + j] = c0;
          lc[(i + j) * a.RP + j + a.SC * a.RP + 1] = c1;
        } else {
          data_term_compute<NOC, FM>(a, pair, i, j, pin[q], hr, hl, vb, vt, duv.x, duv.y, half_delta_over3, half_gamma_over3);
        }
      }
    }
    __syncthreads();                                             // also drains the C stores (vmcnt(0)) before the solver reads them
    if (a.redblack) rb_sweeps_lds<CL>(a, pair, a.nsweeps, omega, lc);
    else {
      sor_sync_wave<P, U, FOTG_FUSED_NT, FOTG_SYNC_M, CL, true, true>(a, pair, omega, wv, lane, lc);   // host: single band only
      __syncthreads();
    }
  }
  float *f = flow + (size_t)pair * flow_stride;                  // refine_variational.cpp:208-221
  for (int px = threadIdx.x; px < w * h; px += blockDim.x) {
    const int i = px % w, j = px / w, o = j * st + i;
    const float2 d = lds_d_ld((i + j) * RPD + j);
    f[2 * px] = wx[o] + d.x;
    f[2 * px + 1] = wy[o] + d.y;
  }
  if (!a.taps) return;
  float2 *Dg = a.Dp(pair);                                       // test taps: global copies of (du,dv) and of the last system
  for (int k = threadIdx.x; k < (S + 1) * RPD; k += blockDim.x) Dg[k] = lds_d_ld(k);
  if constexpr (CL) {
    float4 *Cg = a.Cp(pair);
    for (int k = threadIdx.x; k < a.SC * a.RP; k += blockDim.x) { Cg[2 * k] = lc[k]; Cg[2 * k + 1] = lc[k + a.SC * a.RP + 1]; }
  }
}

// Red-black ordering of the same 2x2 block update (FOTG_SOR_REDBLACK; not reference-equivalent: ~0.07 px on alley_1).
// In the skewed arrays a pixel's colour (i + j) & 1 is the parity of its diagonal s = i + j: a half-sweep relaxes every cell of the
// even (odd) diagonals -- whole contiguous rows of C and D -- from values of the odd (even) ones, all cells independent.
//   left (i-1, j) = [s-1][r]   top (i, j-1) = [s-1][r-1]   right (i+1, j) = [s+1][r]   bottom (i, j+1) = [s+1][r+1]   (r = j)
// Neighbours outside the image are zero cells of D (and their weights psi in the live cell are 0): no border cases.
// One launch per half-sweep, the whole batch and every CU at once; D and C stay in L2.
__global__ __launch_bounds__(256) void vr_rb_halfsweep_kernel(VrArgs a, int col, float omega)
{
  const WgId wg = xcd_local_wg();
  const int pair = wg.y, RP = a.RP, RPD = a.RPD;
  const int k = wg.x * blockDim.x + threadIdx.x;
  const int s = col + 2 * (k / RP), r = k - (k / RP) * RP, i = s - r;
  if (s >= a.S || r >= a.h || i < 0 || i >= a.w) return;
  const float4 *C = a.Cp(pair);
  float2 *D = a.Dp(pair);
  const size_t c = ((size_t)s * RP + r) * 2, d = (size_t)s * RPD + r;
  const float hl = i > 0 ? C[c - 2 * (size_t)RP + 1].y : 0.f;     // psi_right of the left neighbour
  const float2 z = make_float2(0.f, 0.f);
  // (row 0 has no row above: the cell [s-1][-1] would belong to the previous diagonal's padding)
  const float2 nl = s > 0 ? D[d - RPD] : z, nt = (s > 0 && r > 0) ? D[d - RPD - 1] : z, nr = D[d + RPD], nb = D[d + RPD + 1];
  D[d] = sor_update(D[d], C[c], C[c + 1], hl, nl, nt, nr, nb, omega);
}

__global__ __launch_bounds__(256) void vr_finish_kernel(VrArgs a, float *__restrict__ flow, long flow_stride)
{
  const WgId wg = xcd_local_wg();
  const int idx = wg.x * blockDim.x + threadIdx.x;
  if (idx >= a.w * a.h) return;
  const int pair = wg.y, i = idx % a.w, j = idx / a.w, o = j * a.st + i;
  float *f = flow + (size_t)pair * flow_stride + 2 * (size_t)idx;
  const float2 d = a.Dp(pair)[a.didx(i, j)];
  f[0] = a.single(pair, P_WX)[o] + d.x;
  f[1] = a.single(pair, P_WY)[o] + d.y;
}

}  // namespace fotg
