// varref.hip.h -- variational refinement of one pyramid level
// (kroeger/refine_variational.cpp:25-241 driving FDF1.0.1/{opticalflow_aux,solver,image}.c).
//
// All planes use the FDF image_t layout (stride = ceil4(w), FDF1.0.1/image.c:15-31), one set per pair:
//   wx wy du dv mask s sh sv a11 a12 a22 b1 b2          (13 single planes)
//   avg Iz Ix Iy Ixx Ixy Iyy Ixz Iyz                    (9 x NOC planes, channel-planar like color_image_t)
// Kernels (per inner iteration: smooth -> data -> sor):
//   vr_prep     de-interleave flow, image_warp + mask (opticalflow_aux.c:18-60), 0.5*(I0+Iw), Iw-I0 (:80-83)
//   vr_deriv1/2 the seven 5-tap derivative images (opticalflow_aux.c:85-92, image.c:401-434,466-502)
//   vr_smooth   s = 1/4 alpha / sqrt(|grad uu|^2+|grad vv|^2+eps)  (opticalflow_aux.c:123-139), uu = wx+du on the fly
//   vr_data     sh/sv (:141-163), data term (:310-438), sub_laplacian (:172-199, gather form),
//               and the 2x2 block inverse of sor_coupled's first sweep (solver.c:115-120)
//   vr_sor_*    the sweeps of sor_coupled (solver.c:77-421)
//   vr_finish   flow = (wx+du, wy+dv) (refine_variational.cpp:208-221)
#pragma once
#include "common.h"

namespace fotg {

enum VrPlane { P_WX = 0, P_WY, P_DU, P_DV, P_MASK, P_S, P_SH, P_SV, P_A11, P_A12, P_A22, P_B1, P_B2, P_NSINGLE };
enum VrCPlane { C_AVG = 0, C_IZ, C_IX, C_IY, C_IXX, C_IXY, C_IYY, C_IXZ, C_IYZ, C_NCOLOR };

struct VrArgs {
  float *base;           // workspace of pair 0
  long pair_stride;      // floats between pairs
  long pl;               // floats per plane (st*h)
  int w, h, st, noc;
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

template <int NOC>
__global__ __launch_bounds__(256) void vr_prep_kernel(VrArgs a, const float *__restrict__ I0, const float *__restrict__ I1,
                                                      long img_stride, int tw, int pad,
                                                      const float *__restrict__ flow, long flow_stride)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.w * a.h) return;
  const int pair = blockIdx.y, i = idx % a.w, j = idx / a.w, o = j * a.st + i;
  const float *f = flow + (size_t)pair * flow_stride + 2 * (size_t)idx;
  const float wx = f[0], wy = f[1];
  a.single(pair, P_WX)[o] = wx;
  a.single(pair, P_WY)[o] = wy;
  a.single(pair, P_DU)[o] = 0.f;
  a.single(pair, P_DV)[o] = 0.f;
  // image_warp (opticalflow_aux.c:18-60)
  const float xx = i + wx, yy = j + wy;
  const int x = (int)floorf(xx), y = (int)floorf(yy);
  const float dx = xx - x, dy = yy - y;
  a.single(pair, P_MASK)[o] = (xx >= 0 && xx <= a.w - 1 && yy >= 0 && yy <= a.h - 1) ? 1.f : 0.f;
  const int x1 = clampi(x, a.w), x2 = clampi(x + 1, a.w), y1 = clampi(y, a.h), y2 = clampi(y + 1, a.h);
  const float *s1 = I1 + (size_t)pair * img_stride, *s0 = I0 + (size_t)pair * img_stride;
#pragma unroll
  for (int c = 0; c < NOC; ++c) {
#define SRC(yy_, xx_) s1[((size_t)((yy_) + pad) * tw + ((xx_) + pad)) * NOC + c]
    const float wv = SRC(y1, x1) * (1.0f - dx) * (1.0f - dy) + SRC(y1, x2) * dx * (1.0f - dy) +
                     SRC(y2, x1) * (1.0f - dx) * dy + SRC(y2, x2) * dx * dy;
#undef SRC
    const float i1 = s0[((size_t)(j + pad) * tw + (i + pad)) * NOC + c];
    a.color(pair, C_AVG, c)[o] = 0.5f * (wv + i1);        // get_derivatives :81
    a.color(pair, C_IZ, c)[o] = wv - i1;                  // :82
  }
}

template <int NOC>
__global__ __launch_bounds__(256) void vr_deriv1_kernel(VrArgs a)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.w * a.h) return;
  const int pair = blockIdx.y, i = idx % a.w, j = idx / a.w, o = j * a.st + i;
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
__global__ __launch_bounds__(256) void vr_deriv2_kernel(VrArgs a)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.w * a.h) return;
  const int pair = blockIdx.y, i = idx % a.w, j = idx / a.w, o = j * a.st + i;
#pragma unroll
  for (int c = 0; c < NOC; ++c) {
    const float *ix = a.color(pair, C_IX, c), *iy = a.color(pair, C_IY, c);
    a.color(pair, C_IXX, c)[o] = conv_h5(ix + (size_t)j * a.st, i, a.w);
    a.color(pair, C_IXY, c)[o] = conv_v5(ix + i, j, a.h, a.st);
    a.color(pair, C_IYY, c)[o] = conv_v5(iy + i, j, a.h, a.st);
  }
}

// compute_smoothness, first half (opticalflow_aux.c:126-139); 3-tap {-0.5,-0,0.5} (image.c:376-399,436-464)
__global__ __launch_bounds__(256) void vr_smooth_kernel(VrArgs a, float quarter_alpha)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.w * a.h) return;
  const int pair = blockIdx.y, i = idx % a.w, j = idx / a.w, st = a.st, w = a.w, h = a.h;
  const float *wx = a.single(pair, P_WX), *wy = a.single(pair, P_WY), *du = a.single(pair, P_DU), *dv = a.single(pair, P_DV);
  auto UU = [&](int jj, int ii) { const int q = jj * st + ii; return wx[q] + du[q]; };   // refine_variational.cpp:208-214
  auto VV = [&](int jj, int ii) { const int q = jj * st + ii; return wy[q] + dv[q]; };
  const float c0 = -0.5f, c1 = -0.0f, c2 = 0.5f;
  const int im = clampi(i - 1, w), ip = clampi(i + 1, w);
  const float ux = c0 * UU(j, im) + c1 * UU(j, i) + c2 * UU(j, ip);
  const float vx = c0 * VV(j, im) + c1 * VV(j, i) + c2 * VV(j, ip);
  float uy, vy;
  if (j == 0) { uy = (c0 + c1) * UU(0, i) + c2 * UU(1, i); vy = (c0 + c1) * VV(0, i) + c2 * VV(1, i); }
  else if (j == h - 1) { uy = c0 * UU(j - 1, i) + (c1 + c2) * UU(j, i); vy = c0 * VV(j - 1, i) + (c1 + c2) * VV(j, i); }
  else { uy = c0 * UU(j - 1, i) + c1 * UU(j, i) + c2 * UU(j + 1, i); vy = c0 * VV(j - 1, i) + c1 * VV(j, i) + c2 * VV(j + 1, i); }
  const float eps = 0.001f * 0.001f;
  a.single(pair, P_S)[j * st + i] = quarter_alpha / sqrtf(ux * ux + uy * uy + vx * vx + vy * vy + eps);
}

template <int NOC>
__global__ __launch_bounds__(256) void vr_data_kernel(VrArgs a, float half_delta_over3, float half_gamma_over3)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.w * a.h) return;
  const int pair = blockIdx.y, i = idx % a.w, j = idx / a.w, st = a.st, w = a.w, h = a.h, o = j * st + i;
  const float *s = a.single(pair, P_S);
  // compute_smoothness second half (:141-163): horiz(i) = s(i)+s(i+1) (0 in the last column), vert likewise
  const float s_o = s[o];
  const float hr = (i < w - 1) ? s_o + s[o + 1] : 0.0f;
  const float hl = (i > 0) ? s[o - 1] + s_o : 0.0f;
  const float vb = (j < h - 1) ? s_o + s[o + st] : 0.0f;
  const float vt = (j > 0) ? s[o - st] + s_o : 0.0f;
  a.single(pair, P_SH)[o] = hr;
  a.single(pair, P_SV)[o] = vb;

  // compute_data (:310-438)
  const float dnorm = 0.1f * 0.1f, epsc = 0.001f * 0.001f, epsg = 0.001f * 0.001f;
  const float u = a.single(pair, P_DU)[o], v = a.single(pair, P_DV)[o], m = a.single(pair, P_MASK)[o];
  float A11 = 0, A12 = 0, A22 = 0, B1 = 0, B2 = 0;
  if constexpr (NOC == 1) {
    const float Ix = a.color(pair, C_IX, 0)[o], Iy = a.color(pair, C_IY, 0)[o], Iz = a.color(pair, C_IZ, 0)[o];
    const float Ixx = a.color(pair, C_IXX, 0)[o], Ixy = a.color(pair, C_IXY, 0)[o], Iyy = a.color(pair, C_IYY, 0)[o];
    const float Ixz = a.color(pair, C_IXZ, 0)[o], Iyz = a.color(pair, C_IYZ, 0)[o];
    float tmp, tmp2, n1, n2;
    if (half_delta_over3) {
      tmp = Iz + Ix * u + Iy * v;
      n1 = Ix * Ix + Iy * Iy + dnorm;
      tmp = m * half_delta_over3 / sqrtf(3 * tmp * tmp / n1 + epsc);
      tmp /= n1;
      A11 += tmp * Ix * Ix;
      A12 += tmp * Ix * Iy;
      A22 += tmp * Iy * Iy;
      B1 -= tmp * Iz * Ix;
      B2 -= tmp * Iz * Iy;
    }
    n1 = Ixx * Ixx + Ixy * Ixy + dnorm;
    n2 = Iyy * Iyy + Ixy * Ixy + dnorm;
    tmp = Ixz + Ixx * u + Ixy * v;
    tmp2 = Iyz + Ixy * u + Iyy * v;
    tmp = m * half_gamma_over3 / sqrtf(3 * tmp * tmp / n1 + 3 * tmp2 * tmp2 / n2 + epsg);
    tmp2 = tmp / n2; tmp /= n1;
    A11 += tmp * Ixx * Ixx + tmp2 * Ixy * Ixy;
    A12 += tmp * Ixx * Ixy + tmp2 * Ixy * Iyy;
    A22 += tmp2 * Iyy * Iyy + tmp * Ixy * Ixy;
    B1 -= tmp * Ixx * Ixz + tmp2 * Ixy * Iyz;
    B2 -= tmp2 * Iyy * Iyz + tmp * Ixy * Ixz;
    A11 *= 3; A12 *= 3; A22 *= 3; B1 *= 3; B2 *= 3;       // :420-426
  } else {
    float ix[3], iy[3], iz[3], ixx[3], ixy[3], iyy[3], ixz[3], iyz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ix[c] = a.color(pair, C_IX, c)[o]; iy[c] = a.color(pair, C_IY, c)[o]; iz[c] = a.color(pair, C_IZ, c)[o];
      ixx[c] = a.color(pair, C_IXX, c)[o]; ixy[c] = a.color(pair, C_IXY, c)[o]; iyy[c] = a.color(pair, C_IYY, c)[o];
      ixz[c] = a.color(pair, C_IXZ, c)[o]; iyz[c] = a.color(pair, C_IYZ, c)[o];
    }
    if (half_delta_over3) {
      float t[3], n[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { t[c] = iz[c] + ix[c] * u + iy[c] * v; n[c] = ix[c] * ix[c] + iy[c] * iy[c] + dnorm; }
      float tmp = m * half_delta_over3 / sqrtf(t[0] * t[0] / n[0] + t[1] * t[1] / n[1] + t[2] * t[2] / n[2] + epsc);
      const float k2 = tmp / n[2], k1 = tmp / n[1], k0 = tmp / n[0];
      const float k[3] = {k0, k1, k2};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        A11 += k[c] * ix[c] * ix[c]; A12 += k[c] * ix[c] * iy[c]; A22 += k[c] * iy[c] * iy[c];
        B1 -= k[c] * iz[c] * ix[c];  B2 -= k[c] * iz[c] * iy[c];
      }
    }
    float n1[3], n2[3], t1[3], t2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      n1[c] = ixx[c] * ixx[c] + ixy[c] * ixy[c] + dnorm; n2[c] = iyy[c] * iyy[c] + ixy[c] * ixy[c] + dnorm;
      t1[c] = ixz[c] + ixx[c] * u + ixy[c] * v;           t2[c] = iyz[c] + ixy[c] * u + iyy[c] * v;
    }
    const float tmp = m * half_gamma_over3 / sqrtf(t1[0] * t1[0] / n1[0] + t2[0] * t2[0] / n2[0] + t1[1] * t1[1] / n1[1] +
                                                   t2[1] * t2[1] / n2[1] + t1[2] * t1[2] / n1[2] + t2[2] * t2[2] / n2[2] + epsg);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float ka = tmp / n1[c], kb = tmp / n2[c];
      A11 += ka * ixx[c] * ixx[c] + kb * ixy[c] * ixy[c];
      A12 += ka * ixx[c] * ixy[c] + kb * ixy[c] * iyy[c];
      A22 += kb * iyy[c] * iyy[c] + ka * ixy[c] * ixy[c];
      B1 -= ka * ixx[c] * ixz[c] + kb * ixy[c] * iyz[c];
      B2 -= kb * iyy[c] * iyz[c] + ka * ixy[c] * ixz[c];
    }
  }

  // sub_laplacian (:172-199) for b1 (src wx) and b2 (src wy): -left, +right, -top, +bottom
  const float *wx = a.single(pair, P_WX), *wy = a.single(pair, P_WY);
  if (i > 0)     { B1 -= hl * (wx[o] - wx[o - 1]);  B2 -= hl * (wy[o] - wy[o - 1]); }
  if (i < w - 1) { B1 += hr * (wx[o + 1] - wx[o]);  B2 += hr * (wy[o + 1] - wy[o]); }
  if (j > 0)     { B1 -= vt * (wx[o] - wx[o - st]); B2 -= vt * (wy[o] - wy[o - st]); }
  if (j < h - 1) { B1 += vb * (wx[o + st] - wx[o]); B2 += vb * (wy[o + st] - wy[o]); }

  // first sweep of sor_coupled inverts the 2x2 block (solver.c:115-120): dpsis = hl+hr(+vt)(+vb)
  float dps = hl + hr;
  if (j > 0) dps = dps + vt;
  if (j < h - 1) dps = dps + vb;
  const float M11 = A22 + dps, M22 = A11 + dps;
  const float det = M11 * M22 - A12 * A12;
  a.single(pair, P_A11)[o] = M11 / det;
  a.single(pair, P_A22)[o] = M22 / det;
  a.single(pair, P_A12)[o] = A12 / -det;
  a.single(pair, P_B1)[o] = B1;
  a.single(pair, P_B2)[o] = B2;
}

// one pixel update of sor_coupled (solver.c:122-130 etc.); du_l/du_t are the NEW left/top values
__device__ __forceinline__ void sor_update(float &du, float &dv, float a11, float a12, float a22, float b1, float b2,
                                           float hl, float hr, float vt, float vb, float du_l, float dv_l, float du_t,
                                           float dv_t, float du_r, float dv_r, float du_b, float dv_b, bool has_l,
                                           bool has_t, bool has_b, float omega)
{
  float s1 = hr * du_r, s2 = hr * dv_r;
  if (has_t) { s1 = s1 + vt * du_t; s2 = s2 + vt * dv_t; }
  if (has_b) { s1 = s1 + vb * du_b; s2 = s2 + vb * dv_b; }
  s1 = s1 + b1; s2 = s2 + b2;
  float B1 = s1, B2 = s2;
  if (has_l) { B1 = hl * du_l + s1; B2 = hl * dv_l + s2; }
  du += omega * (a11 * B1 + a12 * B2 - du);
  dv += omega * (a12 * B1 + a22 * B2 - dv);
}

// Lexicographic sweeps as an anti-diagonal wavefront: pixel (i,j) runs at step i+j, after its NEW left
// (i-1,j) and top (i,j-1) neighbours (step i+j-1) and before its OLD right/bottom neighbours (step
// i+j+1) -- exactly the dependency order of the row-major loop of solver.c, hence bit-identical.
// One workgroup per pair, thread r owns row r; the top neighbour's fresh value travels through LDS.
__global__ __launch_bounds__(1024) void vr_sor_wavefront_kernel(VrArgs a, int iterations, float omega)
{
  __shared__ float xdu[2][1024], xdv[2][1024];
  const int pair = blockIdx.x, r = threadIdx.x, w = a.w, h = a.h, st = a.st;
  float *du = a.single(pair, P_DU), *dv = a.single(pair, P_DV);
  const float *a11 = a.single(pair, P_A11), *a12 = a.single(pair, P_A12), *a22 = a.single(pair, P_A22);
  const float *b1 = a.single(pair, P_B1), *b2 = a.single(pair, P_B2), *sh = a.single(pair, P_SH), *sv = a.single(pair, P_SV);
  const bool row = r < h;
  for (int it = 0; it < iterations; ++it) {
    float du_l = 0.f, dv_l = 0.f;
    for (int s = 0; s < w + h - 1; ++s) {
      const int i = s - r;
      float ndu = 0.f, ndv = 0.f;
      if (row && i >= 0 && i < w) {
        const int o = r * st + i;
        float cu = du[o], cv = dv[o];
        const bool has_r = i < w - 1, has_t = r > 0, has_b = r < h - 1;
        const float du_r = has_r ? du[o + 1] : 0.f, dv_r = has_r ? dv[o + 1] : 0.f;
        const float du_b = has_b ? du[o + st] : 0.f, dv_b = has_b ? dv[o + st] : 0.f;
        const float du_t = has_t ? xdu[(s + 1) & 1][r - 1] : 0.f, dv_t = has_t ? xdv[(s + 1) & 1][r - 1] : 0.f;
        const float hl = i > 0 ? sh[o - 1] : 0.f, vt = has_t ? sv[o - st] : 0.f;
        sor_update(cu, cv, a11[o], a12[o], a22[o], b1[o], b2[o], hl, sh[o], vt, sv[o], du_l, dv_l, du_t, dv_t,
                   du_r, dv_r, du_b, dv_b, i > 0, has_t, has_b, omega);
        du[o] = cu; dv[o] = cv;
        du_l = cu; dv_l = cv;
        ndu = cu; ndv = cv;
      }
      if (r < 1024) { xdu[s & 1][r] = ndu; xdv[s & 1][r] = ndv; }
      __syncthreads();
    }
  }
}

// red-black ordering of the same block update (throughput mode; deviates from the reference by ~0.07 px)
__global__ __launch_bounds__(1024) void vr_sor_redblack_kernel(VrArgs a, int iterations, float omega)
{
  const int pair = blockIdx.x, w = a.w, h = a.h, st = a.st;
  float *du = a.single(pair, P_DU), *dv = a.single(pair, P_DV);
  const float *a11 = a.single(pair, P_A11), *a12 = a.single(pair, P_A12), *a22 = a.single(pair, P_A22);
  const float *b1 = a.single(pair, P_B1), *b2 = a.single(pair, P_B2), *sh = a.single(pair, P_SH), *sv = a.single(pair, P_SV);
  for (int it = 0; it < iterations; ++it)
    for (int col = 0; col < 2; ++col) {
      for (int idx = threadIdx.x; idx < w * h; idx += blockDim.x) {
        const int i = idx % w, j = idx / w;
        if (((i + j) & 1) != col) continue;
        const int o = j * st + i;
        float cu = du[o], cv = dv[o];
        const bool has_l = i > 0, has_r = i < w - 1, has_t = j > 0, has_b = j < h - 1;
        sor_update(cu, cv, a11[o], a12[o], a22[o], b1[o], b2[o], has_l ? sh[o - 1] : 0.f, sh[o], has_t ? sv[o - st] : 0.f, sv[o],
                   has_l ? du[o - 1] : 0.f, has_l ? dv[o - 1] : 0.f, has_t ? du[o - st] : 0.f, has_t ? dv[o - st] : 0.f,
                   has_r ? du[o + 1] : 0.f, has_r ? dv[o + 1] : 0.f, has_b ? du[o + st] : 0.f, has_b ? dv[o + st] : 0.f,
                   has_l, has_t, has_b, omega);
        du[o] = cu; dv[o] = cv;
      }
      __syncthreads();
    }
}

__global__ __launch_bounds__(256) void vr_finish_kernel(VrArgs a, float *__restrict__ flow, long flow_stride)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.w * a.h) return;
  const int pair = blockIdx.y, i = idx % a.w, j = idx / a.w, o = j * a.st + i;
  float *f = flow + (size_t)pair * flow_stride + 2 * (size_t)idx;
  f[0] = a.single(pair, P_WX)[o] + a.single(pair, P_DU)[o];
  f[1] = a.single(pair, P_WY)[o] + a.single(pair, P_DV)[o];
}

}  // namespace fotg
