// upsample.hip.h -- final flow to full resolution (kroeger/run_dense.cpp:407-414 == src/run_dense.cpp:293-303):
// flow *= 2^sc_l; cv::resize(INTER_LINEAR) by 2^sc_l; crop the divisibility padding.
// cv::resize semantics: source coordinate (d+0.5)/s-0.5 evaluated in double, floor, weights clamped
// to 0 at the borders; horizontal pass then vertical pass.
#pragma once
#include "common.h"

namespace fotg {

__global__ __launch_bounds__(256) void upsample_crop_kernel(const float *__restrict__ flow, long in_stride, int wl, int hl,
                                                            int sc_l, int x0, int y0, int w_org, int h_org,
                                                            float *__restrict__ out, long out_stride, int nch)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= w_org * h_org) return;
  const int pair = blockIdx.y, x = idx % w_org, y = idx / w_org;
  const float *f = flow + (size_t)pair * in_stride;
  const float scf = (float)(1 << sc_l);
  const double scale = 1.0 / (double)(1 << sc_l);
  float fy = (float)((y + y0 + 0.5) * scale - 0.5);
  int sy = (int)floorf(fy); fy -= sy;
  if (sy < 0) { fy = 0; sy = 0; }
  if (sy >= hl - 1) { fy = 0; sy = hl - 1; }
  const int sy1 = sy + 1 < hl ? sy + 1 : hl - 1;
  float fx = (float)((x + x0 + 0.5) * scale - 0.5);
  int sx = (int)floorf(fx); fx -= sx;
  if (sx < 0) { fx = 0; sx = 0; }
  if (sx >= wl - 1) { fx = 0; sx = wl - 1; }
  const int sx1 = sx + 1 < wl ? sx + 1 : wl - 1;
  float *o = out + (size_t)pair * out_stride + nch * (size_t)idx;      // nch = 1: stereo depth (run_dense.cpp:387-388)
  for (int c = 0; c < nch; ++c) {
    float v00 = f[nch * (sy * wl + sx) + c], v01 = f[nch * (sy * wl + sx1) + c];
    float v10 = f[nch * (sy1 * wl + sx) + c], v11 = f[nch * (sy1 * wl + sx1) + c];
    if (sc_l != 0) { v00 *= scf; v01 *= scf; v10 *= scf; v11 *= scf; }
    const float r0 = v00 * (1.f - fx) + v01 * fx;
    const float r1 = v10 * (1.f - fx) + v11 * fx;
    o[c] = r0 * (1.f - fy) + r1 * fy;
  }
}

// The same, four consecutive output pixels per thread on a (columns, row, pair) grid, the two coarse rows a workgroup's pixels
// interpolate between staged in LDS (the per-pixel gathers otherwise go through the texture path: 16 scattered dword loads per
// thread, 0.26 of the HBM roof), 16-byte nontemporal stores: the kernel is bound by its HBM writes (2 W H 4 bytes per pair).
// Same values as upsample_crop_kernel: the source coordinate (d + 0.5) / s - 0.5 with s = 2^k is N / 2^(k+1) for the integer
// N = 2 d + 1 - 2^k -- exact in double AND in float (|N| < 2^24), so the double expression of the reference rounds to exactly
// (float)N * 2^-(k+1); the three lerps are evaluated in the same order.
template <int NCH>
__global__ __launch_bounds__(256) void upsample_crop4_kernel(const float *__restrict__ flow, long in_stride, int wl, int hl,
                                                             int sc_l, int x0, int y0, int w_org, int h_org,
                                                             float *__restrict__ out, long out_stride)
{
  constexpr int MAXC = 1024 + 4;                                 // source columns a workgroup can touch at scale 1
  __shared__ float rows[2][MAXC * NCH];
  const int xw = blockIdx.x * blockDim.x * 4, xq = xw + threadIdx.x * 4, y = blockIdx.y, pair = blockIdx.z;
  const float *f = flow + (size_t)pair * in_stride;
  const float scf = (float)(1 << sc_l);
  const float inv = __builtin_ldexpf(1.0f, -(sc_l + 1));
  auto coord = [&](int d, int n, int &s0, int &s1, float &fr) {   // d: padded destination coordinate; n: source extent
    const int N = 2 * d + 1 - (1 << sc_l);
    float fc = (float)N * inv;
    int si = (int)floorf(fc); fc -= si;
    if (si < 0) { fc = 0; si = 0; }
    if (si >= n - 1) { fc = 0; si = n - 1; }
    s0 = si; s1 = si + 1 < n ? si + 1 : n - 1; fr = fc;
  };
  int sy, sy1; float fy;
  coord(y + y0, hl, sy, sy1, fy);
  // source columns of the workgroup's pixels: [c0, c0 + ncol)
  int c0, c0b; float dummy;
  coord(xw + x0, wl, c0, c0b, dummy);
  const int xlast = (xw + (int)blockDim.x * 4 - 1 < w_org ? xw + (int)blockDim.x * 4 - 1 : w_org - 1);
  int cl, cl1;
  coord(xlast + x0, wl, cl, cl1, dummy);
  const int ncol = cl1 - c0 + 1;                                  // <= blockDim * 4 / scale + 2
  for (int t = threadIdx.x; t < ncol * NCH; t += blockDim.x) {
    rows[0][t] = f[(size_t)NCH * (sy * wl + c0) + t];
    rows[1][t] = f[(size_t)NCH * (sy1 * wl + c0) + t];
  }
  __syncthreads();
  if (xq >= w_org) return;
  float res[4 * NCH];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = xq + k < w_org ? xq + k : w_org - 1;
    int sx, sx1; float fx;
    coord(x + x0, wl, sx, sx1, fx);
    sx -= c0; sx1 -= c0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // (flow *= 2^sc_l per tap, run_dense.cpp:407; scaling the staged rows once instead measured SLOWER: 0.43 against 0.27 ms)
      float v00 = rows[0][NCH * sx + c], v01 = rows[0][NCH * sx1 + c], v10 = rows[1][NCH * sx + c], v11 = rows[1][NCH * sx1 + c];
      if (sc_l != 0) { v00 *= scf; v01 *= scf; v10 *= scf; v11 *= scf; }
      const float a0 = v00 * (1.f - fx) + v01 * fx;
      const float a1 = v10 * (1.f - fx) + v11 * fx;
      res[k * NCH + c] = a0 * (1.f - fy) + a1 * fy;
    }
  }
  float *o = out + (size_t)pair * out_stride + NCH * ((size_t)y * w_org + xq);
  typedef float vf4 __attribute__((ext_vector_type(4)));
  if (xq + 4 <= w_org && ((((size_t)o) & 15) == 0)) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const vf4 v = {res[4 * q], res[4 * q + 1], res[4 * q + 2], res[4 * q + 3]};
      __builtin_nontemporal_store(v, reinterpret_cast<vf4 *>(o) + q);
    }
  } else {
    for (int k = 0; k < 4 && xq + k < w_org; ++k)
#pragma unroll
      for (int c = 0; c < NCH; ++c) o[k * NCH + c] = res[k * NCH + c];
  }
}

// Gradient-magnitude input (kroeger/run_dense.cpp:138-147, the reference's SELECTCHANNEL==2 build): level 0 of the pyramid is
// sqrt(dx^2 + dy^2) of the PADDED frame, dx = P(x+1) - P(x-1), dy = P(y+1) - P(y-1) per channel (cv::Sobel ksize 1) with
// REFLECT_101 at the padded frame's edge.  P is the replicate-padded frame (run_dense.cpp:306-310), read through clamped
// coordinates from the original one; out: n x Hp x Wp x noc.  One thread per value; a pre-pass outside the hot path.
template <typename T>
__global__ __launch_bounds__(256) void gradmag_kernel(const T *__restrict__ frames, long frame_stride, int w_org, int h_org, int noc,
                                                      int left, int top, int Wp, int Hp, float *__restrict__ out)
{
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Wp * Hp * noc) return;
  const int c = (int)(idx % noc), x = (int)((idx / noc) % Wp), y = (int)(idx / ((long)noc * Wp));
  const T *f = frames + (size_t)blockIdx.y * frame_stride;
  auto P = [&](int px, int py) { return (float)f[((size_t)clampi(py - top, h_org) * w_org + clampi(px - left, w_org)) * noc + c]; };
  const int xm = x > 0 ? x - 1 : 1, xp = x < Wp - 1 ? x + 1 : Wp - 2;
  const int ym = y > 0 ? y - 1 : 1, yp = y < Hp - 1 ? y + 1 : Hp - 2;
  const float dx = P(xp, y) - P(xm, y), dy = P(x, yp) - P(x, ym);
  const float dx2 = dx * dx, dy2 = dy * dy;
  out[(size_t)blockIdx.y * ((size_t)Wp * Hp * noc) + idx] = sqrtf(dx2 + dy2);
}

}  // namespace fotg
