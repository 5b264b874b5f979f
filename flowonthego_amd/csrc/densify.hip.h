// densify.hip.h -- patch -> dense flow aggregation (kroeger/patchgrid.cpp:213-275, 377-397).
//
// The reference scatters every patch into the flow image serially in patch-id order.  Here each
// pixel GATHERS its covering patches (at most (ps/steps)^2: 4 at op-pt 2, 16 at op-pt 3/4) in the
// same patch-id order (x-major: id = xi*noph + yi), so the float sums are bit-identical to the
// serial reference and no atomics are needed (deterministic, unlike src/kernels/densify.cu:81-84).
#pragma once
#include "common.h"

namespace fotg {

template <int PS, int NOC>
__global__ __launch_bounds__(256) void densify_kernel(const float *__restrict__ p_iter, const float *__restrict__ pweight,
                                                      float *__restrict__ flowout, long flow_stride, LevelGeom g)
{
  constexpr int NV = PS * PS * NOC;
  constexpr int LB = -PS / 2, UB = PS / 2 - 1;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= g.w * g.h) return;
  const int pair = blockIdx.y;
  const int xt = idx % g.w, yt = idx / g.w;
  const float minerr = 2.0f;                                   // kroeger/oflow.h:62
  // patches (xi, yi) with  ref + LB <= t <= ref + UB,  ref = i*steps + off
  // -> i >= (t - UB - off)/steps, i <= (t - LB - off)/steps
  auto lo = [&](int t, int off) { int v = t - UB - off; return v <= 0 ? 0 : (v + g.steps - 1) / g.steps; };
  auto hi = [&](int t, int off, int n) { int v = t - LB - off; if (v < 0) return -1; int q = v / g.steps; return q > n - 1 ? n - 1 : q; };
  const int xlo = lo(xt, g.offw), xhi = hi(xt, g.offw, g.nopw);
  const int ylo = lo(yt, g.offh), yhi = hi(yt, g.offh, g.noph);
  float we = 0.f, f0 = 0.f, f1 = 0.f;
  for (int xi = xlo; xi <= xhi; ++xi) {
    for (int yi = ylo; yi <= yhi; ++yi) {
      const int ip = xi * g.noph + yi;
      const size_t pb = (size_t)pair * g.nop + ip;
      const int refx = xi * g.steps + g.offw, refy = yi * g.steps + g.offh;
      const int wx = xt - refx - LB, wy = yt - refy - LB;       // window coordinates 0..PS-1
      const float *pw = pweight + pb * NV;
      float absw;
      if constexpr (NOC == 1) {
        const float v = pw[wy * PS + wx];
        absw = 1.0f / (v > minerr ? v : minerr);
      } else {
        // patchgrid.cpp:253-258 advances the weight pointer by 3 for a pixel inside the image but only by 1
        // for a skipped one; reproduce its index: 3*(#inside before) + (#outside before)
        const int vx0 = refx + LB < 0 ? -(refx + LB) : 0;                         // first inside column of the window
        const int vx1 = refx + UB > g.w - 1 ? PS - 1 - (refx + UB - (g.w - 1)) : PS - 1;
        const int vy0 = refy + LB < 0 ? -(refy + LB) : 0;
        const int vy1 = refy + UB > g.h - 1 ? PS - 1 - (refy + UB - (g.h - 1)) : PS - 1;
        const int ncols = vx1 - vx0 + 1;
        const int rows_before = wy - vy0;                                          // wy is inside, so >= 0
        const int inside_before = rows_before * ncols + (wx - vx0);
        const int n = wy * PS + wx;
        const int k = 3 * inside_before + (n - inside_before);
        float s = (pw[k] > minerr ? pw[k] : minerr);
        s += (pw[k + 1] > minerr ? pw[k + 1] : minerr);
        s += (pw[k + 2] > minerr ? pw[k + 2] : minerr);
        absw = 1.0f / s;
      }
      const float u = p_iter[pb * 2], v = p_iter[pb * 2 + 1];
      we += absw;
      f0 += u * absw;
      f1 += v * absw;
    }
  }
  if (we > 0) { f0 /= we; f1 /= we; }
  float *out = flowout + (size_t)pair * flow_stride + 2 * (size_t)idx;
  out[0] = f0;
  out[1] = f1;
}

}  // namespace fotg
