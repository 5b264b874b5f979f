// densify.hip.h -- patch -> dense flow aggregation (kroeger/patchgrid.cpp:213-275, 377-397).
//
// The reference scatters every patch into the flow image serially in patch-id order.  Here each
// pixel GATHERS its covering patches (at most (ps/steps)^2: 4 at op-pt 2, 16 at op-pt 3/4) in the
// same patch-id order (x-major: id = xi*noph + yi), so the float sums are bit-identical to the
// serial reference and no atomics are needed (deterministic, unlike src/kernels/densify.cu:81-84).
#pragma once
#include "common.h"

namespace fotg {

// weight 1 / max(minerr, |error|) of window pixel (wx, wy) of a patch (RGB: 1 / sum over channels).  The reference walks
// the weight vector with a pointer that advances by 3 for a window pixel that passes the inside test but only by 1 for a
// skipped one (patchgrid.cpp:253-258, 332-337): index = 3*(#inside before) + (#outside before), where the inside
// pixels of the window are the rectangle [vx0,vx1] x [vy0,vy1].
template <int PS, int NOC>
__device__ __forceinline__ float patch_absw(const float *__restrict__ pw, int wx, int wy, int vx0, int vx1, int vy0, int vy1)
{
  const float minerr = 2.0f;                                   // kroeger/oflow.h:62
  if constexpr (NOC == 1) {
    const float v = pw[wy * PS + wx];
    return 1.0f / (v > minerr ? v : minerr);
  } else {
    const int ncols = vx1 - vx0 + 1;
    const int rows_before = wy - vy0;                          // (wx, wy) is inside, so >= 0
    const int inside_before = rows_before * ncols + (wx - vx0);
    const int n = wy * PS + wx;
    const int k = 3 * inside_before + (n - inside_before);
    float s = (pw[k] > minerr ? pw[k] : minerr);
    s += (pw[k + 1] > minerr ? pw[k + 1] : minerr);
    s += (pw[k + 2] > minerr ? pw[k + 2] : minerr);
    return 1.0f / s;
  }
}

// own patches of pixel (xt, yt), accumulated in patch-id order
template <int PS, int NOC>
__device__ __forceinline__ void gather_own(const float *__restrict__ p_iter, const float *__restrict__ pweight, int pair, int xt, int yt,
                                           const LevelGeom &g, float &we, float &f0, float &f1)
{
  constexpr int NV = PS * PS * NOC;
  constexpr int LB = -PS / 2, UB = PS / 2 - 1;
  // patches (xi, yi) with  ref + LB <= t <= ref + UB,  ref = i*steps + off
  // -> i >= (t - UB - off)/steps, i <= (t - LB - off)/steps
  auto lo = [&](int t, int off) { int v = t - UB - off; return v <= 0 ? 0 : (v + g.steps - 1) / g.steps; };
  auto hi = [&](int t, int off, int n) { int v = t - LB - off; if (v < 0) return -1; int q = v / g.steps; return q > n - 1 ? n - 1 : q; };
  const int xlo = lo(xt, g.offw), xhi = hi(xt, g.offw, g.nopw);
  const int ylo = lo(yt, g.offh), yhi = hi(yt, g.offh, g.noph);
  for (int xi = xlo; xi <= xhi; ++xi) {
    for (int yi = ylo; yi <= yhi; ++yi) {
      const int ip = xi * g.noph + yi;
      const size_t pb = (size_t)pair * g.nop + ip;
      const int refx = xi * g.steps + g.offw, refy = yi * g.steps + g.offh;
      const int wx = xt - refx - LB, wy = yt - refy - LB;       // window coordinates 0..PS-1
      const int vx0 = refx + LB < 0 ? -(refx + LB) : 0;         // inside test xt >= 0 && xt < w (patchgrid.cpp:247)
      const int vx1 = refx + UB > g.w - 1 ? PS - 1 - (refx + UB - (g.w - 1)) : PS - 1;
      const int vy0 = refy + LB < 0 ? -(refy + LB) : 0;
      const int vy1 = refy + UB > g.h - 1 ? PS - 1 - (refy + UB - (g.h - 1)) : PS - 1;
      const float absw = patch_absw<PS, NOC>(pweight + pb * NV, wx, wy, vx0, vx1, vy0, vy1);
      const float u = p_iter[pb * 2], v = p_iter[pb * 2 + 1];
      we += absw;
      f0 += u * absw;
      f1 += v * absw;
    }
  }
}

template <int PS, int NOC>
__global__ __launch_bounds__(256) void densify_kernel(const float *__restrict__ p_iter, const float *__restrict__ pweight,
                                                      float *__restrict__ flowout, long flow_stride, LevelGeom g, int nch, int nwg)
{
  // batches: all of a pair on one XCD; otherwise (nwg > 0: the launch has rounded gridDim.x up to a multiple of 8) every XCD a
  // contiguous band of rows, so that a patch is fetched by one L2 (two at a band boundary) instead of by all eight
  WgId wg = xcd_local_wg();
  if (nwg > 0) { wg.x = xcd_banded_x(nwg); if (wg.x < 0) return; }
  const int idx = wg.x * blockDim.x + threadIdx.x;
  if (idx >= g.w * g.h) return;
  const int pair = wg.y;
  float we = 0.f, f0 = 0.f, f1 = 0.f;
  gather_own<PS, NOC>(p_iter, pweight, pair, idx % g.w, idx / g.w, g, we, f0, f1);
  if (we > 0) { f0 /= we; f1 /= we; }
  // nch = 1: stereo depth mode, one displacement channel (patchgrid.cpp:268, :386-390); the patches' second slot is 0
  float *out = flowout + (size_t)pair * flow_stride + nch * (size_t)idx;
  out[0] = f0;
  if (nch == 2) out[1] = f1;
}

// Forward-backward merge (usefbcon, patchgrid.cpp:278-375): after its own patches every pixel also receives the patches
// of the complementary grid, splatted at their position AFTER optimisation with bilinear weights and reversed flow.
// Those positions are data dependent, so the covering patches cannot be enumerated from the pixel.  One workgroup owns a
// 16x16 pixel tile: it scans all complementary patches in id order, 256 at a time, compacts (in order) the ones whose
// footprint touches the tile into an LDS list, and every pixel then walks the list -- the serial reference order
// (patch id; inside a patch: window row, window column; taps cc, fc, cf, ff) is kept, so the sums are bit-identical.
__device__ __forceinline__ int clamp_pos(int v) { return v < -(1 << 24) ? -(1 << 24) : (v > (1 << 24) ? (1 << 24) : v); }

template <int PS, int NOC>
__global__ __launch_bounds__(256) void densify_fb_kernel(const float *__restrict__ p_iter, const float *__restrict__ pweight,
                                                         const float *__restrict__ cg_p_iter, const float *__restrict__ cg_pweight,
                                                         float *__restrict__ flowout, long flow_stride, LevelGeom g, int nch)
{
  constexpr int NV = PS * PS * NOC;
  constexpr int LB = -PS / 2, UB = PS / 2 - 1;
  __shared__ int list[256];
  __shared__ int wave_cnt[4];
  const WgId wg = xcd_local_wg();
  const int pair = wg.y;
  const int tiles_x = (g.w + 15) >> 4;
  const int tx0 = (wg.x % tiles_x) << 4, ty0 = (wg.x / tiles_x) << 4;
  const int xt = tx0 + (threadIdx.x & 15), yt = ty0 + (threadIdx.x >> 4);
  const bool live = xt < g.w && yt < g.h;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float we = 0.f, f0 = 0.f, f1 = 0.f;
  if (live) gather_own<PS, NOC>(p_iter, pweight, pair, xt, yt, g, we, f0, f1);
  for (int base = 0; base < g.nop; base += 256) {
    // ---- does patch base + tid touch the tile?  targets: x in [pos0 + LB - 1, pos0 + UB], same for y ----
    const int ip = base + threadIdx.x;
    bool hit = false;
    if (ip < g.nop) {
      const size_t pb = (size_t)pair * g.nop + ip;
      const int xi = ip / g.noph, yi = ip - xi * g.noph;
      const float rx = (float)(xi * g.steps + g.offw) + cg_p_iter[pb * 2], ry = (float)(yi * g.steps + g.offh) + cg_p_iter[pb * 2 + 1];
      // (positions of a diverged backward flow saturate the conversion: keep them where the sums below cannot overflow)
      const int pos0 = clamp_pos((int)ceil((double)rx + .00001)), pos1 = clamp_pos((int)ceil((double)ry + .00001));
      hit = pos0 + UB >= tx0 && pos0 + LB - 1 <= tx0 + 15 && pos1 + UB >= ty0 && pos1 + LB - 1 <= ty0 + 15;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (k < wv) off += wave_cnt[k]; total += wave_cnt[k]; }
    if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = ip;
    __syncthreads();
    if (live) {
      for (int q = 0; q < total; ++q) {
        const int jp = list[q];
        const size_t pb = (size_t)pair * g.nop + jp;
        const int xi = jp / g.noph, yi = jp - xi * g.noph;
        const float u = cg_p_iter[pb * 2], v = cg_p_iter[pb * 2 + 1];
        const float rx = (float)(xi * g.steps + g.offw) + u, ry = (float)(yi * g.steps + g.offh) + v;   // pt_iter (patch.cpp:214-221)
        const int pos0 = clamp_pos((int)ceil((double)rx + .00001)), pos1 = clamp_pos((int)ceil((double)ry + .00001));           // :302-305
        const int pos2 = (int)floorf(rx), pos3 = (int)floorf(ry);
        const float r0 = rx - pos2, r1 = ry - pos3;
        const float wb[4] = {r0 * r1, (1 - r0) * r1, r0 * (1 - r1), (1 - r0) * (1 - r1)};
        // inside rectangle of the window: 1 <= x + pos0 <= w - 2  (patchgrid.cpp:324)
        const int vx0 = 1 - pos0 - LB > 0 ? 1 - pos0 - LB : 0, vx1 = g.w - 2 - pos0 - LB < PS - 1 ? g.w - 2 - pos0 - LB : PS - 1;
        const int vy0 = 1 - pos1 - LB > 0 ? 1 - pos1 - LB : 0, vy1 = g.h - 2 - pos1 - LB < PS - 1 ? g.h - 2 - pos1 - LB : PS - 1;
        // window pixels whose taps land on (xt, yt), in serial order: (xt,yt) cc, (xt+1,yt) fc, (xt,yt+1) cf, (xt+1,yt+1) ff
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int wx = xt + (t & 1) - pos0 - LB, wy = yt + (t >> 1) - pos1 - LB;     // window coordinates of the source pixel
          if (wx < vx0 || wx > vx1 || wy < vy0 || wy > vy1) continue;
          const float absw = patch_absw<PS, NOC>(cg_pweight + pb * NV, wx, wy, vx0, vx1, vy0, vy1);
          const float n0 = u * absw, n1 = v * absw;
          we += wb[t] * absw;
          f0 -= wb[t] * n0;
          f1 -= wb[t] * n1;
        }
      }
    }
    __syncthreads();
  }
  if (!live) return;
  if (we > 0) { f0 /= we; f1 /= we; }
  float *out = flowout + (size_t)pair * flow_stride + nch * ((size_t)yt * g.w + xt);
  out[0] = f0;
  if (nch == 2) out[1] = f1;
}

}  // namespace fotg
