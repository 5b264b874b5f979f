// pyramid.hip.h -- image pyramid + gradients (kroeger/run_dense.cpp:130-178, :298-311).
//
// The only HBM-heavy stage of the path: each input frame is read exactly once.
//   pyr_base_kernel   frame (h_org x w_org x NOC, unpadded) -> level LV interior, cascading the 2x2 means
//                     in registers / across lanes (one wave = 256 px x 2^LV rows, 16-B loads per lane).
//                     The replicate padding to multiples of 2^sc_f (run_dense.cpp:298-311) is a clamp
//                     on the source coordinate -- no padded copy of the frame is ever materialised.
//   pyr_halve_kernel  level l -> level l+1 (small images)
//   pyr_border_grad_kernel  replicate border of the image, [-1 0 1] gradients with REFLECT_101 inside,
//                     zero border for the gradients (run_dense.cpp:156-175)
// 2x2 mean order: ((a+c)+(b+d))*0.25, a,b = upper row, c,d = lower row (oracle definition D4).
#pragma once
#include "common.h"

namespace fotg {

// SRCC != NOC (8-bit frames only, SRCC = 3, NOC = 1): the frames are 3-channel 8-bit colour and the flow is computed on their gray
// value -- what cv::imread(.., IMREAD_GRAYSCALE) hands kroeger/run_dense.cpp:199-209 for a colour file: OpenCV's fixed-point
// BGR2GRAY, (1868 B + 9617 G + 4899 R + 8192) >> 14, here on load (integer arithmetic: exact by construction).  coef0 / coef2 are
// the weights of the first and third byte of a pixel (BGR order: 1868, 4899; RGB order: 4899, 1868).
// one workgroup's four tiles (a wave = 256 source pixels x 2^LV rows -> one output row segment): wgx = tile group, wgy = image
template <typename T, int NOC, int LV, bool FAST, int SRCC>
__device__ __forceinline__ void pyr_base_tiles(
    const T *__restrict__ frames0, const T *__restrict__ frames1, int n_per_src, long frame_stride,  // 2 x n frames, h_org x w_org x SRCC
    int w_org, int h_org, int left, int top,               // padding offsets (floor(pad/2))
    int Wp, int Hp,                                        // padded frame size
    float *__restrict__ dst0, float *__restrict__ dst1, long dst_stride, int tw, int ps,  // level LV padded buffers
    int coef0, int coef2, const int wgx, const int wgy)
    // FAST: 16-B loads legal (no horizontal pad, 16-B aligned rows)
{
  constexpr bool C2G = SRCC != NOC;
  static_assert(!C2G || (SRCC == 3 && NOC == 1 && sizeof(T) == 1), "colour -> gray on load: 8-bit, three channels");
  auto gray3 = [&](unsigned b0, unsigned b1, unsigned b2) {
    return (float)((b0 * (unsigned)coef0 + b1 * 9617u + b2 * (unsigned)coef2 + 8192u) >> 14);
  };
  constexpr int R = 1 << LV;                 // source rows per output row
  constexpr int C = 4 * NOC;                 // floats per lane per row
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int strips = (Wp + 255) >> 8;
  const WgId wg = {wgx, wgy};
  const int tile = wg.x * 4 + wave;
  const int oh = Hp >> LV;
  if (tile >= strips * oh) return;
  const int strip = tile % strips, oy = tile / strips;
  const int which = wg.y >= n_per_src, img = wg.y - (which ? n_per_src : 0);   // I0 batch first, then I1 batch
  const T *src = (which ? frames1 : frames0) + (size_t)img * frame_stride;
  float *dst = which ? dst1 : dst0;
  const int x0 = strip * 256 + lane * 4;     // first of this lane's 4 source pixels (padded coords)
  const bool active = x0 < Wp;               // Wp is a multiple of 4 whenever LV >= 2
  // one source row (clamped: replicate padding) -> 4 pixels x NOC floats of this lane
  auto load_row = [&](int r, float (&dstv)[C]) {
    const int sy = clampi(oy * R + r - top, h_org);
    const T *row = src + (size_t)sy * w_org * SRCC;
    if constexpr (C2G) {
      if constexpr (FAST) {
        // 4 pixels x 3 bytes = three dwords per lane and row
        const unsigned *p32 = reinterpret_cast<const unsigned *>(row + (size_t)x0 * 3);
        const unsigned t0 = p32[0], t1 = p32[1], t2 = p32[2];
        dstv[0] = gray3(t0 & 0xffu, (t0 >> 8) & 0xffu, (t0 >> 16) & 0xffu);
        dstv[1] = gray3(t0 >> 24, t1 & 0xffu, (t1 >> 8) & 0xffu);
        dstv[2] = gray3((t1 >> 16) & 0xffu, t1 >> 24, t2 & 0xffu);
        dstv[3] = gray3((t2 >> 8) & 0xffu, (t2 >> 16) & 0xffu, t2 >> 24);
      } else {
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          const int sx = clampi(x0 + px - left, w_org);
          dstv[px] = gray3(row[(size_t)sx * 3], row[(size_t)sx * 3 + 1], row[(size_t)sx * 3 + 2]);
        }
      }
    } else if constexpr (FAST && sizeof(T) == 1) {
      // 8-bit frames ("next" row f2 of SURVEY 8f): 4 pixels x NOC bytes = NOC dwords per lane and row; u8 -> f32 is exact
      const unsigned *p32 = reinterpret_cast<const unsigned *>(row + (size_t)x0 * NOC);
#pragma unroll
      for (int k = 0; k < NOC; ++k) {
        const unsigned t = NOC == 1 ? __builtin_nontemporal_load(p32 + k) : p32[k];
        dstv[4 * k] = (float)(t & 0xffu); dstv[4 * k + 1] = (float)((t >> 8) & 0xffu);
        dstv[4 * k + 2] = (float)((t >> 16) & 0xffu); dstv[4 * k + 3] = (float)(t >> 24);
      }
    } else if constexpr (FAST) {
      // streamed once, never re-read: nontemporal 16-B loads keep the frames out of L2/MALL
      typedef float vf4 __attribute__((ext_vector_type(4)));
      const vf4 *p4 = reinterpret_cast<const vf4 *>(reinterpret_cast<const float *>(row) + (size_t)x0 * NOC);
#pragma unroll
      for (int k = 0; k < NOC; ++k) {
        // (three channels: a lane's three 16-byte pieces share cache lines with its neighbours' -- keep them cacheable)
        const vf4 t = NOC == 1 ? __builtin_nontemporal_load(p4 + k) : p4[k];
        dstv[4 * k] = t.x; dstv[4 * k + 1] = t.y; dstv[4 * k + 2] = t.z; dstv[4 * k + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const int sx = clampi(x0 + px - left, w_org);
#pragma unroll
        for (int c = 0; c < NOC; ++c) dstv[px * NOC + c] = (float)row[(size_t)sx * NOC + c];
      }
    }
  };
  float *out = dst + (size_t)img * dst_stride;
  if constexpr (LV == 0) {
    if (active) {
      float v[1][C];
      load_row(0, v[0]);
#pragma unroll
      for (int px = 0; px < 4; ++px)
        if (x0 + px < Wp)
#pragma unroll
          for (int c = 0; c < NOC; ++c) out[((size_t)(oy + ps) * tw + (x0 + px + ps)) * NOC + c] = v[0][px * NOC + c];
    }
    return;
  } else {
    // level 1: 2 px per lane.  The rows are loaded in groups of 4 (one level-2 row); with three channels at most 8 rows
    // (24 16-byte loads) are kept in flight per lane -- all 16 would need 192 VGPRs and leave 2 waves per SIMD
    constexpr int RG = R >= 4 ? 4 : R;                               // rows per group
    constexpr int NGRP = R / RG;
    constexpr int GROUPS_IN_FLIGHT = (NOC == 1 || sizeof(T) == 1) ? NGRP : 2;
    // Three-channel f32 frames: a lane's 4 pixels are 48 contiguous bytes, so lane-private 16-byte loads have a 48-byte lane
    // stride and every 128-byte line is requested by three different instructions.  Instead the wave loads its 3072-byte row
    // segment with three fully coalesced instructions (lane L: bytes 1024 k + 16 L) and transposes through a wave-private LDS
    // slab (written linearly, read back at 48 L + 16 k: 16 lanes cover the 64 banks exactly, no conflicts).  LDS executes a
    // wave's accesses in order, so no barrier is needed.
    constexpr bool XPOSE = FAST && NOC == 3 && sizeof(T) == 4 && R >= 4;
    typedef float vf4 __attribute__((ext_vector_type(4)));
    __shared__ vf4 xpose[XPOSE ? 4 : 1][XPOSE ? RG : 1][XPOSE ? 192 : 1];
    // 8-bit gray frames: a lane's 4 pixels are one dword, 256 bytes per wave and row.  Where the rows are 16-byte aligned the
    // wave instead fetches FOUR rows with one 16-byte load per lane (lane L: row L/16, bytes 16 (L%16)..) and hands the dwords
    // out through a wave-private 1 KB LDS slab -- a quarter of the memory instructions for the same bytes.
    constexpr bool XPOSE8 = FAST && NOC == 1 && sizeof(T) == 1 && R >= 4 && !C2G;
    typedef unsigned vu4 __attribute__((ext_vector_type(4)));
    __shared__ vu4 xpose8[XPOSE8 ? 4 : 1][XPOSE8 ? 64 : 1];
    const bool x8 = XPOSE8 && (w_org & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    // 8-bit colour -> gray: a row segment of the wave is 768 bytes, four rows 3072 = three fully coalesced 16-byte loads per lane
    // (piece i = 64 k + lane: row i / 48, bytes 16 (i % 48)..), handed out through a wave-private 3 KB slab: lane L reads its
    // 12 bytes of row r at 768 r + 12 L (a 3-dword lane stride: no bank conflicts)
    constexpr bool XPOSEC = FAST && C2G && R >= 4;
    __shared__ vu4 xposec[XPOSEC ? 4 : 1][XPOSEC ? 192 : 1];
    const bool xc = XPOSEC && (w_org & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    float l1[R / 2][2 * NOC];
    vu4 t8[XPOSE8 ? NGRP : 1];
    vu4 tc[XPOSEC ? NGRP : 1][3];
    if constexpr (XPOSEC) {
      if (xc) {
        const int segb = (Wp - strip * 256 < 256 ? Wp - strip * 256 : 256) * 3;        // bytes of this strip's row segment
#pragma unroll
        for (int g = 0; g < NGRP; ++g)
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int i = k * 64 + lane, rr = i / 48, c16 = i - rr * 48;
            const int sy = clampi(oy * R + g * RG + rr - top, h_org);
            const vu4 *p = reinterpret_cast<const vu4 *>(reinterpret_cast<const unsigned char *>(src) + ((size_t)sy * w_org + strip * 256) * 3) + c16;
            tc[g][k] = (c16 * 16 < segb) ? __builtin_nontemporal_load(p) : vu4{0u, 0u, 0u, 0u};
          }
      }
    }
    if constexpr (XPOSE8) {
      if (x8) {
        const int segb = Wp - strip * 256 < 256 ? Wp - strip * 256 : 256;              // bytes of this strip's row segment
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
          const int sy = clampi(oy * R + g * RG + (lane >> 4) - top, h_org);
          const vu4 *p = reinterpret_cast<const vu4 *>(reinterpret_cast<const unsigned char *>(src) + (size_t)sy * w_org + strip * 256) + (lane & 15);
          t8[g] = ((lane & 15) * 16 < segb) ? __builtin_nontemporal_load(p) : vu4{0u, 0u, 0u, 0u};
        }
      }
    }
#pragma unroll
    for (int g = 0; g < NGRP; ++g) {
      float v[RG][C];
      if constexpr (XPOSE) {
        const int seg4 = ((Wp - strip * 256 < 256 ? Wp - strip * 256 : 256) * 3) / 4;     // float4s of this strip's row segment
        vf4 t[RG][3];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
          const int sy = clampi(oy * R + g * RG + r - top, h_org);
          const vf4 *p4 = reinterpret_cast<const vf4 *>(reinterpret_cast<const float *>(src) + ((size_t)sy * w_org + strip * 256) * 3);
#pragma unroll
          for (int k = 0; k < 3; ++k) t[r][k] = (k * 64 + lane < seg4) ? __builtin_nontemporal_load(p4 + k * 64 + lane) : vf4{0.f, 0.f, 0.f, 0.f};
        }
        // The slab is how the LANES of this wave exchange data, and the compiler reasons per thread: it may prove that a
        // thread's own writes never hit the cell it reads (64 k + lane != 3 lane + 1 for every lane) and then reuse the
        // value it read for the previous row group, or move the next group's writes above this group's reads.  The
        // compiler barriers pin write phase | read phase | next write phase; the hardware keeps a wave's LDS accesses in order.
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
          for (int k = 0; k < 3; ++k) xpose[wave][r][k * 64 + lane] = t[r][k];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const vf4 q = xpose[wave][r][3 * lane + k];
            v[r][4 * k] = q.x; v[r][4 * k + 1] = q.y; v[r][4 * k + 2] = q.z; v[r][4 * k + 3] = q.w;
          }
        asm volatile("" ::: "memory");
      } else if (XPOSEC && xc) {
        if constexpr (XPOSEC) {
          asm volatile("" ::: "memory");               // same lane-to-lane hand-over as above
#pragma unroll
          for (int k = 0; k < 3; ++k) xposec[wave][k * 64 + lane] = tc[g][k];
          asm volatile("" ::: "memory");
          const unsigned *slab = reinterpret_cast<const unsigned *>(&xposec[wave][0]);
#pragma unroll
          for (int r = 0; r < RG; ++r) {
            const unsigned t0 = slab[r * 192 + lane * 3], t1 = slab[r * 192 + lane * 3 + 1], t2 = slab[r * 192 + lane * 3 + 2];
            v[r][0] = gray3(t0 & 0xffu, (t0 >> 8) & 0xffu, (t0 >> 16) & 0xffu);
            v[r][1] = gray3(t0 >> 24, t1 & 0xffu, (t1 >> 8) & 0xffu);
            v[r][2] = gray3((t1 >> 16) & 0xffu, t1 >> 24, t2 & 0xffu);
            v[r][3] = gray3((t2 >> 8) & 0xffu, (t2 >> 16) & 0xffu, t2 >> 24);
          }
          asm volatile("" ::: "memory");
        }
      } else if (XPOSE8 && x8) {
        if constexpr (XPOSE8) {
          asm volatile("" ::: "memory");               // same lane-to-lane hand-over as above
          xpose8[wave][lane] = t8[g];
          asm volatile("" ::: "memory");
          const unsigned *slab = reinterpret_cast<const unsigned *>(&xpose8[wave][0]);
#pragma unroll
          for (int r = 0; r < RG; ++r) {
            const unsigned t = slab[r * 64 + lane];
            v[r][0] = (float)(t & 0xffu); v[r][1] = (float)((t >> 8) & 0xffu); v[r][2] = (float)((t >> 16) & 0xffu); v[r][3] = (float)(t >> 24);
          }
          asm volatile("" ::: "memory");
        }
      } else if (active) {
#pragma unroll
        for (int r = 0; r < RG; ++r) load_row(g * RG + r, v[r]);
      } else {
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
          for (int k = 0; k < C; ++k) v[r][k] = 0.f;
      }
#pragma unroll
      for (int y = 0; y < RG / 2; ++y)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
          for (int c = 0; c < NOC; ++c) {
            const float a = v[2 * y][(2 * px) * NOC + c], b = v[2 * y][(2 * px + 1) * NOC + c];
            const float cc = v[2 * y + 1][(2 * px) * NOC + c], d = v[2 * y + 1][(2 * px + 1) * NOC + c];
            l1[g * (RG / 2) + y][px * NOC + c] = ((a + cc) + (b + d)) * 0.25f;
          }
      if ((g + 1) % GROUPS_IN_FLIGHT == 0 && g + 1 < NGRP) asm volatile("" ::: "memory");   // later rows' loads stay behind this point
    }
    if constexpr (LV == 1) {
      if (active) {
        const int ox = x0 >> 1;
#pragma unroll
        for (int px = 0; px < 2; ++px)
          if (ox + px < (Wp >> 1))
#pragma unroll
            for (int c = 0; c < NOC; ++c) out[((size_t)(oy + ps) * tw + (ox + px + ps)) * NOC + c] = l1[0][px * NOC + c];
      }
      return;
    } else {
      // level 2: 1 px per lane
      float cur[R / 4][NOC];
#pragma unroll
      for (int y = 0; y < R / 4; ++y)
#pragma unroll
        for (int c = 0; c < NOC; ++c) {
          const float a = l1[2 * y][c], b = l1[2 * y][NOC + c], cc = l1[2 * y + 1][c], d = l1[2 * y + 1][NOC + c];
          cur[y][c] = ((a + cc) + (b + d)) * 0.25f;
        }
      // levels 3..LV: partner lanes lane^1, lane^2 hold the neighbouring column
      int rows = R / 4;
#pragma unroll
      for (int k = 3; k <= LV; ++k) {
        const int xm = 1 << (k - 3);
#pragma unroll
        for (int y = 0; y < (R >> k); ++y)
#pragma unroll
          for (int c = 0; c < NOC; ++c) {
            const float s = cur[2 * y][c] + cur[2 * y + 1][c];     // (a+c) of this column
            // (b+d) of the partner column: lane^1 / lane^2 by DPP quad_perm (no LDS round trip)
            const float t = __builtin_bit_cast(float, xm == 1 ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true)
                                                                : __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true));
            cur[y][c] = (s + t) * 0.25f;
          }
        rows >>= 1;
      }
      (void)rows;
      constexpr int G = 1 << (LV - 2);          // lanes sharing one output pixel
      if (active && (lane % G) == 0) {
        const int ox = (strip * 256 >> LV) + lane / G;
#pragma unroll
        for (int c = 0; c < NOC; ++c) out[((size_t)(oy + ps) * tw + (ox + ps)) * NOC + c] = cur[0][c];
      }
    }
  }
}

// The kernel: grid (tile groups, images)
template <typename T, int NOC, int LV, bool FAST, int SRCC = NOC>
__global__ __launch_bounds__(256) void pyr_base_kernel(
    const T *__restrict__ frames0, const T *__restrict__ frames1, int n_per_src, long frame_stride,
    int w_org, int h_org, int left, int top, int Wp, int Hp,
    float *__restrict__ dst0, float *__restrict__ dst1, long dst_stride, int tw, int ps,
    int coef0 = 0, int coef2 = 0)
{
  const WgId wg = xcd_local_wg();            // image k (pair k % n) on XCD k % 8, where pyr_finish and the per-pair kernels run
  pyr_base_tiles<T, NOC, LV, FAST, SRCC>(frames0, frames1, n_per_src, frame_stride, w_org, h_org, left, top, Wp, Hp, dst0, dst1, dst_stride, tw, ps, coef0, coef2, wg.x, wg.y);
}

// level l (padded buffer, interior valid) -> level l+1 interior
template <int NOC>
__global__ __launch_bounds__(256) void pyr_halve_kernel(const float *__restrict__ src, long src_stride, int stw,
                                                        float *__restrict__ dst, long dst_stride, int dtw,
                                                        int dw, int dh, int ps,
                                                        const float *__restrict__ src_b = nullptr, float *__restrict__ dst_b = nullptr, int n_a = 1 << 30)
{
  // images n_a.. of the launch come from / go to a second pair of buffers (the target frames of the batch): both frames' levels
  // in one launch
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= dw * dh * NOC) return;
  if ((int)blockIdx.y >= n_a) {
    src = src_b - (size_t)n_a * src_stride;
    dst = dst_b - (size_t)n_a * dst_stride;
  }
  const int c = idx % NOC, x = (idx / NOC) % dw, y = idx / (NOC * dw);
  const float *s = src + (size_t)blockIdx.y * src_stride;
  const float a = s[((size_t)(2 * y + ps) * stw + (2 * x + ps)) * NOC + c];
  const float b = s[((size_t)(2 * y + ps) * stw + (2 * x + 1 + ps)) * NOC + c];
  const float cc = s[((size_t)(2 * y + 1 + ps) * stw + (2 * x + ps)) * NOC + c];
  const float d = s[((size_t)(2 * y + 1 + ps) * stw + (2 * x + 1 + ps)) * NOC + c];
  dst[(size_t)blockIdx.y * dst_stride + ((size_t)(y + ps) * dtw + (x + ps)) * NOC + c] = ((a + cc) + (b + d)) * 0.25f;
}

// one thread per element of the padded level buffer
template <int NOC>
__global__ __launch_bounds__(256) void pyr_border_grad_kernel(float *__restrict__ im, float *__restrict__ dx,
                                                              float *__restrict__ dy, long stride,
                                                              int w, int h, int ps)
{
  const int tw = w + 2 * ps, th = h + 2 * ps;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= tw * th * NOC) return;
  const int c = idx % NOC, X = (idx / NOC) % tw, Y = idx / (NOC * tw);
  float *I = im + (size_t)blockIdx.y * stride;
  const int x = X - ps, y = Y - ps;
  const bool inside = x >= 0 && x < w && y >= 0 && y < h;
#define PIX(yy, xx) I[((size_t)((yy) + ps) * tw + ((xx) + ps)) * NOC + c]
  if (!inside) I[idx] = PIX(clampi(y, h), clampi(x, w));                    // copyMakeBorder REPLICATE (:166)
  if (dx) {
    float gx = 0.f, gy = 0.f;                                                 // BORDER_CONSTANT 0 (:171-172)
    if (inside) {                                                             // Sobel ksize=1, REFLECT_101 (:156-157)
      gx = PIX(y, reflect101(x + 1, w)) - PIX(y, reflect101(x - 1, w));
      gy = PIX(reflect101(y + 1, h), x) - PIX(reflect101(y - 1, h), x);
    }
    dx[(size_t)blockIdx.y * stride + idx] = gx;
    dy[(size_t)blockIdx.y * stride + idx] = gy;
  }
#undef PIX
}

// borders + gradients of SEVERAL levels of both frames of a batch in one launch: blockIdx.z = level, blockIdx.y = image (template
// frames first, then n_b target frames, which get no gradients); the grid is sized for the largest level
struct PyrBorderArgs {
  float *im[2][FOTG_MAXLEV], *dx[FOTG_MAXLEV], *dy[FOTG_MAXLEV];
  long stride[FOTG_MAXLEV];
  int w[FOTG_MAXLEV], h[FOTG_MAXLEV];
  int n_a, ps;
};
template <int NOC>
__global__ __launch_bounds__(256) void pyr_border_grad_multi_kernel(PyrBorderArgs a)
{
  const int k = blockIdx.z, w = a.w[k], h = a.h[k], ps = a.ps;
  const int tw = w + 2 * ps, th = h + 2 * ps;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= tw * th * NOC) return;
  const bool second = (int)blockIdx.y >= a.n_a;
  const int img = second ? blockIdx.y - a.n_a : blockIdx.y;
  const long stride = a.stride[k];
  const int c = idx % NOC, X = (idx / NOC) % tw, Y = idx / (NOC * tw);
  float *I = a.im[second ? 1 : 0][k] + (size_t)img * stride;
  const int x = X - ps, y = Y - ps;
  const bool inside = x >= 0 && x < w && y >= 0 && y < h;
#define PIX(yy, xx) I[((size_t)((yy) + ps) * tw + ((xx) + ps)) * NOC + c]
  if (!inside) I[idx] = PIX(clampi(y, h), clampi(x, w));                    // copyMakeBorder REPLICATE (:166)
  if (!second) {
    float gx = 0.f, gy = 0.f;                                                 // BORDER_CONSTANT 0 (:171-172)
    if (inside) {                                                             // Sobel ksize=1, REFLECT_101 (:156-157)
      gx = PIX(y, reflect101(x + 1, w)) - PIX(y, reflect101(x - 1, w));
      gy = PIX(reflect101(y + 1, h), x) - PIX(reflect101(y - 1, h), x);
    }
    a.dx[k][(size_t)img * stride + idx] = gx;
    a.dy[k][(size_t)img * stride + idx] = gy;
  }
#undef PIX
}

// Everything after the base level for the small levels of one image, in ONE launch (one workgroup per image):
// halve down to the coarsest level, then borders + gradients of every used level.  Replaces 2 + 3 tiny launches per
// frame batch at 1080p (each ~5 us) by one.
struct PyrFinishArgs {
  float *im[2][FOTG_MAXLEV];      // [which][k], k = level - base
  float *dx[FOTG_MAXLEV], *dy[FOTG_MAXLEV];   // I0 only
  long stride[FOTG_MAXLEV];
  int w[FOTG_MAXLEV], h[FOTG_MAXLEV];
  int nlev, first_used, ps, n_per_src;
};

template <int NOC>
__global__ __launch_bounds__(1024) void pyr_finish_kernel(PyrFinishArgs a)
{
  const int which = blockIdx.x >= a.n_per_src, img = blockIdx.x - (which ? a.n_per_src : 0);
  const int ps = a.ps;
  // two workgroups per image: blockIdx.y = 1 does border + gradients of the base level (its interior is complete when this
  // kernel starts), blockIdx.y = 0 the chain halve -> halve -> borders + gradients of the coarser levels
  const int part = blockIdx.y, kb = part == 1 ? a.first_used : (a.first_used == 0 && gridDim.y > 1 ? 1 : a.first_used);
  const int ke = part == 1 ? (a.first_used == 0 ? 1 : 0) : a.nlev;
  if (part == 0)
  for (int k = 1; k < a.nlev; ++k) {
    const float *s = a.im[which][k - 1] + (size_t)img * a.stride[k - 1];
    float *d = a.im[which][k] + (size_t)img * a.stride[k];
    const int stw = a.w[k - 1] + 2 * ps, dtw = a.w[k] + 2 * ps, dw = a.w[k], dh = a.h[k];
    for (int idx = threadIdx.x; idx < dw * dh * NOC; idx += blockDim.x) {
      const int c = idx % NOC, x = (idx / NOC) % dw, y = idx / (NOC * dw);
      const float p00 = s[((size_t)(2 * y + ps) * stw + (2 * x + ps)) * NOC + c];
      const float p01 = s[((size_t)(2 * y + ps) * stw + (2 * x + 1 + ps)) * NOC + c];
      const float p10 = s[((size_t)(2 * y + 1 + ps) * stw + (2 * x + ps)) * NOC + c];
      const float p11 = s[((size_t)(2 * y + 1 + ps) * stw + (2 * x + 1 + ps)) * NOC + c];
      d[((size_t)(y + ps) * dtw + (x + ps)) * NOC + c] = ((p00 + p10) + (p01 + p11)) * 0.25f;
    }
    __syncthreads();
  }
  for (int k = kb; k < ke; ++k) {
    const int w = a.w[k], h = a.h[k], tw = w + 2 * ps, th = h + 2 * ps;
    float *I = a.im[which][k] + (size_t)img * a.stride[k];
    float *gx = which == 0 ? a.dx[k] + (size_t)img * a.stride[k] : nullptr;
    float *gy = which == 0 ? a.dy[k] + (size_t)img * a.stride[k] : nullptr;
    for (int idx = threadIdx.x; idx < tw * th * NOC; idx += blockDim.x) {
      const int c = idx % NOC, X = (idx / NOC) % tw, Y = idx / (NOC * tw);
      const int x = X - ps, y = Y - ps;
      const bool inside = x >= 0 && x < w && y >= 0 && y < h;
#define PIX(yy, xx) I[((size_t)((yy) + ps) * tw + ((xx) + ps)) * NOC + c]
      if (!inside) I[idx] = PIX(clampi(y, h), clampi(x, w));
      if (gx) {
        float vx = 0.f, vy = 0.f;
        if (inside) {
          vx = PIX(y, reflect101(x + 1, w)) - PIX(y, reflect101(x - 1, w));
          vy = PIX(reflect101(y + 1, h), x) - PIX(reflect101(y - 1, h), x);
        }
        gx[idx] = vx;
        gy[idx] = vy;
      }
#undef PIX
    }
  }
}

}  // namespace fotg
