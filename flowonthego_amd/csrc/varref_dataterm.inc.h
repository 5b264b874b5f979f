// smooth_w() and data_term_cell(): included twice by varref.hip.h.
//   FOTG_DT_FAST 0 -- the parity mode: every operation as the reference performs it (IEEE divisions and square roots, no contraction)
//   FOTG_DT_FAST 1 -- the tolerance mode (fotg_params::fast_math): one v_rcp per divisor, v_rsq for 1 / sqrt (the eleven IEEE divisions and
//                     three square roots of a gray cell are a third of vr_data_kernel's 475 VALU instructions).  No contraction: the
//                     compiler would fuse differently in every kernel that inlines this, and the mode's results must not depend on
//                     which kernel ran (a pair alone = the same pair in a batch, level pipeline = launch per stage).
// FOTG_DT_NAME(x) names the two copies (x_exact / x_fast).
#undef FOTG_DT_INV
#undef FOTG_DT_DIV
#undef FOTG_DT_RSQ
#if FOTG_DT_FAST
#define FOTG_DT_INV(r, y) const float r = __builtin_amdgcn_rcpf(y)
#define FOTG_DT_DIV(x, y, r) ((x) * (r))
#define FOTG_DT_RSQ(k, x) ((k) * __builtin_amdgcn_rsqf(x))
#else
#define FOTG_DT_INV(r, y) const float r = 0.f
#define FOTG_DT_DIV(x, y, r) ((x) / (y))
#define FOTG_DT_RSQ(k, x) ((k) / sqrtf(x))
#endif

__device__ __forceinline__ float FOTG_DT_NAME(smooth_w)(float2 l, float2 c, float2 r, float2 t, float2 b, int j, int h, float quarter_alpha)
{
  const float c0 = -0.5f, c1 = -0.0f, c2 = 0.5f;
  const float ux = c0 * l.x + c1 * c.x + c2 * r.x;
  const float vx = c0 * l.y + c1 * c.y + c2 * r.y;
  float uy, vy;
  if (j == 0) { uy = (c0 + c1) * c.x + c2 * b.x; vy = (c0 + c1) * c.y + c2 * b.y; }
  else if (j == h - 1) { uy = c0 * t.x + (c1 + c2) * c.x; vy = c0 * t.y + (c1 + c2) * c.y; }
  else { uy = c0 * t.x + c1 * c.x + c2 * b.x; vy = c0 * t.y + c1 * c.y + c2 * b.y; }
  const float eps = 0.001f * 0.001f;
  return FOTG_DT_RSQ(quarter_alpha, ux * ux + uy * uy + vx * vx + vy * vy + eps);
}


// compute_data (:310-438) + sub_laplacian (:172-199) + the 2x2 block inverse of sor_coupled's first sweep
// (solver.c:115-120) for pixel (i,j), given the four smoothness pair sums and (du,dv); writes the skewed system cell.
template <int NOC>
__device__ __forceinline__ void FOTG_DT_NAME(data_term_cell)(const VrArgs &a, int i, int j, const PixIn<NOC> &p, float hr, float hl, float vb, float vt,
                                               float u, float v, float half_delta_over3, float half_gamma_over3, float4 &c0, float4 &c1)
{
  const int w = a.w, h = a.h;
  // compute_data (:310-438)
  const float dnorm = 0.1f * 0.1f, epsc = 0.001f * 0.001f, epsg = 0.001f * 0.001f;
  const float m = p.m;
  float A11 = 0, A12 = 0, A22 = 0, B1 = 0, B2 = 0;
  if constexpr (NOC == 1) {
    const float Ix = p.Ix[0], Iy = p.Iy[0], Iz = p.Iz[0], Ixx = p.Ixx[0], Ixy = p.Ixy[0], Iyy = p.Iyy[0], Ixz = p.Ixz[0], Iyz = p.Iyz[0];
    float tmp, tmp2, n1, n2;
    if (half_delta_over3) {
      tmp = Iz + Ix * u + Iy * v;
      n1 = Ix * Ix + Iy * Iy + dnorm;
      FOTG_DT_INV(rn1, n1);
      tmp = FOTG_DT_RSQ(m * half_delta_over3, FOTG_DT_DIV(3 * tmp * tmp, n1, rn1) + epsc);
      tmp = FOTG_DT_DIV(tmp, n1, rn1);
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
    FOTG_DT_INV(rg1, n1); FOTG_DT_INV(rg2, n2);
    tmp = FOTG_DT_RSQ(m * half_gamma_over3, FOTG_DT_DIV(3 * tmp * tmp, n1, rg1) + FOTG_DT_DIV(3 * tmp2 * tmp2, n2, rg2) + epsg);
    tmp2 = FOTG_DT_DIV(tmp, n2, rg2); tmp = FOTG_DT_DIV(tmp, n1, rg1);
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
      ix[c] = p.Ix[c]; iy[c] = p.Iy[c]; iz[c] = p.Iz[c]; ixx[c] = p.Ixx[c]; ixy[c] = p.Ixy[c]; iyy[c] = p.Iyy[c];
      ixz[c] = p.Ixz[c]; iyz[c] = p.Iyz[c];
    }
    if (half_delta_over3) {
      float t[3], n[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { t[c] = iz[c] + ix[c] * u + iy[c] * v; n[c] = ix[c] * ix[c] + iy[c] * iy[c] + dnorm; }
      FOTG_DT_INV(rc0, n[0]); FOTG_DT_INV(rc1, n[1]); FOTG_DT_INV(rc2, n[2]);
      float tmp = FOTG_DT_RSQ(m * half_delta_over3, FOTG_DT_DIV(t[0] * t[0], n[0], rc0) + FOTG_DT_DIV(t[1] * t[1], n[1], rc1) + FOTG_DT_DIV(t[2] * t[2], n[2], rc2) + epsc);
      const float k2 = FOTG_DT_DIV(tmp, n[2], rc2), k1 = FOTG_DT_DIV(tmp, n[1], rc1), k0 = FOTG_DT_DIV(tmp, n[0], rc0);
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
    float r1[3], r2[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { FOTG_DT_INV(ra, n1[c]); FOTG_DT_INV(rb, n2[c]); r1[c] = ra; r2[c] = rb; }
    const float tmp = FOTG_DT_RSQ(m * half_gamma_over3, FOTG_DT_DIV(t1[0] * t1[0], n1[0], r1[0]) + FOTG_DT_DIV(t2[0] * t2[0], n2[0], r2[0]) + FOTG_DT_DIV(t1[1] * t1[1], n1[1], r1[1]) +
                                                        FOTG_DT_DIV(t2[1] * t2[1], n2[1], r2[1]) + FOTG_DT_DIV(t1[2] * t1[2], n1[2], r1[2]) + FOTG_DT_DIV(t2[2] * t2[2], n2[2], r2[2]) + epsg);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float ka = FOTG_DT_DIV(tmp, n1[c], r1[c]), kb = FOTG_DT_DIV(tmp, n2[c], r2[c]);
      A11 += ka * ixx[c] * ixx[c] + kb * ixy[c] * ixy[c];
      A12 += ka * ixx[c] * ixy[c] + kb * ixy[c] * iyy[c];
      A22 += kb * iyy[c] * iyy[c] + ka * ixy[c] * ixy[c];
      B1 -= ka * ixx[c] * ixz[c] + kb * ixy[c] * iyz[c];
      B2 -= kb * iyy[c] * iyz[c] + ka * ixy[c] * ixz[c];
    }
  }

  // sub_laplacian (:172-199) for b1 (src wx) and b2 (src wy): -left, +right, -top, +bottom
  if (i > 0)     { B1 -= hl * p.dxl(); B2 -= hl * p.dyl(); }
  if (i < w - 1) { B1 += hr * p.dxr(); B2 += hr * p.dyr(); }
  if (j > 0)     { B1 -= vt * p.dxt(); B2 -= vt * p.dyt(); }
  if (j < h - 1) { B1 += vb * p.dxb(); B2 += vb * p.dyb(); }

  if (a.point) {
    // sor_coupled_slow_but_readable (solver.c:19-72) keeps the system as it is: A11 + sum_dpsis, A12, A22 + sum_dpsis with
    // sum_dpsis accumulated top, left, bottom, right from 0 (:31-55)
    float sum = 0.0f;
    if (j > 0) sum += vt;
    if (i > 0) sum += hl;
    if (j < h - 1) sum += vb;
    if (i < w - 1) sum += hr;
    c0 = make_float4(A11 + sum, A12, B1, B2);
    c1 = make_float4(A22 + sum, hr, vb, vt);
    return;
  }
  // first sweep of sor_coupled inverts the 2x2 block (solver.c:115-120): dpsis = hl+hr(+vt)(+vb)
  float dps = hl + hr;
  if (j > 0) dps = dps + vt;
  if (j < h - 1) dps = dps + vb;
  const float M11 = A22 + dps, M22 = A11 + dps;
  const float det = M11 * M22 - A12 * A12;
  FOTG_DT_INV(rdet, det);
  c0 = make_float4(FOTG_DT_DIV(M11, det, rdet), FOTG_DT_DIV(A12, -det, -rdet), B1, B2);             // cell layout: (a11', a12', b1, b2 | a22', psi_r, psi_b, psi_t)
  c1 = make_float4(FOTG_DT_DIV(M22, det, rdet), hr, vb, vt);
}

