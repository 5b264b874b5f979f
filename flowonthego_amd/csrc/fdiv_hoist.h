// fdiv_hoist.h -- x / d for a LOOP-INVARIANT denominator d, bit-identical to the IEEE division.
//
// The compiler's correctly rounded f32 division is eleven instructions (gfx950, denormals on):
//     s = v_div_scale(d)   r0 = v_rcp(s)   e = fma(-s, r0, 1)   r = fma(e, r0, r0)            <- depend on d only
//     n = v_div_scale(x)   q0 = n r   e2 = fma(-s, q0, n)   q1 = fma(e2, r, q0)   e3 = fma(-s, q1, n)
//     q = v_div_fmas(e3, r, q1)   result = v_div_fixup(q, d, x)
// v_div_scale returns its operand unchanged (and v_div_fmas is a plain fma, v_div_fixup the identity) unless d is denormal or
// >= 2^126, x is zero / < 2^-103 / not finite, or the quotient is denormal or >= 2^96.  make_invdiv() runs the first line once;
// fdiv_fast() is the second and third line without the scaling and fix-up steps: five instructions.  The caller must check
// fdiv_in_range() of the quotient (together with InvDiv::ok: 2^-40 <= |d| <= 2^40): it implies 2^-80 <= |x| <= 2^80 and a
// quotient far from the denormal / overflow ranges, i.e. every scaling step would have been the identity; anything else -- zero,
// tiny, huge or non-finite numerators -- goes through the compiler's division.  tools/div_hoist_probe.hip compares the guarded
// fast quotient with x / d on 2^35 pairs including the edge classes: no mismatch.
#pragma once
#include <hip/hip_runtime.h>

namespace fotg {

struct InvDiv { float d, r; bool ok; };

__device__ __forceinline__ InvDiv make_invdiv(float d)
{
  InvDiv v;
  v.d = d;
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r0, 1.0f);
  v.r = __builtin_fmaf(e, r0, r0);
  v.ok = __builtin_fabsf(d) >= 0x1p-40f && __builtin_fabsf(d) <= 0x1p40f;
  return v;
}

__device__ __forceinline__ float fdiv_fast(float x, const InvDiv &v)
{
  const float q0 = x * v.r;
  const float e2 = __builtin_fmaf(-v.d, q0, x);
  const float q1 = __builtin_fmaf(e2, v.r, q0);
  const float e3 = __builtin_fmaf(-v.d, q1, x);
  return __builtin_fmaf(e3, v.r, q1);
}

__device__ __forceinline__ bool fdiv_in_range(float q) { return __builtin_fabsf(q) >= 0x1p-40f && __builtin_fabsf(q) <= 0x1p40f; }

}  // namespace fotg
