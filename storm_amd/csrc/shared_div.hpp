#pragma once
// Several FP64 divisions by ONE denominator, bit-identical to `a / x`.
//
// hipcc expands an IEEE f64 division into  v_div_scale x2, v_rcp_f64, two Newton steps on the reciprocal
// (4 fma), q = n*r, one residual fma, v_div_fmas, v_div_fixup.  Everything up to the refined reciprocal depends on
// the denominator only, so k divisions by the same x can share it: 8 instruction slots once + 3 per division
// instead of ~13 each.  The shortcut is exact only while v_div_scale would not rescale and v_div_fixup has
// nothing to fix, i.e. for finite, normal, non-zero operands of moderate exponent; anything else takes the real
// division.  tools/shared_div_check.hip compares the two bit for bit on the device over random and edge operands.
#include <hip/hip_runtime.h>

namespace storm {

struct SharedDenominator {
  double x, r;
  bool ok;
};

// |v| in [2^-400, 2^400]: far from every rescaling threshold of v_div_scale_f64
__device__ __forceinline__ bool shared_div_moderate(double v) {
  const int e = (__double2hiint(v) >> 20) & 0x7ff;
  return e >= 1023 - 400 && e <= 1023 + 400;
}

__device__ __forceinline__ SharedDenominator shared_div_prepare(double x) {
  SharedDenominator d;
  d.x = x;
  d.ok = shared_div_moderate(x);
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  d.r = __builtin_fma(r, e, r);
  return d;
}

// One quotient.  The fallback is a per-lane branch: fine for an occasional division, see shared_div6 for a batch.
__device__ __forceinline__ double shared_div(double a, const SharedDenominator& d) {
  const double q = a * d.r;
  const double e = __builtin_fma(-d.x, q, a);
  double res = __builtin_fma(e, d.r, q);
  if (!(d.ok && shared_div_moderate(a))) res = a / d.x;
  return res;
}

// Six quotients n[k] / x in place, with ONE wave-uniform decision: if every lane of the wave has moderate operands
// the whole wave takes the shortcut, otherwise the whole wave divides for real (same bits either way).
__device__ __forceinline__ void shared_div6(double (&n)[6], const SharedDenominator& d) {
  unsigned lo = 0x7fffffffu, hi = 0u;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const unsigned w = unsigned(__double2hiint(n[k])) & 0x7fffffffu;  // exponent and top of the mantissa, sign off
    lo = w < lo ? w : lo;
    hi = w > hi ? w : hi;
  }
  const bool ok = d.ok && lo >= (unsigned(1023 - 400) << 20) && hi < (unsigned(1023 + 401) << 20);
  if (__all(ok)) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double q = n[k] * d.r;
      const double e = __builtin_fma(-d.x, q, n[k]);
      n[k] = __builtin_fma(e, d.r, q);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) n[k] = n[k] / d.x;
  }
}

// Two quotients, same contract as shared_div6.
__device__ __forceinline__ void shared_div2(double& n0, double& n1, const SharedDenominator& d) {
  const unsigned w0 = unsigned(__double2hiint(n0)) & 0x7fffffffu, w1 = unsigned(__double2hiint(n1)) & 0x7fffffffu;
  const unsigned lo = w0 < w1 ? w0 : w1, hi = w0 < w1 ? w1 : w0;
  const bool ok = d.ok && lo >= (unsigned(1023 - 400) << 20) && hi < (unsigned(1023 + 401) << 20);
  if (__all(ok)) {
    const double q0 = n0 * d.r, q1 = n1 * d.r;
    const double e0 = __builtin_fma(-d.x, q0, n0), e1 = __builtin_fma(-d.x, q1, n1);
    n0 = __builtin_fma(e0, d.r, q0);
    n1 = __builtin_fma(e1, d.r, q1);
  } else {
    n0 = n0 / d.x;
    n1 = n1 / d.x;
  }
}

}  // namespace storm
