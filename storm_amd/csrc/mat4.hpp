// mat4.hpp -- register-resident 4x4 FP64 matrices for gfx950 device code.
//
// Everything is fully unrolled with compile-time indices so a matrix lives in 32 VGPRs
// and never touches scratch.  The evaluation order of every routine is part of the
// contract: in the parity build (-ffp-contract=off) results are bit-identical to the
// reference's nalgebra-based arithmetic (reference src/stepper.rs, src/linalg/storage.rs;
// third-party semantics recorded in SURVEY.md 8(c)):
//   * product: per result column, first term a pure product, then k = 1..3 ascending,
//     term added to the running value;
//   * inverse: cofactor expansion on the column-major entries, det from the first row
//     of the input against the first column of cofactors, then multiply by 1/det.
// Column-major: element (r, c) lives at v[c*4 + r].
#pragma once
#include "types.hpp"

namespace STORM_NS {
using namespace storm;

#define SDEV __device__ __forceinline__

struct M4 {
  double v[16];
};

SDEV M4 m4_identity() {
  M4 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r.v[i] = ((i % 5) == 0) ? 1.0 : 0.0;
  return r;
}

SDEV M4 m4_mul(const M4& a, const M4& b) {
  M4 c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) c.v[j * 4 + r] = a.v[r] * b.v[j * 4];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
#pragma unroll
      for (int r = 0; r < 4; ++r) c.v[j * 4 + r] = a.v[k * 4 + r] * b.v[j * 4 + k] + c.v[j * 4 + r];
    }
  }
  return c;
}

#ifdef STORM_FAST_MATH
// FMA build: products with the point matrices A_j = (dx/x) A(x_j, omega), whose entries (2,0), (2,1) and (0,3) are
// structurally zero (Rotating1D::eval / add_frequency never write them, src/system/adiabatic.rs:78-152).  ZA / ZB: bit
// c*4+r set = entry (r, c) of the left / right factor is a structural zero; the products with them are skipped (the parity
// build keeps them: 0*x is not 0 for x = inf/NaN, and x + 0 is not x for x = -0).
constexpr unsigned kPointMatrixZeros = (1u << (0 * 4 + 2)) | (1u << (1 * 4 + 2)) | (1u << (3 * 4 + 0));
template <unsigned ZA, unsigned ZB>
SDEV M4 m4_mul_sparse(const M4& a, const M4& b) {
  M4 c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = 0.0;
      bool first = true;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool zero = ((ZA >> (k * 4 + r)) & 1u) || ((ZB >> (j * 4 + k)) & 1u);
        if (!zero) {
          v = first ? a.v[k * 4 + r] * b.v[j * 4 + k] : a.v[k * 4 + r] * b.v[j * 4 + k] + v;
          first = false;
        }
      }
      c.v[j * 4 + r] = v;
    }
  }
  return c;
}
#endif

SDEV M4 m4_add(const M4& a, const M4& b) {
  M4 c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c.v[i] = a.v[i] + b.v[i];
  return c;
}
SDEV M4 m4_sub(const M4& a, const M4& b) {
  M4 c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c.v[i] = a.v[i] - b.v[i];
  return c;
}
SDEV M4 m4_scale(const M4& a, double s) {
  M4 c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c.v[i] = a.v[i] * s;
  return c;
}
SDEV M4 m4_neg(const M4& a) {
  M4 c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c.v[i] = -a.v[i];
  return c;
}
// I + a, I - a, a + I, a - I with the operand order of the expression they replace
// (addition commutes bitwise, subtraction does not).
SDEV M4 m4_eye_plus(const M4& a) {  // eye + a
  M4 c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c.v[i] = (((i % 5) == 0) ? 1.0 : 0.0) + a.v[i];
  return c;
}
SDEV M4 m4_eye_minus(const M4& a) {  // eye - a
  M4 c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c.v[i] = (((i % 5) == 0) ? 1.0 : 0.0) - a.v[i];
  return c;
}
SDEV M4 m4_minus_eye(const M4& a) {  // a - eye
  M4 c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c.v[i] = a.v[i] - (((i % 5) == 0) ? 1.0 : 0.0);
  return c;
}

#ifdef STORM_FAST_MATH
SDEV double fast_recip(double x);
// fast_recip with IEEE's special cases restored by one v_div_fixup_f64: 1/0 = inf, 1/inf = 0, 1/denormal (fast_recip alone
// gives NaN there: its seed is inf or 0 and the Newton step forms 0 * inf).  For operands the data can drive to the edge --
// add_frequency's omega_r^2 c1, which is 0 at omega = m Omega -- so that the FMA build's non-finite pattern is the parity
// build's (tests/test_hip_fast_mode.py::test_fast_zero_frequency_edge).
SDEV double fast_recip_guarded(double x) { return __builtin_amdgcn_div_fixup(fast_recip(x), x, 1.0); }
#endif

// Cofactor inverse.  `ok` is false when det == 0 (the reference would panic on
// try_inverse().unwrap()); the caller turns that into a NaN result + error flag.
SDEV M4 m4_inverse_ref(const M4& a, bool& ok);
#ifdef STORM_FAST_MATH
// FMA build: 1/x from the hardware seed (v_rcp_f64, ~2^-26) and two Newton steps: 6 instructions, against ~17 for the IEEE
// division sequence (v_div_scale x2, rcp, the fma chain, v_div_fmas, v_div_fixup).  Error <= ~1 ulp instead of 1/2 ulp; no
// rescue for zero, infinite, denormal or near-overflow x (the result is then NaN): used where the caller flags those itself
// (the determinant of an inverse: `ok`; the pivots of lu4_factor: its acceptance test); fast_recip_guarded elsewhere.
SDEV double fast_recip(double x) {
#ifdef STORM_FAST_IEEE_DIV  // (attribution experiment, tests/attribution_config3.py)
  return 1.0 / x;
#endif
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}

// FMA build: the same adjugate inverse, evaluated through the twelve 2x2 sub-determinants of the row pairs (0,1) and (2,3)
// (Laplace expansion; each 3x3 cofactor of the reference's formula is a 3-term combination of them), ~100 instructions
// instead of ~210.  Re-associates the cofactors, so it is not offered in the parity build.  Returns scale * a^-1 (the factor
// goes into the reciprocal of the determinant).
SDEV M4 m4_inverse_fast(const M4& a, bool& ok, double scale = 1.0) {
#ifdef STORM_FAST_REF_INVERSE  // (attribution experiment: the reference's sixteen cofactors, then the scale)
  return m4_scale(m4_inverse_ref(a, ok), scale);
#endif
  auto e = [&](int r, int c) { return a.v[c * 4 + r]; };
  const double s0 = e(0, 0) * e(1, 1) - e(1, 0) * e(0, 1), s1 = e(0, 0) * e(1, 2) - e(1, 0) * e(0, 2);
  const double s2 = e(0, 0) * e(1, 3) - e(1, 0) * e(0, 3), s3 = e(0, 1) * e(1, 2) - e(1, 1) * e(0, 2);
  const double s4 = e(0, 1) * e(1, 3) - e(1, 1) * e(0, 3), s5 = e(0, 2) * e(1, 3) - e(1, 2) * e(0, 3);
  const double c5 = e(2, 2) * e(3, 3) - e(3, 2) * e(2, 3), c4 = e(2, 1) * e(3, 3) - e(3, 1) * e(2, 3);
  const double c3 = e(2, 1) * e(3, 2) - e(3, 1) * e(2, 2), c2 = e(2, 0) * e(3, 3) - e(3, 0) * e(2, 3);
  const double c1 = e(2, 0) * e(3, 2) - e(3, 0) * e(2, 2), c0 = e(2, 0) * e(3, 1) - e(3, 0) * e(2, 1);
  const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
  ok = ok && (det != 0.0);
  const double q = scale * fast_recip(det);
  M4 o;
  auto put = [&](int r, int c, double v) { o.v[c * 4 + r] = v * q; };
  put(0, 0, e(1, 1) * c5 - e(1, 2) * c4 + e(1, 3) * c3);
  put(0, 1, -e(0, 1) * c5 + e(0, 2) * c4 - e(0, 3) * c3);
  put(0, 2, e(3, 1) * s5 - e(3, 2) * s4 + e(3, 3) * s3);
  put(0, 3, -e(2, 1) * s5 + e(2, 2) * s4 - e(2, 3) * s3);
  put(1, 0, -e(1, 0) * c5 + e(1, 2) * c2 - e(1, 3) * c1);
  put(1, 1, e(0, 0) * c5 - e(0, 2) * c2 + e(0, 3) * c1);
  put(1, 2, -e(3, 0) * s5 + e(3, 2) * s2 - e(3, 3) * s1);
  put(1, 3, e(2, 0) * s5 - e(2, 2) * s2 + e(2, 3) * s1);
  put(2, 0, e(1, 0) * c4 - e(1, 1) * c2 + e(1, 3) * c0);
  put(2, 1, -e(0, 0) * c4 + e(0, 1) * c2 - e(0, 3) * c0);
  put(2, 2, e(3, 0) * s4 - e(3, 1) * s2 + e(3, 3) * s0);
  put(2, 3, -e(2, 0) * s4 + e(2, 1) * s2 - e(2, 3) * s0);
  put(3, 0, -e(1, 0) * c3 + e(1, 1) * c1 - e(1, 2) * c0);
  put(3, 1, e(0, 0) * c3 - e(0, 1) * c1 + e(0, 2) * c0);
  put(3, 2, -e(3, 0) * s3 + e(3, 1) * s1 - e(3, 2) * s0);
  put(3, 3, e(2, 0) * s3 - e(2, 1) * s1 + e(2, 2) * s0);
  return o;
}

// FMA build: a = L U without row exchanges (L unit lower triangular; d = reciprocals of U's diagonal), ~45 instructions,
// and a^-1 b for one column in 16.  The stage matrices of the collocation schemes are I + O((dx/x)^2) and c S - I with
// S = O(dx/x): diagonally dominant everywhere but in the first few intervals next to the centre.  lu4_factor returns false
// unless every multiplier is <= 1 in magnitude -- the factorisation partial pivoting would have produced, element growth
// <= 8 -- and the caller then takes the adjugate inverse instead (also for a zero or NaN pivot: the comparisons fail).
struct Lu4 {
  double l10, l20, l30, l21, l31, l32;
  double u01, u02, u03, u12, u13, u23;
  double d0, d1, d2, d3;
};
SDEV bool lu4_factor(const M4& a, Lu4& f) {
  auto e = [&](int r, int c) { return a.v[c * 4 + r]; };
  f.d0 = fast_recip(e(0, 0));
  f.u01 = e(0, 1); f.u02 = e(0, 2); f.u03 = e(0, 3);
  f.l10 = e(1, 0) * f.d0; f.l20 = e(2, 0) * f.d0; f.l30 = e(3, 0) * f.d0;
  const double b11 = e(1, 1) - f.l10 * f.u01, b12 = e(1, 2) - f.l10 * f.u02, b13 = e(1, 3) - f.l10 * f.u03;
  const double b21 = e(2, 1) - f.l20 * f.u01, b22 = e(2, 2) - f.l20 * f.u02, b23 = e(2, 3) - f.l20 * f.u03;
  const double b31 = e(3, 1) - f.l30 * f.u01, b32 = e(3, 2) - f.l30 * f.u02, b33 = e(3, 3) - f.l30 * f.u03;
  f.d1 = fast_recip(b11);
  f.u12 = b12; f.u13 = b13;
  f.l21 = b21 * f.d1; f.l31 = b31 * f.d1;
  const double c22 = b22 - f.l21 * f.u12, c23 = b23 - f.l21 * f.u13;
  const double c32 = b32 - f.l31 * f.u12, c33 = b33 - f.l31 * f.u13;
  f.d2 = fast_recip(c22);
  f.u23 = c23;
  f.l32 = c32 * f.d2;
  const double e33 = c33 - f.l32 * f.u23;
  f.d3 = fast_recip(e33);
  return __builtin_fabs(f.l10) <= 1.0 && __builtin_fabs(f.l20) <= 1.0 && __builtin_fabs(f.l30) <= 1.0 &&
         __builtin_fabs(f.l21) <= 1.0 && __builtin_fabs(f.l31) <= 1.0 && __builtin_fabs(f.l32) <= 1.0 &&
         __builtin_fabs(f.d3) < __builtin_inf();
}
// x <- a^-1 x
SDEV void lu4_solve(const Lu4& f, double* x) {
  const double y0 = x[0];
  const double y1 = x[1] - f.l10 * y0;
  const double y2 = (x[2] - f.l20 * y0) - f.l21 * y1;
  const double y3 = ((x[3] - f.l30 * y0) - f.l31 * y1) - f.l32 * y2;
  const double x3 = y3 * f.d3;
  const double x2 = (y2 - f.u23 * x3) * f.d2;
  const double x1 = ((y1 - f.u12 * x2) - f.u13 * x3) * f.d1;
  const double x0 = (((y0 - f.u01 * x1) - f.u02 * x2) - f.u03 * x3) * f.d0;
  x[0] = x0; x[1] = x1; x[2] = x2; x[3] = x3;
}
// x[k] <- a^-1 x[k], falling back on the adjugate inverse where the factorisation would need row exchanges
template <int N>
SDEV void m4_solve_n(const M4& a, M4 (&x)[N], bool& ok) {
  Lu4 f;
#ifdef STORM_FAST_NO_LU  // (attribution experiment: always the adjugate inverse)
  if (false) {
#else
  if (lu4_factor(a, f)) {
#endif
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) lu4_solve(f, &x[k].v[j * 4]);
  } else {
    const M4 inv = m4_inverse_fast(a, ok);
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] = m4_mul(inv, x[k]);
  }
}
SDEV M4 m4_solve(const M4& a, const M4& b, bool& ok) {
  M4 x[1] = {b};
  m4_solve_n<1>(a, x, ok);
  return x[0];
}
#endif

SDEV M4 m4_inverse(const M4& a, bool& ok) {
#ifdef STORM_FAST_MATH
  return m4_inverse_fast(a, ok);
#else
  return m4_inverse_ref(a, ok);
#endif
}
SDEV M4 m4_inverse_ref(const M4& a, bool& ok) {
  const double* m = a.v;
  M4 o;
  o.v[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
           m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  o.v[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
           m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  o.v[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
           m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  o.v[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
            m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  o.v[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
           m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  o.v[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
           m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  o.v[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
           m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  o.v[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
            m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  o.v[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
           m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  o.v[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
           m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  o.v[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
            m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  o.v[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
            m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  o.v[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
           m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  o.v[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
           m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  o.v[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
            m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  o.v[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
            m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const double det = m[0] * o.v[0] + m[1] * o.v[4] + m[2] * o.v[8] + m[3] * o.v[12];
  ok = ok && (det != 0.0);
  const double inv_det = 1.0 / det;
#pragma unroll
  for (int i = 0; i < 16; ++i) o.v[i] = o.v[i] * inv_det;
  return o;
}

// ceil(log2(x)) from the IEEE fields, the way the reference derives its squaring count
// (integer_decode + next_power_of_two().ilog2(), reference src/linalg/storage.rs:418-421).
SDEV int exp_scaling_exponent(double norm) {
  const uint64_t bits = (uint64_t)__double_as_longlong(norm);
  const int biased = (int)((bits >> 52) & 0x7ff);
  const uint64_t frac = bits & 0xfffffffffffffULL;
  const uint64_t mantissa = biased == 0 ? (frac << 1) : (frac | 0x10000000000000ULL);
  int lg = 0;
  if (mantissa > 1) lg = 64 - __clzll((long long)(mantissa - 1));
  return biased - 1075 + lg;
}

// exp(A) by scaling and squaring with the 18-term Taylor polynomial of Bader, Blanes &
// Casas evaluated with 5 products (reference src/linalg/storage.rs:402-470).  The
// polynomial terms are accumulated left to right including the zero coefficients.
// SPARSE_IN (FMA build only): `a` has the point matrices' zero pattern (Magnus2: Omega is the midpoint matrix itself), so
// the first two products skip it.
template <bool SPARSE_IN = false>
SDEV M4 m4_exp(M4 a) {
  constexpr double T[5][5] = {
      {0., -0.10036558103014462, -0.00802924648241157, -0.00089213849804573, 0.},
      {0., 0.3978497494996451, 1.3678377846041172, 0.49828962252538267, -0.0006378981945947233},
      {-10.967639605296206, 1.680158138789062, 0.05717798464788655, -0.0069821012248805206,
       0.00003349750170860705},
      {-0.09043168323908106, -0.06764045190713819, 0.06759613017704597, 0.029555257042931552,
       -0.00001391802575160607},
      {0., 0., -0.09233646193671186, -0.016936493900208172, -0.00001400867981820361}};
  double norm = 0.0;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) s = s + __builtin_fabs(a.v[c * 4 + r]);
    norm = (c == 0) ? s : ((s > norm || norm != norm) ? s : norm);
  }
  const int exponent = exp_scaling_exponent(norm);
  if (exponent > 0) {
    // 2^-exponent, exact (subnormal for absurd norms, like ldexp)
    const double sc = __builtin_ldexp(1.0, -exponent);
    a = m4_scale(a, sc);
  }
#ifdef STORM_FAST_MATH
  constexpr unsigned kZ = SPARSE_IN ? kPointMatrixZeros : 0u;
  const M4 a2 = m4_mul_sparse<kZ, kZ>(a, a);
  const M4 a3 = m4_mul_sparse<0u, kZ>(a2, a);
#else
  const M4 a2 = m4_mul(a, a);
  const M4 a3 = m4_mul(a2, a);
#endif
  const M4 a6 = m4_mul(a3, a3);
  // b_i = T[i][0] I + T[i][1] A + T[i][2] A^2 + T[i][3] A^3 + T[i][4] A^6, formed where it is consumed (five of them at
  // once are 160 registers)
  auto combo = [&](int i) {
    M4 b;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const double id = ((e % 5) == 0) ? 1.0 : 0.0;
      b.v[e] = (((id * T[i][0] + a.v[e] * T[i][1]) + a2.v[e] * T[i][2]) + a3.v[e] * T[i][3]) + a6.v[e] * T[i][4];
    }
    return b;
  };
  const M4 a9 = m4_add(m4_mul(combo(0), combo(4)), combo(3));
  const M4 p9 = m4_mul(m4_add(combo(2), a9), a9);
  M4 r = m4_add(combo(1), p9);
  for (int i = 0; i < exponent; ++i) r = m4_mul(r, r);
  return r;
}

SDEV M4 m4_commutator(const M4& a, const M4& b) { return m4_sub(m4_mul(a, b), m4_mul(b, a)); }

}  // namespace STORM_NS
