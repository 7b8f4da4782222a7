// steppers.hpp -- per-interval transfer ("step") matrices on the device.
//
// step_matrix<SCHEME>(a[P]) turns the P collocation-point matrices A_j = (dx/x_j) A(x_j, omega)
// of one grid interval into the explicit transfer matrix L with y_{i+1} = L y_i:
//   * Gauss-Legendre collocation of order 2/4/6/8: block elimination of the stage system gives
//     `jump`; the implicit relation (jump + I) y_i + (jump - I) y_{i+1} = 0 is made explicit as
//     L = (-(jump - I)^-1) (jump + I)          (reference src/stepper.rs:64-90,138-348)
//   * Magnus 2/4/6/8: L = exp(Omega) with Omega from nested commutators of the samples
//                                               (reference src/stepper.rs:350-525)
// Expression association follows the reference exactly (see mat4.hpp for the contract).
#pragma once
#include "mat4.hpp"

namespace STORM_NS {
using namespace storm;

// Gauss-Legendre nodes on [-1/2, 1/2] (reference src/stepper.rs:94-105).
constexpr double kC2_1 = -0.28867513459481287, kC2_2 = 0.28867513459481287;
constexpr double kC3_1 = -0.3872983346207417, kC3_3 = 0.3872983346207417;
constexpr double kC4_1 = -0.8611363115940526 / 2., kC4_2 = -0.33998104358485626 / 2.;
constexpr double kC4_3 = 0.33998104358485626 / 2., kC4_4 = 0.8611363115940526 / 2.;
// Order-8 collocation tableau, c_ij - b_j/2 (reference src/stepper.rs:272-286).
constexpr double kD12 = 1.89640468800635328970119143171e-1, kD13 = 1.5040882602623181114167713212e-1;
constexpr double kD14 = 9.05188609701591475001769691551e-2, kD21 = -1.01154406215504607307419557782e-1;
constexpr double kD23 = 1.90916717318107430880885119114e-1, kD24 = 8.02282106898253088278673182198e-2;
constexpr double kD31 = -8.02282106898253088278673182198e-2, kD32 = -1.90916717318107430880885119114e-1;
constexpr double kD34 = 1.01154406215504607307419557782e-1, kD41 = -9.05188609701591475001769691551e-2;
constexpr double kD42 = -1.5040882602623181114167713212e-1, kD43 = -1.89640468800635328970119143171e-1;
// Order-8 weights and Magnus-8 combination constants (reference src/stepper.rs:446-459).
constexpr double kW1 = 0.34785484513745385 / 2., kW2 = 0.6521451548625461 / 2.;
constexpr double kB1A1S = 9. / 4. * kW1 - 15. * kW1 * kC4_1 * kC4_1;
constexpr double kB1A2S = 9. / 4. * kW2 - 15. * kW2 * kC4_2 * kC4_2;
constexpr double kB2A1A = 75. * kW1 * kC4_4 - 420. * kW1 * kC4_4 * kC4_4 * kC4_4;
constexpr double kB2A2A = 75. * kW2 * kC4_3 - 420. * kW2 * kC4_3 * kC4_3 * kC4_3;
constexpr double kB3A1S = -15. * kW1 + 180. * kW1 * kC4_1 * kC4_1;
constexpr double kB4A1A = -420. * kW1 * kC4_4 + 2800. * kW1 * kC4_4 * kC4_4 * kC4_4;
constexpr double kB4A2A = -420. * kW2 * kC4_3 + 2800. * kW2 * kC4_3 * kC4_3 * kC4_3;

// Collocation abscissae inside an interval, as fractions of its width.
inline void scheme_abscissae(int scheme, double out[4]) {
  switch (scheme_points(scheme)) {
    case 1: out[0] = 0.5; break;
    case 2: out[0] = 0.5 + kC2_1; out[1] = 0.5 + kC2_2; break;
    case 3: out[0] = 0.5 + kC3_1; out[1] = 0.5 + 0.; out[2] = 0.5 + kC3_3; break;
    default: out[0] = 0.5 + kC4_1; out[1] = 0.5 + kC4_2; out[2] = 0.5 + kC4_3; out[3] = 0.5 + kC4_4; break;
  }
}

// Products whose factors are point matrices: a_i * a_j and a_i * X.
#ifdef STORM_FAST_MATH
SDEV M4 m4_mul_aa(const M4& a, const M4& b) { return m4_mul_sparse<kPointMatrixZeros, kPointMatrixZeros>(a, b); }
SDEV M4 m4_mul_ax(const M4& a, const M4& x) { return m4_mul_sparse<kPointMatrixZeros, 0u>(a, x); }
#else
SDEV M4 m4_mul_aa(const M4& a, const M4& b) { return m4_mul(a, b); }
SDEV M4 m4_mul_ax(const M4& a, const M4& x) { return m4_mul(a, x); }
#endif

// FMA build: colloc_jump returns the weighted stage sum WITHOUT its common factor c (jump = c * S), and step_matrix folds c
// into the scaled inverse: (c S - I)^-1 = (S - I/c)^-1 / c.  Saves one matrix scaling per interval.
template <int SCHEME>
constexpr double colloc_jump_factor() {
  return SCHEME == kColloc2 ? 0.5 : SCHEME == kColloc4 ? 0.25 : SCHEME == kColloc6 ? 5. / 36. : kW1 / 2.;
}

template <int SCHEME>
SDEV M4 colloc_jump(const M4* a, bool& ok) {
  if constexpr (SCHEME == kColloc2) {
#ifdef STORM_FAST_MATH
    return a[0];
#endif
    return m4_scale(a[0], 0.5);
  } else if constexpr (SCHEME == kColloc4) {
    const M4 a2a1 = m4_mul_aa(a[1], a[0]);
#ifdef STORM_FAST_MATH
    // k2 = (I + a2a1/12)^-1 rhs = 12 z with (12 I + a2a1) z = rhs, solved by LU; k1 + k2 = a1 + 12 (z + c1 a1 z)
    M4 st = a2a1;
    st.v[0] += 12.0; st.v[5] += 12.0; st.v[10] += 12.0; st.v[15] += 12.0;
    const M4 z = m4_solve(st, m4_add(a[1], m4_scale(a2a1, kC2_2)), ok);
    const M4 w = m4_add(z, m4_scale(m4_mul_ax(a[0], z), kC2_1));
    return m4_add(a[0], m4_scale(w, 12.0));
#else
    const M4 k2 = m4_mul(m4_inverse(m4_eye_plus(m4_scale(a2a1, 1. / 12.)), ok), m4_add(a[1], m4_scale(a2a1, kC2_2)));
    const M4 k1 = m4_add(a[0], m4_scale(m4_mul_ax(a[0], k2), kC2_1));
    return m4_scale(m4_add(k1, k2), 0.25);
#endif
  } else if constexpr (SCHEME == kColloc6) {
    constexpr double d12 = -2. / 3. * kC3_1, d13 = -1. / 3. * kC3_1, d21 = 5. / 12. * kC3_1;
    constexpr double d23 = -5. / 12. * kC3_1, d31 = 1. / 3. * kC3_1, d32 = 2. / 3. * kC3_1;
    const M4 a2a1 = m4_mul_aa(a[1], a[0]);
    const M4 a3a1 = m4_mul_aa(a[2], a[0]);
    // (FMA build: solving this first system, two right-hand sides, by LU as well takes det_hex past 256 registers, its
    // second wave per SIMD, and costs more than it saves)
    const M4 inv1 = m4_inverse(m4_eye_minus(m4_scale(a2a1, d12 * d21)), ok);
    const M4 o2 = m4_mul(inv1, m4_sub(a[1], m4_scale(a2a1, d21)));
    const M4 l2 = m4_mul(inv1, m4_sub(m4_scale(a[1], d23), m4_scale(a2a1, d13 * d21)));
    const M4 f3 = m4_sub(m4_scale(a[2], d32), m4_scale(a3a1, d12 * d31));
    const M4 o3 = m4_sub(m4_sub(a[2], m4_scale(a3a1, d31)), m4_mul(f3, o2));
    const M4 l3 = m4_sub(m4_eye_minus(m4_scale(a3a1, d13 * d31)), m4_mul(f3, l2));
#ifdef STORM_FAST_MATH
    const M4 k3 = m4_solve(l3, o3, ok);
#else
    const M4 k3 = m4_mul(m4_inverse(l3, ok), o3);
#endif
    const M4 k2 = m4_sub(o2, m4_mul(l2, k3));
    const M4 k1 = m4_mul_ax(a[0], m4_sub(m4_eye_minus(m4_scale(k2, d12)), m4_scale(k3, d13)));
#ifdef STORM_FAST_MATH
    return m4_add(m4_add(k1, k3), m4_scale(k2, (2. / 9.) / (5. / 36.)));
#endif
    return m4_add(m4_add(m4_scale(k1, 5. / 36.), m4_scale(k2, 2. / 9.)), m4_scale(k3, 5. / 36.));
  } else {
    static_assert(SCHEME == kColloc8, "collocation scheme expected");
    const M4 a2a1 = m4_mul_aa(a[1], a[0]), a3a1 = m4_mul_aa(a[2], a[0]), a4a1 = m4_mul_aa(a[3], a[0]);
#ifdef STORM_FAST_MATH
    // the stage systems are solved by LU (m4_solve_n) instead of inverse + product
    M4 s2[3] = {m4_sub(m4_scale(a[1], kD23), m4_scale(a2a1, kD21 * kD13)),
                m4_sub(m4_scale(a[1], kD24), m4_scale(a2a1, kD21 * kD14)), m4_sub(a[1], m4_scale(a2a1, kD21))};
    m4_solve_n<3>(m4_eye_minus(m4_scale(a2a1, kD21 * kD12)), s2, ok);
    const M4 u23 = s2[0], u24 = s2[1], r2 = s2[2];
    const M4 l32 = m4_sub(m4_scale(a[2], kD32), m4_scale(a3a1, kD31 * kD12));
    M4 s3[2] = {m4_sub(m4_sub(m4_scale(a[2], kD34), m4_scale(a3a1, kD31 * kD14)), m4_mul(l32, u24)),
                m4_sub(m4_sub(a[2], m4_scale(a3a1, kD31)), m4_mul(l32, r2))};
    m4_solve_n<2>(m4_sub(m4_eye_minus(m4_scale(a3a1, kD31 * kD13)), m4_mul(l32, u23)), s3, ok);
    const M4 u34 = s3[0], r3 = s3[1];
    const M4 l42 = m4_sub(m4_scale(a[3], kD42), m4_scale(a4a1, kD41 * kD12));
    const M4 l43 = m4_sub(m4_sub(m4_scale(a[3], kD43), m4_scale(a4a1, kD41 * kD13)), m4_mul(l42, u23));
    const M4 k4 = m4_solve(m4_sub(m4_sub(m4_eye_minus(m4_scale(a4a1, kD41 * kD14)), m4_mul(l42, u24)), m4_mul(l43, u34)),
                           m4_sub(m4_sub(m4_sub(a[3], m4_scale(a4a1, kD41)), m4_mul(l42, r2)), m4_mul(l43, r3)), ok);
#else
    const M4 inv2 = m4_inverse(m4_eye_minus(m4_scale(a2a1, kD21 * kD12)), ok);
    const M4 u23 = m4_mul(inv2, m4_sub(m4_scale(a[1], kD23), m4_scale(a2a1, kD21 * kD13)));
    const M4 u24 = m4_mul(inv2, m4_sub(m4_scale(a[1], kD24), m4_scale(a2a1, kD21 * kD14)));
    const M4 r2 = m4_mul(inv2, m4_sub(a[1], m4_scale(a2a1, kD21)));
    const M4 l32 = m4_sub(m4_scale(a[2], kD32), m4_scale(a3a1, kD31 * kD12));
    const M4 inv3 = m4_inverse(m4_sub(m4_eye_minus(m4_scale(a3a1, kD31 * kD13)), m4_mul(l32, u23)), ok);
    const M4 u34 = m4_mul(inv3, m4_sub(m4_sub(m4_scale(a[2], kD34), m4_scale(a3a1, kD31 * kD14)), m4_mul(l32, u24)));
    const M4 r3 = m4_mul(inv3, m4_sub(m4_sub(a[2], m4_scale(a3a1, kD31)), m4_mul(l32, r2)));
    const M4 l42 = m4_sub(m4_scale(a[3], kD42), m4_scale(a4a1, kD41 * kD12));
    const M4 l43 = m4_sub(m4_sub(m4_scale(a[3], kD43), m4_scale(a4a1, kD41 * kD13)), m4_mul(l42, u23));
    const M4 inv4 = m4_inverse(
        m4_sub(m4_sub(m4_eye_minus(m4_scale(a4a1, kD41 * kD14)), m4_mul(l42, u24)), m4_mul(l43, u34)), ok);
    const M4 k4 = m4_mul(inv4, m4_sub(m4_sub(m4_sub(a[3], m4_scale(a4a1, kD41)), m4_mul(l42, r2)), m4_mul(l43, r3)));
#endif
    const M4 k3 = m4_sub(r3, m4_mul(u34, k4));
    const M4 k2 = m4_sub(m4_sub(r2, m4_mul(u24, k4)), m4_mul(u23, k3));
    const M4 k1 = m4_sub(m4_sub(m4_sub(a[0], m4_mul_ax(m4_scale(a[0], kD14), k4)), m4_mul_ax(m4_scale(a[0], kD13), k3)),
                         m4_mul_ax(m4_scale(a[0], kD12), k2));
#ifdef STORM_FAST_MATH
    return m4_add(m4_add(k1, k4), m4_scale(m4_add(k2, k3), kW2 / kW1));
#endif
    return m4_add(m4_add(m4_add(m4_scale(k1, kW1 / 2.), m4_scale(k2, kW2 / 2.)), m4_scale(k3, kW2 / 2.)),
                  m4_scale(k4, kW1 / 2.));
  }
}

// Commutators whose factors are (combinations of) point matrices: the products skip the structural zeros in the FMA build.
// SA / SB: the left / right factor of [a, b] has the point matrices' zero pattern.
template <bool SA, bool SB>
SDEV M4 magnus_commutator(const M4& a, const M4& b) {
#ifdef STORM_FAST_MATH
  constexpr unsigned ZA = SA ? kPointMatrixZeros : 0u, ZB = SB ? kPointMatrixZeros : 0u;
  return m4_sub(m4_mul_sparse<ZA, ZB>(a, b), m4_mul_sparse<ZB, ZA>(b, a));
#else
  return m4_commutator(a, b);
#endif
}

template <int SCHEME>
SDEV M4 magnus_omega(const M4* a) {
  if constexpr (SCHEME == kMagnus2) {
    return a[0];
  } else if constexpr (SCHEME == kMagnus4) {
    const M4 b1 = m4_scale(m4_add(a[0], a[1]), 0.5);
    const M4 b2 = m4_scale(m4_sub(a[1], a[0]), 0.5 * kC2_2);
    return m4_sub(b1, magnus_commutator<true, true>(b1, b2));
  } else if constexpr (SCHEME == kMagnus6) {
    constexpr double ten_three = 10. / 3.;
    const M4& b1 = a[1];
    const M4 b2 = m4_scale(m4_sub(a[2], a[0]), kC3_3 * ten_three);
    const M4 b3 = m4_scale(m4_add(m4_sub(a[2], m4_scale(a[1], 2.)), a[0]), ten_three);
    const M4 c1 = magnus_commutator<true, true>(b1, b2);
    const M4 c2 = m4_scale(magnus_commutator<true, false>(b1, m4_add(m4_scale(b3, 2.), c1)), -1. / 60.);
    return m4_add(m4_add(b1, m4_scale(b3, 1. / 12.)),
                  m4_scale(m4_commutator(m4_add(m4_sub(m4_scale(b1, -20.), b3), c1), m4_add(b2, c2)), 1. / 240.));
  } else {
    static_assert(SCHEME == kMagnus8, "magnus scheme expected");
    const M4 a1s = m4_add(a[0], a[3]), a1a = m4_sub(a[3], a[0]);
    const M4 a2s = m4_add(a[1], a[2]), a2a = m4_sub(a[2], a[1]);
    const M4 b1 = m4_add(m4_scale(a1s, kB1A1S), m4_scale(a2s, kB1A2S));
    const M4 b2 = m4_add(m4_scale(a1a, kB2A1A), m4_scale(a2a, kB2A2A));
    const M4 b3 = m4_scale(m4_sub(a1s, a2s), kB3A1S);
    const M4 b4 = m4_add(m4_scale(a1a, kB4A1A), m4_scale(a2a, kB4A2A));
    const M4 b1p = m4_add(b1, m4_scale(b3, 1. / 28.));
    const M4 b2p = m4_add(b2, m4_scale(b4, 3. / 28.));
    const M4 s1 = m4_scale(magnus_commutator<true, true>(b1p, b2p), -1. / 28.);
    const M4 r1 = m4_scale(magnus_commutator<true, false>(b1, m4_add(m4_scale(b3, -1. / 14.), s1)), 1. / 3.);
    const M4 s2 = m4_commutator(m4_add(b1p, s1), m4_add(b2p, r1));
    const M4 s2p = magnus_commutator<true, false>(b2, s1);
    const M4 r2 = m4_commutator(m4_add(b1, m4_scale(s1, 5. / 4.)),
                                m4_add(m4_add(m4_scale(b3, 2.0), s2), m4_scale(s2p, 0.5)));
    const M4 base = m4_add(b1, m4_scale(b3, 1. / 12.));
    const M4 s3 = m4_commutator(
        m4_add(m4_add(base, m4_scale(s1, -7. / 3.)), m4_scale(s2, -1. / 6.)),
        m4_add(m4_add(m4_add(m4_scale(b2, -9.), m4_scale(b4, -9. / 4.)), m4_scale(r1, 63.)), r2));
    return m4_add(m4_add(base, m4_scale(s2, -7. / 120.)), m4_scale(s3, 1. / 360.));
  }
}

#ifdef STORM_FAST_MATH
// FMA build, collocation schemes: with jump = c S the transfer matrix is (-right^-1) * left with right = c S - I and
// left = right + 2 I, i.e. -I - 2 right^-1 = -I - (2/c) (S - I/c)^-1: no product (same matrix, re-associated).
// colloc_shifted returns S - I/c, step_from_shifted inverts it.  (Factoring S - I/c and applying the solve to the columns
// of det_col's running product instead was measured: 150 fewer instructions per interval and no faster -- the 20-deep
// dependent chain per step and four more v_rcp_f64 cost what the adjugate inverse did.  DESIGN.md section 12.)
template <int SCHEME>
SDEV M4 colloc_shifted(const M4* a, bool& ok) {
  constexpr double c = colloc_jump_factor<SCHEME>();
  M4 r = colloc_jump<SCHEME>(a, ok);
  r.v[0] -= 1.0 / c; r.v[5] -= 1.0 / c; r.v[10] -= 1.0 / c; r.v[15] -= 1.0 / c;
  return r;
}
template <int SCHEME>
SDEV M4 step_from_shifted(const M4& r, bool& ok) {
  constexpr double c = colloc_jump_factor<SCHEME>();
  M4 t = m4_inverse_fast(r, ok, -2.0 / c);
  t.v[0] = t.v[0] - 1.0; t.v[5] = t.v[5] - 1.0; t.v[10] = t.v[10] - 1.0; t.v[15] = t.v[15] - 1.0;
  return t;
}
#endif

// The transfer matrix of one interval.  `ok` goes false if a collocation stage matrix is
// singular (the reference panics there).
template <int SCHEME>
SDEV M4 step_matrix(const M4* a, bool& ok) {
  if constexpr (scheme_is_magnus(SCHEME)) {
    return m4_exp<SCHEME == kMagnus2>(magnus_omega<SCHEME>(a));
  } else {
#if defined(STORM_FAST_MATH) && !defined(STORM_FAST_NO_FOLD)
    return step_from_shifted<SCHEME>(colloc_shifted<SCHEME>(a, ok), ok);
#else
    M4 jump = colloc_jump<SCHEME>(a, ok);
#ifdef STORM_FAST_MATH  // (attribution experiment: the wrapper's product form on jump = c S)
    jump = m4_scale(jump, colloc_jump_factor<SCHEME>());
#endif
    M4 l, r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const double id = ((i % 5) == 0) ? 1.0 : 0.0;
      l.v[i] = jump.v[i] + id;
      r.v[i] = jump.v[i] - id;
    }
    return m4_mul(m4_neg(m4_inverse(r, ok)), l);
#endif
  }
}

}  // namespace STORM_NS
