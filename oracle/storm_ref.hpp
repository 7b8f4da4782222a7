// storm_ref.hpp -- CPU ORACLE (test infrastructure, NOT the product).
//
// A plain C++17 restatement of the frequency-scan hot path of StORM
// (VincentVanlaer/StORM @ 2025-10-31), written from reading the Rust sources and
// keeping their exact operation order so that results are bit-identical to the
// reference when built with  g++ -O2 -ffp-contract=off  (no FMA contraction, no
// fast-math).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use this file; the product (storm_amd/) never includes or links it.
//
// Parity pinning: tests/test_oracle_golden.py checks this file against the
// reference's own known answers (152 golden roots of
// src/dynamic_interface.rs:268-503, the exp KAT of src/linalg/storage.rs:498-520,
// test_polytrope, and the mode counts of src/postprocessing.rs:424-498).
// Unpinned by any reference test: rot != 0 terms, raw det values.
//
// Third-party arithmetic that is not under /root/reference (nalgebra =0.33.2,
// default-features=false): matrix product = column-wise gemv/axpy, k ascending,
// separate mul and add; 4x4 try_inverse = closed-form cofactor expansion followed
// by a multiply with 1/det.  Restated here from the published algorithm.
//
// Every function cites the reference file:line it follows (paths relative to the
// reference root).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace storm_ref {

// ---------------------------------------------------------------------------
// Op-counting scalar: lets the oracle freeze the exact algorithmic flop count of
// one (frequency, interval) evaluation.  add/sub/mul/div = 1 flop,
// negation, abs and comparisons = 0.
// ---------------------------------------------------------------------------
struct Counted {
  double v;
  static inline thread_local uint64_t n_add = 0, n_mul = 0, n_div = 0;
  Counted() : v(0.) {}
  Counted(double x) : v(x) {}
  static void reset() { n_add = n_mul = n_div = 0; }
  static uint64_t total() { return n_add + n_mul + n_div; }
};
inline Counted operator+(Counted a, Counted b) { ++Counted::n_add; return Counted(a.v + b.v); }
inline Counted operator-(Counted a, Counted b) { ++Counted::n_add; return Counted(a.v - b.v); }
inline Counted operator*(Counted a, Counted b) { ++Counted::n_mul; return Counted(a.v * b.v); }
inline Counted operator/(Counted a, Counted b) { ++Counted::n_div; return Counted(a.v / b.v); }
inline Counted operator-(Counted a) { return Counted(-a.v); }
inline Counted& operator+=(Counted& a, Counted b) { a = a + b; return a; }
inline Counted& operator-=(Counted& a, Counted b) { a = a - b; return a; }
inline Counted& operator*=(Counted& a, Counted b) { a = a * b; return a; }
inline bool operator!=(Counted a, Counted b) { return a.v != b.v; }
inline bool operator==(Counted a, Counted b) { return a.v == b.v; }
inline double to_double(double x) { return x; }
inline double to_double(Counted x) { return x.v; }
template <class T> inline double absval(T x) { return std::fabs(to_double(x)); }

// ---------------------------------------------------------------------------
// Small dense matrices, column-major like nalgebra (element (r,c) at c*N + r).
// ---------------------------------------------------------------------------
template <class T, int N>
struct Mat {
  T a[N * N];
  T& operator()(int r, int c) { return a[c * N + r]; }
  const T& operator()(int r, int c) const { return a[c * N + r]; }
  static Mat zeros() {
    Mat m;
    for (int i = 0; i < N * N; ++i) m.a[i] = T(0.);
    return m;
  }
  static Mat identity() {
    Mat m = zeros();
    for (int i = 0; i < N; ++i) m(i, i) = T(1.);
    return m;
  }
};

// nalgebra `a * b` without matrixmultiply: for every column j of the result,
// gemv = axpy over the columns of `a`, k ascending; first term is a pure product.
template <class T, int N>
inline Mat<T, N> mul(const Mat<T, N>& A, const Mat<T, N>& B) {
  Mat<T, N> C;
  for (int j = 0; j < N; ++j) {
    for (int r = 0; r < N; ++r) C(r, j) = A(r, 0) * B(0, j);
    for (int k = 1; k < N; ++k)
      for (int r = 0; r < N; ++r) C(r, j) = A(r, k) * B(k, j) + C(r, j);
  }
  return C;
}
template <class T, int N>
inline Mat<T, N> add(const Mat<T, N>& A, const Mat<T, N>& B) {
  Mat<T, N> C;
  for (int i = 0; i < N * N; ++i) C.a[i] = A.a[i] + B.a[i];
  return C;
}
template <class T, int N>
inline Mat<T, N> sub(const Mat<T, N>& A, const Mat<T, N>& B) {
  Mat<T, N> C;
  for (int i = 0; i < N * N; ++i) C.a[i] = A.a[i] - B.a[i];
  return C;
}
template <class T, int N>
inline Mat<T, N> scale(const Mat<T, N>& A, double s) {
  Mat<T, N> C;
  for (int i = 0; i < N * N; ++i) C.a[i] = A.a[i] * T(s);
  return C;
}
template <class T, int N>
inline Mat<T, N> neg(const Mat<T, N>& A) {
  Mat<T, N> C;
  for (int i = 0; i < N * N; ++i) C.a[i] = -A.a[i];
  return C;
}

// nalgebra try_inverse for n = 4: cofactor expansion on the column-major linear
// entries, det from the first column of cofactors, then scale by 1/det.
// Call sites: src/stepper.rs:88,199,250,259,319,327,335.
template <class T>
inline bool inv4(const Mat<T, 4>& M, Mat<T, 4>& out) {
  const T* m = M.a;
  T o[16];
  o[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] +
         m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  o[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] -
         m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  o[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] +
         m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  o[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] -
          m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  T det = m[0] * o[0] + m[1] * o[4] + m[2] * o[8] + m[3] * o[12];
  if (det == T(0.)) return false;
  o[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] -
         m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  o[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] +
         m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  o[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] -
         m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  o[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] +
          m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  o[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] +
         m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  o[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] -
         m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  o[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] +
          m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  o[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] -
          m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  o[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] -
         m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  o[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] +
         m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  o[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] -
          m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  o[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] +
          m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  T inv_det = T(1.) / det;
  for (int i = 0; i < 16; ++i) out.a[i] = o[i] * inv_det;
  return true;
}
template <class T>
inline Mat<T, 4> inv4_unwrap(const Mat<T, 4>& M) {
  Mat<T, 4> o;
  if (!inv4(M, o)) throw std::runtime_error("singular matrix in try_inverse().unwrap()");
  return o;
}

// ---------------------------------------------------------------------------
// Matrix exponential, T18 scaling and squaring.  src/linalg/storage.rs:321-357
// (coefficients) and :402-470 (algorithm).
// ---------------------------------------------------------------------------
static const double T18[5][5] = {
    {0., -0.10036558103014462, -0.00802924648241157, -0.00089213849804573, 0.},
    {0., 0.3978497494996451, 1.3678377846041172, 0.49828962252538267, -0.0006378981945947233},
    {-10.967639605296206, 1.680158138789062, 0.05717798464788655, -0.0069821012248805206,
     0.00003349750170860705},
    {-0.09043168323908106, -0.06764045190713819, 0.06759613017704597, 0.029555257042931552,
     -0.00001391802575160607},
    {0., 0., -0.09233646193671186, -0.016936493900208172, -0.00001400867981820361}};

// exponent = integer_decode().1 + ilog2(next_power_of_two(mantissa))
// (src/linalg/storage.rs:418-421), i.e. ceil(log2(norm)) for normal numbers.
inline int exp_scaling_exponent(double norm) {
  uint64_t bits;
  std::memcpy(&bits, &norm, 8);
  int biased = int((bits >> 52) & 0x7ff);
  uint64_t mant = bits & 0xfffffffffffffULL;
  uint64_t mantissa = biased == 0 ? (mant << 1) : (mant | 0x10000000000000ULL);
  int exponent = biased - 1075;
  // next_power_of_two(0) == 1; ilog2 of the rounded-up power of two.
  int lg = 0;
  if (mantissa > 1) {
    lg = 64 - __builtin_clzll(mantissa - 1);
  }
  return exponent + lg;
}

template <class T, int N>
inline void mat_exp(Mat<T, N>& A) {
  // 1-norm: per column left-to-right sum of abs, then max (storage.rs:405-416).
  double norm = 0.;
  for (int c = 0; c < N; ++c) {
    double s = 0.;
    for (int r = 0; r < N; ++r) s = s + absval(A(r, c));
    if (c == 0) norm = s;
    else norm = (s > norm || norm != norm) ? s : norm;  // RealField::max; NaN never occurs for finite input
  }
  int exponent = exp_scaling_exponent(norm);
  if (exponent > 0) A = scale(A, std::ldexp(1.0, -exponent));  // storage.rs:423-425
  Mat<T, N> eye = Mat<T, N>::identity();
  Mat<T, N> a2 = mul(A, A);
  Mat<T, N> a3 = mul(a2, A);
  Mat<T, N> a6 = mul(a3, a3);
  Mat<T, N> b[5];
  for (int i = 0; i < 5; ++i) {  // storage.rs:435-459, zero coefficients included
    b[i] = add(add(add(add(scale(eye, T18[i][0]), scale(A, T18[i][1])), scale(a2, T18[i][2])),
                   scale(a3, T18[i][3])),
               scale(a6, T18[i][4]));
  }
  Mat<T, N> a9 = add(mul(b[0], b[4]), b[3]);    // storage.rs:461
  A = add(b[1], mul(add(b[2], a9), a9));        // storage.rs:463
  if (exponent > 0)
    for (int i = 0; i < exponent; ++i) A = mul(A, A);  // storage.rs:465-469
}

template <class T, int N>
inline Mat<T, N> commutator(const Mat<T, N>& m1, const Mat<T, N>& m2) {  // storage.rs:473-483
  return sub(mul(m1, m2), mul(m2, m1));
}

// ---------------------------------------------------------------------------
// Steppers.  src/stepper.rs
// ---------------------------------------------------------------------------
enum Scheme { COLLOC2 = 0, COLLOC4, COLLOC6, COLLOC8, MAGNUS2, MAGNUS4, MAGNUS6, MAGNUS8 };

// stepper.rs:94-105
static const double C2_1 = -0.28867513459481287, C2_2 = 0.28867513459481287;
static const double C3_1 = -0.3872983346207417, C3_2 = 0., C3_3 = 0.3872983346207417;
static const double C4_1 = -0.8611363115940526 / 2., C4_2 = -0.33998104358485626 / 2.,
                    C4_3 = 0.33998104358485626 / 2., C4_4 = 0.8611363115940526 / 2.;
// stepper.rs:272-286
static const double D4_12 = 1.89640468800635328970119143171e-1, D4_13 = 1.5040882602623181114167713212e-1,
                    D4_14 = 9.05188609701591475001769691551e-2;
static const double D4_21 = -1.01154406215504607307419557782e-1, D4_23 = 1.90916717318107430880885119114e-1,
                    D4_24 = 8.02282106898253088278673182198e-2;
static const double D4_31 = -8.02282106898253088278673182198e-2, D4_32 = -1.90916717318107430880885119114e-1,
                    D4_34 = 1.01154406215504607307419557782e-1;
static const double D4_41 = -9.05188609701591475001769691551e-2, D4_42 = -1.5040882602623181114167713212e-1,
                    D4_43 = -1.89640468800635328970119143171e-1;
// stepper.rs:446-459
static const double W1 = 0.34785484513745385 / 2., W2 = 0.6521451548625461 / 2.;
static const double B1_A1S = 9. / 4. * W1 - 15. * W1 * C4_1 * C4_1;
static const double B1_A2S = 9. / 4. * W2 - 15. * W2 * C4_2 * C4_2;
static const double B2_A1A = 75. * W1 * C4_4 - 420. * W1 * C4_4 * C4_4 * C4_4;
static const double B2_A2A = 75. * W2 * C4_3 - 420. * W2 * C4_3 * C4_3 * C4_3;
static const double B3_A1S = -15. * W1 + 180. * W1 * C4_1 * C4_1;
static const double B4_A1A = -420. * W1 * C4_4 + 2800. * W1 * C4_4 * C4_4 * C4_4;
static const double B4_A2A = -420. * W2 * C4_3 + 2800. * W2 * C4_3 * C4_3 * C4_3;

inline int scheme_points(int scheme) {
  switch (scheme) {
    case COLLOC2: case MAGNUS2: return 1;
    case COLLOC4: case MAGNUS4: return 2;
    case COLLOC6: case MAGNUS6: return 3;
    default: return 4;
  }
}
inline std::vector<double> scheme_abscissae(int scheme) {  // points(), stepper.rs:143,171,222,291,...
  switch (scheme_points(scheme)) {
    case 1: return {0.5};
    case 2: return {0.5 + C2_1, 0.5 + C2_2};
    case 3: return {0.5 + C3_1, 0.5 + C3_2, 0.5 + C3_3};
    default: return {0.5 + C4_1, 0.5 + C4_2, 0.5 + C4_3, 0.5 + C4_4};
  }
}
inline bool scheme_is_magnus(int scheme) { return scheme >= MAGNUS2; }

// Implicit collocation steppers: produce `jump`; left = jump + I, right = jump - I.
template <class T>
inline Mat<T, 4> colloc_jump(int scheme, const Mat<T, 4>* v) {
  typedef Mat<T, 4> M;
  const M eye = M::identity();
  if (scheme == COLLOC2) {  // stepper.rs:148-163
    return scale(v[0], 0.5);
  } else if (scheme == COLLOC4) {  // stepper.rs:176-208
    const M &a1 = v[0], &a2 = v[1];
    M a2a1 = mul(a2, a1);
    M k2 = mul(inv4_unwrap(add(eye, scale(a2a1, 1. / 12.))), add(a2, scale(a2a1, C2_2)));
    M k1 = add(a1, scale(mul(a1, k2), C2_1));
    return scale(add(k1, k2), 0.25);
  } else if (scheme == COLLOC6) {  // stepper.rs:226-267
    const double d12 = -2. / 3. * C3_1, d13 = -1. / 3. * C3_1, d21 = 5. / 12. * C3_1,
                 d23 = -5. / 12. * C3_1, d31 = 1. / 3. * C3_1, d32 = 2. / 3. * C3_1;
    const M &a1 = v[0], &a2 = v[1], &a3 = v[2];
    M a2a1 = mul(a2, a1);
    M a3a1 = mul(a3, a1);
    M inv1 = inv4_unwrap(sub(eye, scale(a2a1, d12 * d21)));
    M o2 = mul(inv1, sub(a2, scale(a2a1, d21)));
    M l2 = mul(inv1, sub(scale(a2, d23), scale(a2a1, d13 * d21)));
    M f3 = sub(scale(a3, d32), scale(a3a1, d12 * d31));
    M o3 = sub(sub(a3, scale(a3a1, d31)), mul(f3, o2));
    M l3 = sub(sub(eye, scale(a3a1, d13 * d31)), mul(f3, l2));
    M k3 = mul(inv4_unwrap(l3), o3);
    M k2 = sub(o2, mul(l2, k3));
    M k1 = mul(a1, sub(sub(eye, scale(k2, d12)), scale(k3, d13)));
    return add(add(scale(k1, 5. / 36.), scale(k2, 2. / 9.)), scale(k3, 5. / 36.));
  } else {  // COLLOC8, stepper.rs:295-347
    const M &a1 = v[0], &a2 = v[1], &a3 = v[2], &a4 = v[3];
    M a2a1 = mul(a2, a1), a3a1 = mul(a3, a1), a4a1 = mul(a4, a1);
    M u12 = scale(a1, D4_12), u13 = scale(a1, D4_13), u14 = scale(a1, D4_14);
    M d2 = sub(eye, scale(a2a1, D4_21 * D4_12));
    M inv2 = inv4_unwrap(d2);
    M u23 = mul(inv2, sub(scale(a2, D4_23), scale(a2a1, D4_21 * D4_13)));
    M u24 = mul(inv2, sub(scale(a2, D4_24), scale(a2a1, D4_21 * D4_14)));
    M r2 = mul(inv2, sub(a2, scale(a2a1, D4_21)));
    M l32 = sub(scale(a3, D4_32), scale(a3a1, D4_31 * D4_12));
    M d3 = sub(sub(eye, scale(a3a1, D4_31 * D4_13)), mul(l32, u23));
    M inv3 = inv4_unwrap(d3);
    M u34 = mul(inv3, sub(sub(scale(a3, D4_34), scale(a3a1, D4_31 * D4_14)), mul(l32, u24)));
    M r3 = mul(inv3, sub(sub(a3, scale(a3a1, D4_31)), mul(l32, r2)));
    M l42 = sub(scale(a4, D4_42), scale(a4a1, D4_41 * D4_12));
    M l43 = sub(sub(scale(a4, D4_43), scale(a4a1, D4_41 * D4_13)), mul(l42, u23));
    M d4 = sub(sub(sub(eye, scale(a4a1, D4_41 * D4_14)), mul(l42, u24)), mul(l43, u34));
    M inv4m = inv4_unwrap(d4);
    M r4 = mul(inv4m, sub(sub(sub(a4, scale(a4a1, D4_41)), mul(l42, r2)), mul(l43, r3)));
    M k4 = r4;
    M k3 = sub(r3, mul(u34, k4));
    M k2 = sub(sub(r2, mul(u24, k4)), mul(u23, k3));
    M k1 = sub(sub(sub(a1, mul(u14, k4)), mul(u13, k3)), mul(u12, k2));
    return add(add(add(scale(k1, W1 / 2.), scale(k2, W2 / 2.)), scale(k3, W2 / 2.)), scale(k4, W1 / 2.));
  }
}

// Explicit step matrix for any scheme.  Collocation schemes go through
// ImplicitWrapper (stepper.rs:76-89): left <- (-(right^-1)) * left.
template <class T>
inline Mat<T, 4> step_matrix(int scheme, const Mat<T, 4>* v) {
  typedef Mat<T, 4> M;
  if (!scheme_is_magnus(scheme)) {
    M jump = colloc_jump(scheme, v);
    M eye = M::identity();
    M l = add(jump, eye), r = sub(jump, eye);
    return mul(neg(inv4_unwrap(r)), l);
  }
  M omega;
  if (scheme == MAGNUS2) {  // stepper.rs:360-372
    omega = v[0];
  } else if (scheme == MAGNUS4) {  // stepper.rs:385-401
    M b1 = scale(add(v[0], v[1]), 0.5);
    M b2 = scale(sub(v[1], v[0]), 0.5 * C2_2);
    omega = sub(b1, commutator(b1, b2));
  } else if (scheme == MAGNUS6) {  // stepper.rs:414-441
    const M &a1 = v[0], &a2 = v[1], &a3 = v[2];
    const double ten_three = 10. / 3.;
    const M& b1 = a2;
    M b2 = scale(sub(a3, a1), C3_3 * ten_three);
    M b3 = scale(add(sub(a3, scale(a2, 2.)), a1), ten_three);
    M c1 = commutator(b1, b2);
    M c2 = scale(commutator(b1, add(scale(b3, 2.), c1)), -1. / 60.);
    omega = add(add(b1, scale(b3, 1. / 12.)),
                scale(commutator(add(sub(scale(b1, -20.), b3), c1), add(b2, c2)), 1. / 240.));
  } else {  // MAGNUS8, stepper.rs:469-524
    const M &a1 = v[0], &a2 = v[1], &a3 = v[2], &a4 = v[3];
    M a1s = add(a1, a4), a1a = sub(a4, a1), a2s = add(a2, a3), a2a = sub(a3, a2);
    M b1 = add(scale(a1s, B1_A1S), scale(a2s, B1_A2S));
    M b2 = add(scale(a1a, B2_A1A), scale(a2a, B2_A2A));
    M b3 = scale(sub(a1s, a2s), B3_A1S);
    M b4 = add(scale(a1a, B4_A1A), scale(a2a, B4_A2A));
    M s1 = scale(commutator(add(b1, scale(b3, 1. / 28.)), add(b2, scale(b4, 3. / 28.))), -1. / 28.);
    M r1 = scale(commutator(b1, add(scale(b3, -1. / 14.), s1)), 1. / 3.);
    M s2 = commutator(add(add(b1, scale(b3, 1. / 28.)), s1), add(add(b2, scale(b4, 3. / 28.)), r1));
    M s2p = commutator(b2, s1);
    M r2 = commutator(add(b1, scale(s1, 5. / 4.)), add(add(scale(b3, 2.0), s2), scale(s2p, 0.5)));
    M s3 = commutator(
        add(add(add(b1, scale(b3, 1. / 12.)), scale(s1, -7. / 3.)), scale(s2, -1. / 6.)),
        add(add(add(scale(b2, -9.), scale(b4, -9. / 4.)), scale(r1, 63.)), r2));
    omega = add(add(add(b1, scale(b3, 1. / 12.)), scale(s2, -7. / 120.)), scale(s3, 1. / 360.));
  }
  mat_exp(omega);
  return omega;
}

// ---------------------------------------------------------------------------
// Model layer.  src/model.rs, src/model/{gsm,interpolate,polytrope}.rs
// DimensionlessProperties, model.rs:70-91 (m_coord, rho, p are consumed by the post-processing only).
// ---------------------------------------------------------------------------
struct DiscreteModel {
  std::vector<double> r_coord, v, u, gamma1, a_star, c1, rot;
  std::vector<double> m_coord, rho, p;
  bool has_scale = false;
  double radius = 0., mass = 0., grav = 0.;
  size_t size() const { return r_coord.size(); }
};

static const double GRAV = 6.67430e-8;  // gsm.rs:13
inline double cube(double x) { return (x * x) * x; }  // f64::powi(3)

// DiscreteModel::from_gsm, gsm.rs:68-112, on arrays already read from the file.
inline DiscreteModel model_from_gsm(size_t n, const double* r, const double* m_r, const double* rho,
                                    const double* p, const double* gamma1, const double* n2,
                                    const double* omega_rot, double radius, double mass) {
  DiscreteModel md;
  md.has_scale = true; md.radius = radius; md.mass = mass; md.grav = GRAV;
  md.r_coord.resize(n); md.v.resize(n); md.u.resize(n); md.gamma1.resize(n);
  md.a_star.resize(n); md.c1.resize(n); md.rot.resize(n);
  md.m_coord.resize(n); md.rho.resize(n); md.p.resize(n);
  const double r3 = cube(radius);
  const double r4 = (radius * radius) * (radius * radius);          // f64::powi(4)
  const double fscale = std::sqrt(GRAV * mass / r3);
  const double fourpi = 4. * M_PI;
  for (size_t i = 0; i < n; ++i) {
    md.rot[i] = omega_rot[i] / fscale;                          // gsm.rs:80-81
    md.c1[i] = cube(r[i]) / r3 * mass / m_r[i];                 // gsm.rs:83
    md.a_star[i] = cube(r[i]) / (GRAV * m_r[i]) * n2[i];        // gsm.rs:85
    md.v[i] = GRAV * m_r[i] * rho[i] / (p[i] * r[i]);           // gsm.rs:87
    md.u[i] = fourpi * rho[i] * cube(r[i]) / m_r[i];            // gsm.rs:89
    md.r_coord[i] = r[i] / radius;                              // gsm.rs:94
    md.gamma1[i] = gamma1[i];
    md.m_coord[i] = m_r[i] / mass;                              // gsm.rs:95
    md.rho[i] = rho[i] / mass * r3;                             // gsm.rs:96
    md.p[i] = p[i] / GRAV / (mass * mass) * r4;                 // gsm.rs:97
  }
  md.c1[0] = mass / r3 * 3. / (fourpi * rho[0]);                // gsm.rs:84
  md.a_star[0] = 0.; md.v[0] = 0.; md.u[0] = 3.;                // gsm.rs:86,88,90
  return md;
}

// LinearInterpolator::eval, interpolate.rs:18-124 (same two-cursor walk).
inline DiscreteModel interpolate_eval(const DiscreteModel& model, const std::vector<double>& grid) {
  DiscreteModel out;
  out.has_scale = model.has_scale; out.radius = model.radius; out.mass = model.mass; out.grav = model.grav;
  const size_t ng = grid.size();
  out.r_coord.assign(ng, 0.); out.v.assign(ng, 0.); out.u.assign(ng, 0.); out.gamma1.assign(ng, 0.);
  out.a_star.assign(ng, 0.); out.c1.assign(ng, 0.); out.rot.assign(ng, 0.);
  out.m_coord.assign(ng, 0.); out.rho.assign(ng, 0.); out.p.assign(ng, 0.);
  if (ng == 0) throw std::runtime_error("empty grid");
  const std::vector<double>& x = model.r_coord;
  size_t old_it = 0;           // next index the old-grid iterator will yield
  size_t prev = 0, next = 0;
  size_t cur = 0;
  for (;;) {
    if (x[next] == grid[cur]) {
      out.r_coord[cur] = model.r_coord[next]; out.v[cur] = model.v[next]; out.u[cur] = model.u[next];
      out.gamma1[cur] = model.gamma1[next]; out.a_star[cur] = model.a_star[next];
      out.c1[cur] = model.c1[next]; out.rot[cur] = model.rot[next];
      out.m_coord[cur] = model.m_coord[next]; out.rho[cur] = model.rho[next]; out.p[cur] = model.p[next];
      if (++cur < ng) continue; else break;
    } else if (x[next] > grid[cur]) {
      double pos = (grid[cur] - x[prev]) / (x[next] - x[prev]);
      out.r_coord[cur] = grid[cur];
#define STORM_REF_INTERP(f) out.f[cur] = model.f[prev] + pos * (model.f[next] - model.f[prev])
      STORM_REF_INTERP(v); STORM_REF_INTERP(u); STORM_REF_INTERP(gamma1);
      STORM_REF_INTERP(a_star); STORM_REF_INTERP(c1); STORM_REF_INTERP(rot);
      STORM_REF_INTERP(m_coord); STORM_REF_INTERP(rho); STORM_REF_INTERP(p);
#undef STORM_REF_INTERP
      if (++cur < ng) continue; else break;
    }
    prev = next;
    if (old_it < x.size()) next = old_it++; else break;
  }
  return out;
}

// Polytrope0::eval, polytrope.rs:198-243.
static const double POLY0_MAX_X = 2.449489742783178098197284074705891391965947480656670128432692567250960377457315;
inline DiscreteModel polytrope0_eval(double gamma1, const std::vector<double>& grid) {
  DiscreteModel out;
  const size_t n = grid.size();
  out.r_coord = grid;
  out.v.resize(n); out.a_star.resize(n);
  out.u.assign(n, 3.); out.gamma1.assign(n, gamma1); out.c1.assign(n, 1.); out.rot.assign(n, 0.);
  out.rho.assign(n, 3. / (4. * M_PI)); out.m_coord.resize(n);               // polytrope.rs:223-225
  out.p.assign(n, std::numeric_limits<double>::infinity());
  const double phi_max = cube(POLY0_MAX_X) / 3.;
  for (size_t i = 0; i < n; ++i) {
    double xi = grid[i] * POLY0_MAX_X;
    double theta = 1. - xi * xi / 6.;
    double phi = cube(xi) / 3.;
    out.m_coord[i] = phi / phi_max;
    out.v[i] = phi / (xi * theta);
    out.a_star[i] = -phi / (xi * theta * gamma1);
  }
  if (n) { out.v[0] = 0.; out.a_star[0] = 0.; }
  return out;
}
inline double polytrope0_exact(double gamma1, uint64_t ell, int64_t radial_order) {  // polytrope.rs:178-186
  double n = double(ell), k = 2. * double(radial_order);
  double d = -2. + gamma1 / 4. * (k * (k + 5. + 2. * n) + 6. + 4. * n);
  double beta = d + std::sqrt(d * d + n * (n + 1.));
  return std::sqrt(beta);
}

// ---------------------------------------------------------------------------
// Bracketing.  src/bracket.rs
// ---------------------------------------------------------------------------
struct Point { double x, f; };
struct BracketResult { Point lower, upper; double root; uint64_t evals; };

inline double signum(double f) {  // f64::signum: sign bit, NaN stays NaN
  if (f != f) return f;
  return std::signbit(f) ? -1. : 1.;
}
inline uint64_t precision_relative_to_ulp(double rel) {  // bracket.rs:30-40
  double q = std::floor(rel / std::numeric_limits<double>::epsilon());
  uint64_t u;
  if (!(q > 0.)) u = 0;                                   // `as u64` saturates, NaN -> 0
  else if (q >= 18446744073709551615.) u = UINT64_MAX;
  else u = uint64_t(q);
  return std::max<uint64_t>(1, u);
}
inline double evaluate_secant(Point x1, Point x2, double y) {  // bracket.rs:73-75
  return (y - x2.f) / (x1.f - x2.f) * x1.x + (y - x1.f) / (x2.f - x1.f) * x2.x;
}
inline double evaluate_inverse_quadratic(Point x1, Point x2, Point x3, double y) {  // bracket.rs:77-81
  return (y - x2.f) * (y - x3.f) / (x1.f - x2.f) / (x1.f - x3.f) * x1.x +
         (y - x1.f) * (y - x3.f) / (x2.f - x1.f) / (x2.f - x3.f) * x2.x +
         (y - x1.f) * (y - x2.f) / (x3.f - x2.f) / (x3.f - x1.f) * x3.x;
}
inline uint64_t to_bits(double x) { uint64_t b; std::memcpy(&b, &x, 8); return b; }
inline double from_bits(uint64_t b) { double x; std::memcpy(&x, &b, 8); return x; }

// bracket.rs:96-132.  Returns true for ControlFlow::Break (x is the root).
inline bool ensure_maximal_bracket(double upper, double lower, double& x, uint64_t ulp) {
  if (!(upper > lower) || !std::isfinite(upper) || !std::isfinite(lower) || !(upper >= 0.) || !(lower >= 0.))
    throw std::runtime_error("ensure_maximal_bracket: assertion failed");
  if (!(ulp < (uint64_t(1) << 54))) throw std::runtime_error("requested ulp precision too large");
  uint64_t ub = to_bits(upper), lb = to_bits(lower);
  if (ub - lb <= ulp) return true;
  uint64_t xb = to_bits(x);
  if ((ub - lb) <= 2 * ulp) xb = lb + (ub - lb) / 2;
  else if ((ub < xb) || (ub - xb) < ulp) xb = ub - ulp;
  else if ((lb > xb) || (xb - lb) < ulp) xb = lb + ulp;
  x = from_bits(xb);
  return false;
}

// InverseQuadratic::optimize, bracket.rs:151-222.
template <class F>
inline BracketResult optimize_bracket(Point lower, Point upper, F&& f, uint64_t ulp) {
  if (!std::isfinite(lower.x) || !std::isfinite(lower.f) || !std::isfinite(upper.x) || !std::isfinite(upper.f))
    throw std::runtime_error("optimize: non-finite bracket");
  if (signum(lower.f) == signum(upper.f)) throw std::runtime_error("Upper and lower values in bracket have the same sign????");
  uint64_t evals = 0;
  bool has_prev = false;
  Point previous{0., 0.};
  for (;;) {
    double x;
    if (!has_prev) {
      x = evaluate_secant(lower, upper, 0.);
    } else {
      x = evaluate_inverse_quadratic(lower, upper, previous, 0.);
      if (x <= lower.x || x >= upper.x) x = evaluate_secant(lower, upper, 0.);
    }
    if (ensure_maximal_bracket(upper.x, lower.x, x, ulp)) return BracketResult{lower, upper, x, evals};
    Point next{x, f(x)};
    if (!std::isfinite(next.x) || !std::isfinite(next.f)) throw std::runtime_error("optimize: non-finite evaluation");
    evals += 1;
    if (signum(upper.f) == signum(next.f)) { previous = upper; has_prev = true; upper = next; }
    else { previous = lower; has_prev = true; lower = next; }
  }
}

// ---------------------------------------------------------------------------
// Polytrope construction.  src/model/polytrope.rs:9-165
// ---------------------------------------------------------------------------
inline void eval_lane_emden(double xi, double phi, double theta, double n, double& d0, double& d1) {
  d0 = std::pow(theta, n) * (xi * xi);
  d1 = (xi == 0.) ? 0. : -phi / (xi * xi);
}
inline void rk4(double xi, double phi, double theta, double n, double step, double& ophi, double& otheta) {
  double hstep = step / 2.;
  double k10, k11, k20, k21, k30, k31, k40, k41;
  eval_lane_emden(xi, phi, theta, n, k10, k11);
  eval_lane_emden(xi + hstep, phi + hstep * k10, theta + hstep * k11, n, k20, k21);
  eval_lane_emden(xi + hstep, phi + hstep * k20, theta + hstep * k21, n, k30, k31);
  eval_lane_emden(xi + step, phi + step * k30, theta + step * k31, n, k40, k41);
  ophi = phi + step / 6. * (k10 + 2. * k20 + 2. * k30 + k40);
  otheta = theta + step / 6. * (k11 + 2. * k21 + 2. * k31 + k41);
}
inline DiscreteModel construct_polytrope(double n, double gamma1, double step) {
  if (!(n < 5.) || !(n >= 0.)) throw std::runtime_error("invalid polytropic index");
  std::vector<double> thetas, phis, xis;
  double theta = 1., phi = 0.;
  uint64_t nsteps = 0;
  while (theta > 0.) {  // polytrope.rs:40-47
    thetas.push_back(theta); phis.push_back(phi); xis.push_back(double(nsteps) * step);
    double np, nt;
    rk4(double(nsteps) * step, phi, theta, n, step, np, nt);
    phi = np; theta = nt;
    nsteps += 1;
  }
  const double xlast = double(nsteps - 1) * step, plast = phis.back(), tlast = thetas.back();
  BracketResult br = optimize_bracket(Point{0., tlast}, Point{step, theta},
      [&](double val) { double a, b; rk4(xlast, plast, tlast, n, val, a, b); return b; }, 1);
  rk4(xlast, plast, tlast, n, br.lower.x, phi, theta);
  thetas.push_back(theta); phis.push_back(phi); xis.push_back(xlast + br.root);
  const double max_phi = phis.back(), max_xi = xis.back();
  const size_t len = xis.size() - 1;
  DiscreteModel md;
  md.r_coord.resize(len); md.v.resize(len); md.u.resize(len); md.a_star.resize(len); md.c1.resize(len);
  md.gamma1.assign(len, gamma1); md.rot.assign(len, 0.);
  md.m_coord.resize(len); md.rho.resize(len); md.p.resize(len);
  for (size_t i = 0; i < len; ++i) {
    double xi = xis[i], th = thetas[i], ph = phis[i];
    md.m_coord[i] = ph / max_phi;                                    // :139
    md.rho[i] = std::pow(th, n) * cube(max_xi) / (max_phi * 4. * M_PI);  // :143
    md.u[i] = cube(xi) * std::pow(th, n) / ph;                       // polytrope.rs:107
    md.v[i] = (n + 1.) * ph / (xi * th);                             // :116
    md.c1[i] = cube(xi) / cube(max_xi) * max_phi / ph;               // :124
    md.a_star[i] = -((n + 1.) / gamma1 - n) * ph / (xi * th);        // :133
    md.r_coord[i] = xi / max_xi;                                     // :138
  }
  md.u[0] = 3.; md.v[0] = 0.; md.c1[0] = max_phi / cube(max_xi) * 3.; md.a_star[0] = 0.;
  const double kk = 4. * M_PI / (std::pow(md.rho[0], 1. / n - 1.) * (n + 1.) * (max_xi * max_xi));  // :146
  for (size_t i = 0; i < len; ++i) md.p[i] = std::pow(md.rho[i], 1. + 1. / n) * kk;                  // :147
  return md;
}

// A ContinuousModel (model.rs:17-26): either a LinearInterpolator over a discrete
// model or the analytic Polytrope0.
struct ContinuousModel {
  bool is_poly0 = false;
  double gamma1 = 0.;
  const DiscreteModel* discrete = nullptr;
  double inner() const { return is_poly0 ? 0. : discrete->r_coord.front(); }
  double outer() const { return is_poly0 ? 1. : discrete->r_coord.back(); }
  DiscreteModel eval(const std::vector<double>& grid) const {
    return is_poly0 ? polytrope0_eval(gamma1, grid) : interpolate_eval(*discrete, grid);
  }
};

// linspace of src/main.rs:1232-1234 and Model::grid of src/main.rs:661-679.
inline std::vector<double> linspace(double lower, double upper, size_t n) {
  std::vector<double> out(n);
  for (size_t i = 0; i < n; ++i) out[i] = lower + (upper - lower) * double(i) / double(n - 1);
  return out;
}
inline std::vector<double> rev_linspace(double lower, double upper, size_t n) {  // main.rs:1236-1238
  std::vector<double> out = linspace(1. / lower, 1. / upper, n);
  for (double& x : out) x = 1. / x;
  return out;
}
inline std::vector<double> model_grid_discrete(const DiscreteModel& m, size_t scale) {
  std::vector<double> g;
  for (size_t i = 0; i + 1 < m.size(); ++i) {
    double a0 = m.r_coord[i], a1 = m.r_coord[i + 1];
    for (size_t k = 0; k < scale; ++k) g.push_back(a0 + (a1 - a0) * double(k) / double(scale));
  }
  g.push_back(m.r_coord.back());
  return g;
}

// ---------------------------------------------------------------------------
// Rotating1D system.  src/system/adiabatic.rs
// ---------------------------------------------------------------------------
struct Coef { double u, v_gamma, a_star, c1, rot; };  // adiabatic.rs:29-40

inline std::vector<Coef> coefs_from_model(const DiscreteModel& m) {  // adiabatic.rs:42-59
  std::vector<Coef> c(m.size());
  for (size_t i = 0; i < m.size(); ++i)
    c[i] = Coef{m.u[i], m.v[i] / m.gamma1[i], m.a_star[i], m.c1[i], m.rot[i]};
  return c;
}

template <class T>
inline void system_eval(double ell_, const Coef& p, double delta_, Mat<T, 4>& o) {  // adiabatic.rs:78-114
  T zero(0.), one(1.), two(2.), three(3.), ell(ell_), delta(delta_), v_gamma(p.v_gamma), u(p.u), a_star(p.a_star);
  T lambda = ell * (ell + one);
  o(0, 0) = delta * (v_gamma - one - ell);
  o(0, 1) = -delta * v_gamma;
  o(0, 2) = zero; o(0, 3) = zero;
  o(1, 0) = -delta * a_star;
  o(1, 1) = delta * (a_star - u + three - ell);
  o(1, 2) = zero;
  o(1, 3) = -delta;
  o(2, 0) = zero; o(2, 1) = zero;
  o(2, 2) = delta * (three - u - ell);
  o(2, 3) = delta;
  o(3, 0) = delta * (u * a_star);
  o(3, 1) = delta * (u * v_gamma);
  o(3, 2) = delta * lambda;
  o(3, 3) = delta * (-u + two - ell);
}

template <class T>
inline void system_add_frequency(double ell_, double m_, T freq, const Coef& p, double delta_, Mat<T, 4>& o) {
  // adiabatic.rs:116-152
  T zero(0.), one(1.), two(2.), ell(ell_), delta(delta_), rot0(p.rot), c1(p.c1);
  T lambda = ell * (ell + one);
  T dc1 = delta * c1;
  T dlambda = delta * lambda;
  if (ell != zero) {
    T rot = T(m_) * rot0;
    T omega_rsq = (lambda * (freq - rot) + two * rot) * (freq - rot);
    T rel_rot = two * rot / (lambda * (freq - rot) + two * rot);
    o(0, 0) += -dlambda * rel_rot;
    o(0, 1) += dlambda * dlambda / (omega_rsq * dc1);
    o(0, 2) += dlambda * dlambda / (omega_rsq * dc1);
    o(1, 1) += dlambda * rel_rot;
    o(1, 2) += dlambda * rel_rot;
    o(1, 0) += dc1 * ((freq - rot) * (freq - rot)) * (one - ((two * rot) * (two * rot)) / omega_rsq);
  } else {
    o(1, 0) += dc1 * (freq * freq);
  }
}

// inner_boundary (2x4) and outer_boundary (2x4), adiabatic.rs:154-208. Row-major [row][col].
template <class T>
inline void system_inner_boundary(double ell_, double m_, T frequency, const Coef& p, T out[2][4]) {
  T zero(0.), one(1.), two(2.), ell(ell_), c1(p.c1);
  if (ell != zero) {
    T rot(m_ * p.rot);
    T omega2 = (frequency - rot + two * rot / ell) * (frequency - rot);
    out[0][0] = c1 * omega2;
  } else {
    out[0][0] = c1 * frequency * frequency;
  }
  out[0][1] = -ell; out[0][2] = -ell; out[0][3] = zero;
  out[1][0] = zero; out[1][1] = zero; out[1][2] = ell; out[1][3] = -one;
}
template <class T>
inline void system_outer_boundary(double ell_, const Coef& p, T out[2][4]) {
  T zero(0.), one(1.);
  out[0][0] = one; out[0][1] = -one; out[0][2] = zero; out[0][3] = zero;
  out[1][0] = T(p.u); out[1][1] = zero; out[1][2] = T(ell_) + one; out[1][3] = one;
}

// ---------------------------------------------------------------------------
// DiscretizedSystemImpl.  src/system/discretized.rs
// ---------------------------------------------------------------------------
struct Discretized {
  int scheme = 0, P = 1;
  double ell = 0., m = 0.;
  std::vector<Mat<double, 4>> matrices;   // cached frequency-independent matrices, [steps*P]
  std::vector<Coef> pts;                  // interpolated_points, [steps*P]
  std::vector<double> delta;              // [steps]
  std::vector<double> xs;                 // collocation abscissae, [steps*P]
  Coef inner{}, outer{};
  size_t len() const { return delta.size(); }
};

inline Discretized discretize(const ContinuousModel& model, int scheme, uint64_t ell, int64_t m,
                              const std::vector<double>& grid) {  // discretized.rs:81-123
  Discretized d;
  d.scheme = scheme; d.P = scheme_points(scheme);
  d.ell = double(ell); d.m = double(m);
  if (grid.size() < 1) throw std::runtime_error("grid too small");
  std::vector<double> loc = scheme_abscissae(scheme);
  size_t steps = grid.size() - 1;
  d.delta.resize(steps);
  for (size_t i = 0; i < steps; ++i) d.delta[i] = grid[i + 1] - grid[i];
  d.xs.resize(steps * d.P);
  for (size_t i = 0; i < steps; ++i)
    for (int j = 0; j < d.P; ++j) d.xs[i * d.P + j] = grid[i] + loc[j] * d.delta[i];
  d.pts = steps ? coefs_from_model(model.eval(d.xs)) : std::vector<Coef>();
  d.matrices.resize(steps * d.P);
  for (size_t i = 0; i < steps; ++i)
    for (int j = 0; j < d.P; ++j) {
      size_t idx = i * d.P + j;
      d.matrices[idx] = Mat<double, 4>::zeros();
      system_eval<double>(d.ell, d.pts[idx], d.delta[i] / d.xs[idx], d.matrices[idx]);
    }
  d.inner = coefs_from_model(model.eval({model.inner()}))[0];
  d.outer = coefs_from_model(model.eval({model.outer()}))[0];
  return d;
}

// fill_explicit, discretized.rs:217-244.
template <class T>
inline Mat<T, 4> fill_explicit(const Discretized& d, size_t step, T frequency) {
  Mat<T, 4> vals[4];
  for (int i = 0; i < d.P; ++i) {
    size_t idx = step * d.P + i;
    for (int k = 0; k < 16; ++k) vals[i].a[k] = T(d.matrices[idx].a[k]);
    system_add_frequency<T>(d.ell, d.m, frequency, d.pts[idx], d.delta[step] / d.xs[idx], vals[i]);
  }
  return step_matrix<T>(d.scheme, vals);
}

// ---------------------------------------------------------------------------
// Determinant.  src/solver.rs
// ---------------------------------------------------------------------------
// `bands` of the reference is a column-major (2n) x (n + n_inner) matrix whose
// columns are system rows: B(j, i) = system row i, system column j.
template <class T>
struct Bands {
  T b[6][8];  // [system row][system column]
  T& operator()(int col, int row) { return b[row][col]; }
};

// row_pivot!, solver.rs:231-269.  Returns pivot, fills pivot_row[cols].
template <class T>
inline T row_pivot(Bands<T>& B, T& det, int rows, int cols, int idx, T* pivot_row) {
  int max_idx = idx;
  double max_val = absval(B(idx, idx));
  for (int i = idx + 1; i < rows; ++i) {
    if (absval(B(idx, i)) > max_val) { max_idx = i; max_val = absval(B(idx, i)); }
  }
  T pivot = B(idx, max_idx);
  if (max_idx != idx) {
    for (int i = 0; i < cols; ++i) { pivot_row[i] = B(i, max_idx); B(i, max_idx) = B(i, idx); }
    det *= -T(1.);
  } else {
    for (int i = 0; i < cols; ++i) pivot_row[i] = B(i, idx);
  }
  return pivot;
}
// column_pivot!, solver.rs:192-230.
template <class T>
inline T column_pivot(Bands<T>& B, std::vector<std::pair<int, int>>* swaps, T& det, int rows, int cols,
                      int idx, T* pivot_row) {
  int max_idx = idx;
  double max_val = absval(B(idx, idx));
  for (int i = idx + 1; i < cols; ++i) {
    if (absval(B(i, idx)) > max_val) { max_idx = i; max_val = absval(B(i, idx)); }
  }
  T pivot = B(max_idx, idx);
  if (max_idx != idx) {
    if (swaps) swaps->push_back({max_idx, idx});
    for (int i = idx; i < rows; ++i) { T c1 = B(max_idx, i), c2 = B(idx, i); B(max_idx, i) = c2; B(idx, i) = c1; }
    det *= -T(1.);
  }
  for (int i = 0; i < cols; ++i) pivot_row[i] = B(i, idx);
  return pivot;
}
// sweep!, solver.rs:163-190 (second arm).
template <class T>
inline void sweep(Bands<T>& B, int rows, int cols, int idx, T pivot, const T* pivot_row) {
  T pinv = T(1.) / pivot;
  for (int r = idx + 1; r < rows; ++r) {
    T m = B(idx, r) * pinv;
    for (int j = 0; j < cols; ++j) B(j, r) -= pivot_row[j] * m;
  }
}

// determinant_explicit, solver.rs:435-542.
template <class T>
inline T determinant_explicit(const Discretized& d, T frequency) {
  const int n = 4, n_inner = 2;
  T outer[2][4], inner[2][4];
  system_outer_boundary<T>(d.ell, d.outer, outer);
  system_inner_boundary<T>(d.ell, d.m, frequency, d.inner, inner);
  Bands<T> B;
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 8; ++c) B.b[r][c] = T(0.);
  for (int i = 0; i < n_inner; ++i) for (int j = 0; j < n; ++j) B(j, i) = inner[i][j];
  T det(1.);
  Mat<T, 4> accum = Mat<T, 4>::identity();
  for (size_t step = 0; step < d.len(); ++step) {
    Mat<T, 4> left = fill_explicit<T>(d, step, frequency);
    accum = mul(left, accum);  // gemm(1, left, accum2, 0), solver.rs:489-491
  }
  for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) {
    B(c, r + n_inner) = accum(r, c);
    B(c + n, r + n_inner) = (r == c) ? -T(1.) : T(0.);
  }
  T pivot_row[8];
  for (int k = 0; k < n; ++k) {
    T pivot = row_pivot(B, det, n + n_inner, 2 * n, k, pivot_row);
    det *= pivot;
    sweep(B, n + n_inner, 2 * n, k, pivot, pivot_row);
  }
  for (int i = 0; i < n_inner; ++i) for (int j = 0; j < n; ++j) { B(j, i) = B(j + n, i + n); B(j + n, i) = T(0.); }
  for (int r = 0; r < n - n_inner; ++r) for (int c = 0; c < n; ++c) B(c, n_inner + r) = outer[r][c];
  for (int k = 0; k < n - 1; ++k) {
    T pivot = row_pivot(B, det, n, n, k, pivot_row);
    det *= pivot;
    sweep(B, n, n, k, pivot, pivot_row);
  }
  return det * B(n - 1, n - 1);
}

// UpperResult, solver.rs:8-13,57-59,86-161.
inline size_t gauss(size_t lower, size_t upper) { return (upper - lower + 1) * (upper + lower) / 2; }
struct UpperResult {
  std::vector<double> data;
  size_t n, n_systems;
  std::vector<std::pair<int, int>> column_pivot;
  UpperResult(size_t matrix_size, size_t points)
      : data(points * gauss(matrix_size, 2 * matrix_size - 1) + gauss(1, matrix_size - 1), 0.),
        n(matrix_size), n_systems(points) {}
  void set(size_t point, size_t k, size_t i, double val) {  // solver.rs:140-151
    if (point != n_systems)
      data[point * gauss(n, 2 * n - 1) + gauss(2 * n - 1 - k, 2 * n - 1) - (2 * n - 1 - k) + (i - 1)] = val;
    else
      data[point * gauss(n, 2 * n - 1) + gauss(n - 1 - k, n - 1) - (n - 1 - k) + (i - 1)] = val;
  }
  std::vector<double> eigenvectors() const {  // solver.rs:100-131
    std::vector<double> e((n_systems + 1) * n, 0.);
    const size_t len = e.size();
    e[len - 1] = 1.;
    size_t it = data.size();  // reverse iterator
    for (size_t i = 1; i < n; ++i) {
      double next_val = 0.;
      for (size_t j = 0; j < i; ++j) next_val -= data[--it] * e[len - j - 1];
      e[len - i - 1] = next_val;
    }
    for (size_t k = 0; k < n_systems; ++k)
      for (size_t i = 0; i < n; ++i) {
        double next_val = 0.;
        for (size_t j = 0; j < i + n; ++j) next_val -= data[--it] * e[len - j - k * n - 1];
        e[len - i - (k + 1) * n - 1] = next_val;
      }
    for (size_t s = column_pivot.size(); s-- > 0;) std::swap(e[column_pivot[s].first], e[column_pivot[s].second]);
    return e;
  }
};

// sweep! first arm (solver.rs:164-174): store the normalised pivot row, then sweep.
inline void sweep_upper(Bands<double>& B, UpperResult& up, size_t n_step, int rows, int cols, int idx,
                        double pivot, const double* pivot_row) {
  double pinv = 1. / pivot;
  for (int i = idx + 1; i < cols; ++i) up.set(n_step, idx, i - idx, pivot_row[i] * pinv);
  sweep(B, rows, cols, idx, pivot, pivot_row);
}

// determinant_inner via determinant_with_upper, solver.rs:271-433.  For explicit
// steppers fill() gives right = -I (stepper.rs:57-58).
inline double determinant_with_upper(const Discretized& d, double frequency, UpperResult& up) {
  const int n = 4, n_inner = 2;
  double outer[2][4], inner[2][4];
  system_outer_boundary<double>(d.ell, d.outer, outer);
  system_inner_boundary<double>(d.ell, d.m, frequency, d.inner, inner);
  Bands<double> B;
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 8; ++c) B.b[r][c] = 0.;
  for (int i = 0; i < n_inner; ++i) for (int j = 0; j < n; ++j) B(j, i) = inner[i][j];
  double det = 1.;
  double pivot_row[8];
  size_t n_step = 0;
  for (; n_step < d.len(); ++n_step) {
    Mat<double, 4> left = fill_explicit<double>(d, n_step, frequency);
    for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) {
      B(c, r + n_inner) = left(r, c);
      B(c + n, r + n_inner) = (r == c) ? -1. : 0.;
    }
    for (int k = 0; k < n; ++k) {
      double pivot;
      if (n_step == 0 && k < n_inner) pivot = column_pivot(B, &up.column_pivot, det, n + n_inner, 2 * n, k, pivot_row);
      else pivot = row_pivot(B, det, n + n_inner, 2 * n, k, pivot_row);
      det *= pivot;
      sweep_upper(B, up, n_step, n + n_inner, 2 * n, k, pivot, pivot_row);
    }
    for (int i = 0; i < n_inner; ++i) for (int j = 0; j < n; ++j) { B(j, i) = B(j + n, i + n); B(j + n, i) = 0.; }
  }
  for (int r = 0; r < n - n_inner; ++r) for (int c = 0; c < n; ++c) B(c, n_inner + r) = outer[r][c];
  for (int k = 0; k < n - 1; ++k) {
    double pivot = row_pivot(B, det, n, n, k, pivot_row);
    det *= pivot;
    sweep_upper(B, up, n_step, n, n, k, pivot, pivot_row);
  }
  return det * B(n - 1, n - 1);
}

// ---------------------------------------------------------------------------
// ErasedSolver.  src/dynamic_interface.rs:60-160
// ---------------------------------------------------------------------------
template <class Fn>
inline void parallel_for(size_t n, int threads, Fn&& fn) {  // order-preserving map like rayon's collect
  if (threads <= 1 || n < 2) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  std::vector<std::thread> pool;
  size_t nt = std::min<size_t>(threads, n);
  for (size_t t = 0; t < nt; ++t)
    pool.emplace_back([&, t]() { for (size_t i = t; i < n; i += nt) fn(i); });
  for (auto& th : pool) th.join();
}

struct ErasedSolver {
  Discretized sys;
  double det(double f) const { return determinant_explicit<double>(sys, f); }
  std::vector<double> eigenvector(double f, double* det_out = nullptr) const {
    UpperResult up(4, sys.len());
    double dv = determinant_with_upper(sys, f, up);
    if (det_out) *det_out = dv;
    return up.eigenvectors();
  }
  // scan_and_optimize, dynamic_interface.rs:116-160.
  std::vector<BracketResult> scan_and_optimize(const std::vector<double>& grid, uint64_t ulp, bool* had_invalid,
                                               int threads, std::vector<double>* dets_out = nullptr,
                                               std::vector<uint64_t>* bracket_index = nullptr) const {
    std::vector<Point> pts(grid.size());
    parallel_for(grid.size(), threads, [&](size_t i) { pts[i] = Point{grid[i], det(grid[i])}; });
    if (dets_out) { dets_out->resize(pts.size()); for (size_t i = 0; i < pts.size(); ++i) (*dets_out)[i] = pts[i].f; }
    bool invalid = false;
    std::vector<std::pair<Point, Point>> pairs;
    for (size_t i = 0; i + 1 < pts.size(); ++i) {  // filter_sign_swap, bracket.rs:236-250
      double s1 = signum(pts[i].f), s2 = signum(pts[i + 1].f);
      if (s1 != s2) {
        if (std::isfinite(pts[i].f) && std::isfinite(pts[i + 1].f)) {
          pairs.push_back({pts[i], pts[i + 1]});
          if (bracket_index) bracket_index->push_back(i);
        } else invalid = true;
      }
    }
    if (had_invalid) *had_invalid = invalid;
    std::vector<BracketResult> res(pairs.size());
    parallel_for(pairs.size(), threads, [&](size_t i) {
      res[i] = optimize_bracket(pairs[i].first, pairs[i].second, [&](double x) { return det(x); }, ulp);
    });
    return res;
  }
};

// ---------------------------------------------------------------------------
// Post-processing.  src/postprocessing.rs, src/gaunt.rs
// Pinned by the reference only through the mode ids (postprocessing.rs:424-498); the floating-point
// fields are asserted nowhere in the reference: "parity unpinned" (GPU == this restatement).
// ---------------------------------------------------------------------------
// f64::powi with a run-time exponent = compiler-builtins' __powidf2 (square-and-multiply on |b|,
// reciprocal at the end for b < 0); constant small exponents are spelled out as products instead.
inline double powi(double a, int b) {
  unsigned pw = b < 0 ? 0u - unsigned(b) : unsigned(b);
  double mul = 1.;
  for (;;) {
    if (pw & 1u) mul *= a;
    pw >>= 1;
    if (pw == 0) break;
    a *= a;
  }
  return b < 0 ? 1. / mul : mul;
}
inline double beta_k(uint64_t k, int64_t m) {  // gaunt.rs:1-6
  double kf = double(k), mf = double(m);
  return std::sqrt((kf * kf - mf * mf) / (4. * kf * kf - 1.));
}
inline double q_kl1(uint64_t k, uint64_t l, int64_t m) {  // gaunt.rs:8-17; m % 2 truncates towards zero
  double sign = powi(-1.0, int(m % 2));
  return sign * (k == l + 1 ? beta_k(k, m) : (l == k + 1 ? beta_k(l, m) : 0.));
}
inline double q_kl1_h(uint64_t k, uint64_t l, int64_t m) {  // gaunt.rs:32-37
  double lk = double(k * (k + 1)), ll = double(l * (l + 1));
  return q_kl1(k, l, m) * ((lk + ll) / 2. - 1.);
}
inline double q_kl1_hd(uint64_t k, uint64_t l, int64_t m) {  // gaunt.rs:46-51
  double lk = double(k * (k + 1)), ll = double(l * (l + 1));
  return q_kl1(k, l, m) * ((ll - lk) / 2. + 1.);
}

struct Windings { uint64_t clockwise = 0, counter_clockwise = 0; std::vector<double> nodes; };
// count_windings, postprocessing.rs:304-355 (n = number of points of the three slices).
inline Windings count_windings(const double* y1, const double* y2, const double* x, size_t n) {
  Windings w;
  for (size_t i = 0; i + 1 < n; ++i) {
    const double y1_1 = y1[i], y1_2 = y1[i + 1], y2_1 = y2[i], y2_2 = y2[i + 1], x_1 = x[i], x_2 = x[i + 1];
    if (y1_1 <= 0. && y1_2 > 0.) {
      double yt = y2_1 - y1_1 * (y2_2 - y2_1) / (y1_2 - y1_1);
      double xt = x_2 - y1_1 * (x_2 - x_1) / (y1_2 - y1_1);
      if (yt > 0.) w.clockwise += 1; else w.counter_clockwise += 1;
      w.nodes.push_back(xt);
    } else if (y1_1 >= 0. && y1_2 < 0.) {
      double yt = y2_1 - y1_1 * (y2_2 - y2_1) / (y1_2 - y1_1);
      double xt = x_2 - y1_1 * (x_2 - x_1) / (y1_2 - y1_1);
      if (yt > 0.) w.counter_clockwise += 1; else w.clockwise += 1;
      w.nodes.push_back(xt);
    }
  }
  return w;
}

// Field order of the flat output (also the HIP library's): include/storm_hip.h storm_post_field.
enum PostField { PF_Y1, PF_Y2, PF_Y3, PF_Y4, PF_XI_R, PF_XI_H, PF_XI_TN, PF_XI_TP, PF_PSI, PF_DPSI, PF_RHO, PF_P, PF_CHI, PF_COUNT };

struct Postprocessing {  // Rotating1DPostprocessing, postprocessing.rs:14-54
  std::vector<double> f[PF_COUNT];
  uint64_t cross_clockwise = 0, cross_counter_clockwise = 0;
  int64_t radial_order = 0;
  std::vector<double> nodes;
};

// Rotating1DPostprocessing::new, postprocessing.rs:58-301.
inline Postprocessing postprocess(double freq, const double* eigenvector, size_t eig_len, uint64_t ell, int64_t m,
                                  const DiscreteModel& model) {
  const size_t np = model.r_coord.size();
  if (eig_len % 4 != 0) throw std::runtime_error("eigenvector length is not a multiple of 4");
  if (np < 2 || eig_len != np * 4) throw std::runtime_error("eigenvector does not live on the model grid");
  if (model.r_coord[0] != 0.) throw std::runtime_error("model.r_coord[0] must be 0");
  Postprocessing out;
  for (auto& v : out.f) v.assign(np, 0.);
  std::vector<double>&y1 = out.f[PF_Y1], &y2 = out.f[PF_Y2], &y3 = out.f[PF_Y3], &y4 = out.f[PF_Y4], &xi_r = out.f[PF_XI_R],
      &xi_h = out.f[PF_XI_H], &xi_tn = out.f[PF_XI_TN], &xi_tp = out.f[PF_XI_TP], &psi_prime = out.f[PF_PSI],
      &dpsi_prime = out.f[PF_DPSI], &rho_prime = out.f[PF_RHO], &p_prime = out.f[PF_P], &chi = out.f[PF_CHI];
  const double lambda = double(ell * (ell + 1));
  const uint64_t ell_n1 = ell == 0 ? 0 : ell - 1;  // saturating_sub
  const double lambda_n1 = double(ell * ell_n1);
  const double lambda_p1 = double((ell + 2) * (ell + 1));
  const double q_hd_n = q_kl1_hd(ell, ell_n1, m), q_hd_p = q_kl1_hd(ell, ell + 1, m);
  const double q_h_n = q_kl1_h(ell, ell_n1, m), q_h_p = q_kl1_h(ell, ell + 1, m);
  const double mf = double(m);
  const int ell_i = int(ell);
  const double ellf = double(ell);
  const uint64_t m_abs = m < 0 ? uint64_t(0) - uint64_t(m) : uint64_t(m);
  const std::vector<double>& r = model.r_coord;
  double norm = 0.;
  std::vector<double> trapezoid(np, 0.);  // :99-111
  trapezoid[0] = 0.5 * (r[1] - r[0]);
  trapezoid[np - 1] = 0.5 * (r[np - 1] - r[np - 2]);
  for (size_t i = 1; i + 1 < np; ++i) trapezoid[i] = 0.5 * (r[i + 1] - r[i - 1]);

  for (size_t i = 1; i < np; ++i) {  // :113-171
    y1[i] = eigenvector[i * 4]; y2[i] = eigenvector[i * 4 + 1]; y3[i] = eigenvector[i * 4 + 2]; y4[i] = eigenvector[i * 4 + 3];
    const double dphi = model.m_coord[i] / (r[i] * r[i]);
    xi_r[i] = y1[i] * powi(r[i], ell_i - 1);
    p_prime[i] = y2[i] * model.rho[i] * dphi * powi(r[i], ell_i - 1);
    psi_prime[i] = y3[i] * dphi * powi(r[i], ell_i - 1);
    dpsi_prime[i] = y4[i] * dphi * powi(r[i], ell_i - 2);
    const double rsigma = freq - mf * model.rot[i];
    double omega_rsq, rel_rot;
    if (ell != 0) {
      const double rot = mf * model.rot[i];
      omega_rsq = (lambda * (freq - rot) + 2. * rot) * (freq - rot);
      rel_rot = 2. * rot / (lambda * (freq - rot) + 2. * rot);
      const double f = 2. * rot / lambda;
      xi_h[i] = 1. / (rsigma * model.rho[i] * (rsigma + f)) *
                ((p_prime[i] + model.rho[i] * psi_prime[i]) / r[i] - f * rsigma * model.rho[i] * xi_r[i]);
    } else {
      omega_rsq = 1.; rel_rot = 0.; xi_h[i] = 0.;
    }
    const double xdy1 = (model.v[i] / model.gamma1[i] - 1. - ellf - lambda * rel_rot) * y1[i] +
                        (-model.v[i] / model.gamma1[i] + (lambda * lambda) / (omega_rsq * model.c1[i])) * y2[i] +
                        (lambda * lambda) / (omega_rsq * model.c1[i]) * y3[i];
    chi[i] = powi(r[i], ell_i - 2) * ((ellf + 1.) * y1[i] + xdy1) - lambda / r[i] * xi_h[i];
    rho_prime[i] = -model.rho[i] * chi[i] + xi_r[i] * model.rho[i] / r[i] * (model.v[i] / model.gamma1[i] + model.a_star[i]);
    if (m_abs == ell || ell == 1) xi_tn[i] = 0.;
    else
      xi_tn[i] = 2. * model.rot[i] / (lambda_n1 * (freq - mf * model.rot[i]) + 2. * mf * model.rot[i]) *
                 (-q_hd_n * xi_r[i] + q_h_n * xi_h[i]);
    xi_tp[i] = 2. * model.rot[i] / (lambda_p1 * (freq - mf * model.rot[i]) + 2. * mf * model.rot[i]) *
               (-q_hd_p * xi_r[i] + q_h_p * xi_h[i]);
    norm += model.rho[i] * (r[i] * r[i]) * (xi_r[i] * xi_r[i] + lambda * xi_h[i] * xi_h[i]) * trapezoid[i];
  }
  // central point, :173-224
  y1[0] = eigenvector[0]; y2[0] = eigenvector[1]; y3[0] = eigenvector[2]; y4[0] = eigenvector[3];
  if (ell != 1) {
    xi_r[0] = 0.; dpsi_prime[0] = 0.; xi_h[0] = 0.;
  } else {
    const double ddphi0 = 4. / 3. * M_PI * model.rho[0];
    xi_r[0] = y1[0];
    dpsi_prime[0] = y4[0] * ddphi0;
    const double rsigma = freq - mf * model.rot[0];
    const double f = 2. * mf * model.rot[0] / lambda;  // ell == 1 here, so the ell != 0 arm
    xi_h[0] = 1. / (rsigma * (rsigma + f)) * ((y2[0] + y3[0]) * ddphi0 - f * xi_r[0]);
  }
  if (ell == 0) {
    p_prime[0] = y2[0] * (model.rho[0] * model.rho[0]) * 4. / 3. * M_PI;
    psi_prime[0] = y3[0] * model.rho[0] * 4. / 3. * M_PI;
    rho_prime[0] = model.rho[0] * p_prime[0] / (model.gamma1[0] * model.p[0]);
    chi[0] = -rho_prime[0] / model.rho[0];
  } else {
    p_prime[0] = 0.; psi_prime[0] = 0.; rho_prime[0] = 0.; chi[0] = 0.;
  }
  if (m_abs == ell || ell == 1) xi_tn[0] = 0.;
  else
    xi_tn[0] = 2. * model.rot[0] / (lambda_n1 * (freq - mf * model.rot[0]) + 2. * mf * model.rot[0]) *
               (-q_hd_n * xi_r[0] + q_h_n * xi_h[0]);
  if (ell == 0) xi_tp[0] = 0.;
  else
    xi_tp[0] = 2. * model.rot[0] / (lambda_p1 * (freq - mf * model.rot[0]) + 2. * mf * model.rot[0]) *
               (-q_hd_p * xi_r[0] + q_h_p * xi_h[0]);

  const double scale = 1. / std::sqrt(norm);  // :226-242
  for (auto& v : out.f) for (double& e : v) e *= scale;

  Windings w;  // :244-279
  if (ell == 0) {
    w = count_windings(y1.data(), y2.data(), r.data(), np);
    out.radial_order = int64_t(w.counter_clockwise);
  } else if (ell == 1) {
    std::vector<double> y1_alt(np), y2_alt(np);
    for (size_t i = 0; i < np; ++i) {
      y1_alt[i] = (1. - model.u[i] / 3.) * y1[i] + (y3[i] - y4[i]) / 3.;
      y2_alt[i] = y2[i] - y1[i];
    }
    if (np < 2) throw std::runtime_error("grid too short");
    w = count_windings(y1_alt.data() + 2, y2_alt.data() + 2, r.data() + 2, np - 2);
    out.radial_order = int64_t(w.counter_clockwise) - int64_t(w.clockwise) + (w.clockwise > w.counter_clockwise ? 0 : 1);
  } else {
    std::vector<double> y2_alt(np);
    for (size_t i = 0; i < np; ++i) y2_alt[i] = y2[i] + y3[i];
    w = count_windings(y1.data(), y2_alt.data(), r.data(), np);
    out.radial_order = int64_t(w.counter_clockwise) - int64_t(w.clockwise);
  }
  out.cross_clockwise = w.clockwise; out.cross_counter_clockwise = w.counter_clockwise;
  out.nodes = std::move(w.nodes);
  return out;
}

// ---------------------------------------------------------------------------
// Mode coupling integrals of the deformed star.
// perturb_deformed, src/perturbed.rs:56-263: the six M x M matrices; the generalised eigenproblem built
// from them (src/perturbed.rs:265-320, src/linalg/qz.rs) stays with the host and is not restated.
// Pinned by nothing in the reference (its only test is #[ignore]d "used for output"): parity unpinned.
// ---------------------------------------------------------------------------
inline double q_kl2(uint64_t k, uint64_t l, int64_t m) {  // gaunt.rs:19-30
  double v;
  if (k == l) { double a = beta_k(k + 1, m), b = beta_k(k, m); v = a * a + b * b - 1. / 3.; }
  else if (k == l + 2) v = beta_k(k, m) * beta_k(l + 1, m);
  else if (k + 2 == l) v = beta_k(k + 1, m) * beta_k(l, m);
  else v = 0.;
  return (3. / 2.) * v;
}
inline double q_kl2_h(uint64_t k, uint64_t l, int64_t m) {  // gaunt.rs:39-44
  double lk = double(k * (k + 1)), ll = double(l * (l + 1));
  return q_kl2(k, l, m) * ((lk + ll) / 2. - 3.);
}
inline double q_kl2_hd(uint64_t k, uint64_t l, int64_t m) {  // gaunt.rs:53-58
  double lk = double(k * (k + 1)), ll = double(l * (l + 1));
  return q_kl2(k, l, m) * ((ll - lk) / 2. + 3.);
}

struct PerturbedMetric { std::vector<double> alpha, dalpha, ddalpha, beta, dbeta, ddbeta; };  // model.rs:104-123
struct ModeToPerturb {  // perturbed.rs:26-33; the post-processing members perturb_deformed reads
  uint64_t ell;
  double freq;
  const double *xi_r, *xi_h, *xi_tn, *xi_tp, *psi, *rho, *p;
};
enum CouplingMatrix { CM_D_SQUARED, CM_D_LINEAR, CM_D_ZERO, CM_R_LINEAR, CM_R_ZERO, CM_L_ZERO, CM_COUNT };

// The frequency-independent radial weights of perturb_deformed (src/perturbed.rs:74-108).
struct CouplingWeights {
  std::vector<double> trapezoid, te_a, adte_a, te_b, adte_b;
  CouplingWeights(const DiscreteModel& model, const PerturbedMetric& pm) {
    const size_t np = model.r_coord.size();
    trapezoid.assign(np, 0.);
    trapezoid[0] = 0.5 * (model.r_coord[1] - model.r_coord[0]);
    trapezoid[np - 1] = 0.5 * (model.r_coord[np - 1] - model.r_coord[np - 2]);
    for (size_t i = 1; i + 1 < np; ++i) trapezoid[i] = 0.5 * (model.r_coord[i + 1] - model.r_coord[i - 1]);
    te_a.resize(np); adte_a.resize(np); te_b.resize(np); adte_b.resize(np);
    for (size_t i = 0; i < np; ++i) {
      te_a[i] = 3. * pm.alpha[i] + pm.dalpha[i];
      adte_a[i] = 3. * pm.dalpha[i] + pm.ddalpha[i];
      te_b[i] = 3. * pm.beta[i] + pm.dbeta[i];
      adte_b[i] = 3. * pm.dbeta[i] + pm.ddbeta[i];
    }
  }
};

// One (left, right) element of the six matrices: the body of the double loop, src/perturbed.rs:122-263.
// Throws where the reference asserts finiteness.
template <class T = double>
inline void coupling_pair(const DiscreteModel& model, const CouplingWeights& w, const PerturbedMetric& pm,
                          const ModeToPerturb& l, const ModeToPerturb& r, int64_t m, T (&out)[CM_COUNT]) {
  const size_t np = model.r_coord.size();
  const std::vector<double>&trapezoid = w.trapezoid, &te_a = w.te_a, &adte_a = w.adte_a, &te_b = w.te_b, &adte_b = w.adte_b;
  const T mf = T(double(m));
  {
    {
      const T q2 = T(q_kl2(l.ell, r.ell, m)), q2_hrd = T(q_kl2_hd(l.ell, r.ell, m)), q2_hld = T(q_kl2_hd(r.ell, l.ell, m)),
              q2_h = T(q_kl2_h(l.ell, r.ell, m));
      T q_rt_p = T(0.), q_rt_n = T(0.), q_ht_p = T(0.), q_ht_n = T(0.);
      if (l.ell == r.ell) {
        const uint64_t n1 = r.ell == 0 ? 0 : r.ell - 1;
        q_rt_p = T(q_kl1_hd(l.ell, r.ell + 1, m));
        q_rt_n = T(q_kl1_hd(l.ell, n1, m));
        q_ht_p = T(-q_kl1_h(l.ell, r.ell + 1, m));
        q_ht_n = T(-q_kl1_h(l.ell, n1, m));
      }
      const T ipr = T(l.ell == r.ell ? 1. : 0.), iph = T(l.ell == r.ell ? double(r.ell * (r.ell + 1)) : 0.);  // :17-23
      T d_squared = T(0.), d_linear = T(0.), d_zero = T(0.), r_linear = T(0.), r_zero = T(0.), l_zero = T(0.);
      for (size_t rc = 1; rc < np; ++rc) {  // :148-261
        const T lxr = T(l.xi_r[rc]), lxh = T(l.xi_h[rc]), lpsi = T(l.psi[rc]), lp = T(l.p[rc]);
        const T rxr = T(r.xi_r[rc]), rxh = T(r.xi_h[rc]), rtpv = T(r.xi_tp[rc]), rtnv = T(r.xi_tn[rc]), rrho = T(r.rho[rc]);
        const T rr = lxr * rxr, rh = lxr * rxh, hr = lxh * rxr, hh = lxh * rxh;
        const T rtp = lxr * rtpv, rtn = lxr * rtnv, htp = lxh * rtpv, htn = lxh * rtnv;
        const T x = T(model.r_coord[rc]), rho = T(model.rho[rc]), rot = T(model.rot[rc]), trap = T(trapezoid[rc]);
        const T ea = T(pm.alpha[rc]), adea = T(pm.dalpha[rc]), eb = T(pm.beta[rc]), adeb = T(pm.dbeta[rc]);
        const T tea = T(te_a[rc]), adtea = T(adte_a[rc]), teb = T(te_b[rc]), adteb = T(adte_b[rc]);
        const T vol = trap * (x * x) * rho;
        T val = vol * (rr * ipr * (T(1.) + T(2.) * (ea + adea)) + hh * iph * (T(1.) + T(2.) * ea) +
                       rr * T(2.) * q2 * (eb + adeb) + (rh * q2_hrd + hr * q2_hld + hh * T(2.) * q2_h) * eb);
        if (!std::isfinite(to_double(val))) throw std::runtime_error("Not finite number (d)");
        d_squared += val;
        d_linear += T(2.) * mf * rot * val;
        d_zero += mf * mf * rot * rot * val;
        val = vol * T(2.) *
              (ipr * ((hr + rh) * (T(1.) + T(2.) * ea + adea) + hh * (T(1.) + T(2.) * ea)) +
               (rh + hr) * q2 * (T(2.) * eb + adeb) + hh * (ipr + T(6.) * q2) * T(2.) * eb);
        const T val_t = vol * T(2.) * rot * (q_rt_p * rtp + q_rt_n * rtn + q_ht_n * htn + q_ht_p * htp);
        if (!std::isfinite(to_double(val)) || !std::isfinite(to_double(val_t))) throw std::runtime_error("Not finite number (r)");
        r_linear += mf * rot * val + val_t;
        r_zero += mf * rot * (mf * rot * val + val_t);
        const T sig = T(r.freq) - mf * rot;
        l_zero += trap * (x * x) *
                  ((sig * sig) * rho * (ipr * rr + iph * hh) + T(2.) * mf * rot * sig * rho * ipr * (rh + hr + hh) +
                   lpsi * rrho * tea - rho * lpsi * rxr * adtea / x + lpsi * q2 * rrho * teb -
                   rho * lpsi * q2 * rxr * adteb / x - lp * rxr * adtea / x - lp * q2 * rxr * adteb / x -
                   lp * q2_hrd * rxh * teb / x + q2_hrd * rho * lpsi * rxh * teb / x);
        if (!std::isfinite(to_double(l_zero))) throw std::runtime_error("Not finite number (l)");
      }
      out[CM_D_SQUARED] = d_squared; out[CM_D_LINEAR] = d_linear; out[CM_D_ZERO] = d_zero;
      out[CM_R_LINEAR] = r_linear; out[CM_R_ZERO] = r_zero; out[CM_L_ZERO] = l_zero;
    }
  }
}

// Fills out[CM_COUNT][M*M], each column-major like nalgebra's DMatrix: (left, right) at left + right*M.
inline void coupling_matrices(const DiscreteModel& model, const std::vector<ModeToPerturb>& modes, int64_t m,
                              const PerturbedMetric& pm, double* const out[CM_COUNT], int threads = 1) {
  const size_t M = modes.size();
  const CouplingWeights w(model, pm);
  std::vector<std::string> errs(M);
  parallel_for(M, threads, [&](size_t lm) {
    try {
      for (size_t rm = 0; rm < M; ++rm) {
        double v[CM_COUNT];
        coupling_pair<double>(model, w, pm, modes[lm], modes[rm], m, v);
        for (int c = 0; c < CM_COUNT; ++c) out[c][lm + rm * M] = v[c];
      }
    } catch (const std::exception& e) { errs[lm] = e.what(); }
  });
  for (const std::string& e : errs)
    if (!e.empty()) throw std::runtime_error(e);
}

// perturb_structure, src/perturbed.rs:324-460: the deformation of the equilibrium structure by uniform rotation.
// Used here only to give the coupling-integral tests realistic metric arrays (it stays on the host in the
// reference).  nalgebra's closed-form 2x2 / 3x3 try_inverse is restated from memory of the
// crate and pinned by nothing: parity unpinned.
inline bool inv3(const Mat<double, 3>& a, Mat<double, 3>& o) {
  const double m11 = a(0, 0), m12 = a(0, 1), m13 = a(0, 2), m21 = a(1, 0), m22 = a(1, 1), m23 = a(1, 2), m31 = a(2, 0),
               m32 = a(2, 1), m33 = a(2, 2);
  const double minor_12_23 = m22 * m33 - m32 * m23, minor_11_23 = m21 * m33 - m31 * m23, minor_11_22 = m21 * m32 - m31 * m22;
  const double det = m11 * minor_12_23 - m12 * minor_11_23 + m13 * minor_11_22;
  if (det == 0.) return false;
  o(0, 0) = minor_12_23 / det; o(0, 1) = (m13 * m32 - m33 * m12) / det; o(0, 2) = (m12 * m23 - m22 * m13) / det;
  o(1, 0) = -minor_11_23 / det; o(1, 1) = (m11 * m33 - m31 * m13) / det; o(1, 2) = (m13 * m21 - m23 * m11) / det;
  o(2, 0) = minor_11_22 / det; o(2, 1) = (m12 * m31 - m32 * m11) / det; o(2, 2) = (m11 * m22 - m21 * m12) / det;
  return true;
}
inline bool inv2(const Mat<double, 2>& a, Mat<double, 2>& o) {
  const double det = a(0, 0) * a(1, 1) - a(1, 0) * a(0, 1);
  if (det == 0.) return false;
  o(0, 0) = a(1, 1) / det; o(0, 1) = -a(0, 1) / det; o(1, 0) = -a(1, 0) / det; o(1, 1) = a(0, 0) / det;
  return true;
}
inline PerturbedMetric perturb_structure(const DiscreteModel& model, double rot, double* mass_delta_out = nullptr) {
  const size_t np = model.r_coord.size();
  const std::vector<double>& x = model.r_coord;
  std::vector<double> y0x(np, 0.), y0y(np, 0.), y2x(np, 0.), y2y(np, 0.);
  if (np > 1) { y2x[1] = 1.; y2y[1] = 2.; }
  auto kcoef = [&](size_t i) {
    return 4. * M_PI / model.m_coord[i] * model.rho[i] * x[i] * (-model.a_star[i] - model.v[i] / model.gamma1[i]);
  };
  for (size_t i = 1; i < np; ++i) {
    const double delta = x[i] - x[i - 1], x_12 = 0.5 * (x[i] + x[i - 1]);
    const double k = kcoef(i), k_prev = i == 1 ? 0. : kcoef(i - 1);
    const double sc = 0.5 * delta / x_12;
    Mat<double, 3> a0 = Mat<double, 3>::zeros();
    a0(0, 0) = -1.; a0(0, 1) = 1.; a0(0, 2) = 0.;
    a0(1, 0) = 0.5 * (x_12 * x_12) * (k + k_prev); a0(1, 1) = -2.; a0(1, 2) = x_12;
    for (double& e : a0.a) e = sc * e;
    Mat<double, 3> inv;
    if (!inv3(sub(Mat<double, 3>::identity(), a0), inv)) throw std::runtime_error("perturb_structure: singular step");
    const Mat<double, 3> step = mul(inv, add(Mat<double, 3>::identity(), a0));
    // step[0..2,0..2] * y0[i-1] + step[0..2,2]  (gemv: column axpy, k ascending)
    const double px = y0x[i - 1], py = y0y[i - 1];
    y0x[i] = (step(0, 1) * py + step(0, 0) * px) + step(0, 2);
    y0y[i] = (step(1, 1) * py + step(1, 0) * px) + step(1, 2);
    if (i != 1) {
      Mat<double, 2> a2;
      a2(0, 0) = -1.; a2(0, 1) = 1.; a2(1, 0) = 6. + 0.5 * (x_12 * x_12) * (k + k_prev); a2(1, 1) = -2.;
      for (double& e : a2.a) e = sc * e;
      Mat<double, 2> inv_2;
      if (!inv2(sub(Mat<double, 2>::identity(), a2), inv_2)) throw std::runtime_error("perturb_structure: singular step");
      const Mat<double, 2> st2 = mul(inv_2, add(Mat<double, 2>::identity(), a2));
      const double qx = y2x[i - 1], qy = y2y[i - 1];
      y2x[i] = st2(0, 1) * qy + st2(0, 0) * qx;
      y2y[i] = st2(1, 1) * qy + st2(1, 0) * qx;
    }
  }
  const double a2c = -5. / 6. / (3. * y2x[np - 1] + y2y[np - 1]);
  PerturbedMetric pm;
  for (auto* v : {&pm.alpha, &pm.dalpha, &pm.ddalpha, &pm.beta, &pm.dbeta, &pm.ddbeta}) v->assign(np, 0.);
  std::vector<double> trapezoid(np, 0.);
  trapezoid[0] = 0.5 * (x[1] - x[0]);
  trapezoid[np - 1] = 0.5 * (x[np - 1] - x[np - 2]);
  for (size_t i = 1; i + 1 < np; ++i) trapezoid[i] = 0.5 * (x[i + 1] - x[i - 1]);
  double mass_delta = 0.;
  const double rot2 = rot * rot;
  for (size_t i = 1; i < np; ++i) {
    const double dln = -model.a_star[i] - model.v[i] / model.gamma1[i];
    const double k = 4. * M_PI / model.m_coord[i] * model.rho[i] * x[i] * dln;
    const double u = model.u[i];
    auto fill = [&](double psi, double dpsi, double ddpsi, double& f, double& df, double& ddf) {
      f = 2. * x[i] / model.m_coord[i] * psi * rot2;
      df = f * (1. - u + x[i] * dpsi / psi);
      ddf = df * (1. - u + x[i] * dpsi / psi) +
            f * (-3. * u - u * dln + u * u + x[i] * dpsi / psi - (x[i] * dpsi / psi) * (x[i] * dpsi / psi) +
                 (x[i] * x[i]) * ddpsi / psi);
    };
    fill(y0x[i] * x[i], y0y[i], ((x[i] * x[i]) * k * y0x[i] - 2. * y0y[i]) / x[i] + 1., pm.alpha[i], pm.dalpha[i], pm.ddalpha[i]);
    fill(y2x[i] * x[i] * a2c, y2y[i] * a2c, ((6. + (x[i] * x[i]) * k) * y2x[i] - 2. * y2y[i]) / x[i] * a2c, pm.beta[i], pm.dbeta[i],
         pm.ddbeta[i]);
    mass_delta += trapezoid[i] * model.rho[i] * (x[i] * x[i]) * (3. * pm.alpha[i] + pm.dalpha[i]);
  }
  if (mass_delta_out) *mass_delta_out = mass_delta * (4. * M_PI);
  return pm;
}

}  // namespace storm_ref
