// device_system.hpp -- the discretised Rotating1D system as the kernels see it.
//
// HBM layout (built once per solver by storm_solver_create, see storm_hip.cpp):
//   tab[field][point j][interval i]   FP64, field-major SoA, kNumFields planes of `plane` doubles,
//                                     element (j, i) of a plane at j*stride + i (stride = n_int padded to
//                                     a multiple of 16, plus 16: rows start 128-B aligned).
// A wave whose lanes walk consecutive intervals reads each plane fully coalesced; a wave whose
// lanes share the interval (lane-per-frequency kernels) reads it through the scalar unit.
// The planes hold the frequency-independent part of the 4x4 coefficient matrix
// (reference Rotating1D::eval, src/system/adiabatic.rs:78-114, which the reference also caches
// per collocation point, src/system/discretized.rs:102-111) plus the three products the
// frequency-dependent terms need (src/system/adiabatic.rs:116-152).
#pragma once
#include "mat4.hpp"

namespace STORM_NS {
using namespace storm;

// Frequency-dependent scalars of add_frequency.  With a uniform m*rot they are the same at
// every collocation point, so they are evaluated once per frequency (same operations, same
// operands => same bits as evaluating them per point like the reference does).
struct FreqTerms {
  double f2;       // ell == 0: f*f
  double w_rsq;    // (lambda*(f-rot) + 2 rot) * (f-rot)
  double rel_rot;  // 2 rot / (lambda*(f-rot) + 2 rot)
  double fr2;      // (f-rot)^2
  double w;        // 1 - (2 rot)^2 / w_rsq
};

template <bool ELL0>
SDEV FreqTerms freq_terms(double f, double rotm, double lambda) {
  FreqTerms t;
  if constexpr (ELL0) {
    t.f2 = f * f;
    t.w_rsq = t.rel_rot = t.fr2 = t.w = 0.0;
  } else {
    const double fr = f - rotm;
    const double two_rot = 2.0 * rotm;
    const double den = lambda * fr + two_rot;
    t.w_rsq = den * fr;
    t.rel_rot = two_rot / den;
    t.fr2 = fr * fr;
    t.w = 1.0 - (two_rot * two_rot) / t.w_rsq;
    t.f2 = 0.0;
  }
  return t;
}

// Where a kernel takes its coefficients from: straight from the planes in global memory ...
struct GlobalCoef {
  const double* __restrict__ base;  // tab + i
  size_t plane, stride;
  SDEV double operator()(int field, int j) const { return base[field * plane + j * stride]; }
};
// ... from the interval-major copy of a single solver (types.hpp, aos_pitch): base = its row of this lane's interval ...
template <int P>
struct AosCoef {
  const double* __restrict__ base;
  SDEV double operator()(int field, int j) const { return base[field * P + j]; }
};
// ... or from a tile the wave staged in LDS: rows (field*P + j) of W intervals, this lane's interval `sub`.
template <int P, int W>
struct LdsCoef {
  const double* base;  // tile + sub
  SDEV double operator()(int field, int j) const { return base[(field * P + j) * W]; }
};

// A_j(omega) of collocation point j of this lane's interval: cached part + add_frequency.
template <bool ELL0, bool UNIFORM, class Coef>
SDEV M4 point_matrix(const Coef& t, int j, double lambda, double f, const FreqTerms& ut) {
  M4 a;
#pragma unroll
  for (int e = 0; e < 16; ++e) a.v[e] = 0.0;
  const double d = t(F_D, j);
  const double dc1 = t(F_DC1, j);
  const double m10 = t(F_M10, j);
  a.v[0 * 4 + 3] = t(F_M30, j);
  a.v[1 * 4 + 3] = t(F_M31, j);
  a.v[2 * 4 + 2] = t(F_M22, j);
  a.v[2 * 4 + 3] = t(F_M32, j);
  a.v[3 * 4 + 1] = -d;
  a.v[3 * 4 + 2] = d;
  a.v[3 * 4 + 3] = t(F_M33, j);
  if constexpr (ELL0) {
    a.v[0 * 4 + 0] = t(F_M00, j);
    a.v[1 * 4 + 0] = t(F_M01, j);
    a.v[1 * 4 + 1] = t(F_M11, j);
    a.v[0 * 4 + 1] = m10 + dc1 * ut.f2;
  } else {
    FreqTerms ft = ut;
    if constexpr (!UNIFORM) ft = freq_terms<false>(f, t(F_ROTM, j), lambda);
    const double dl = a.v[2 * 4 + 3];
    const double q = dl * ft.rel_rot;
#ifdef STORM_FAST_MATH
    const double x = t(F_DL2, j) * fast_recip_guarded(ft.w_rsq * dc1);
#else
    const double x = t(F_DL2, j) / (ft.w_rsq * dc1);
#endif
    a.v[0 * 4 + 0] = t(F_M00, j) - q;  // += (-dl)*rel_rot
    a.v[1 * 4 + 0] = t(F_M01, j) + x;
    a.v[2 * 4 + 0] = 0.0 + x;
    a.v[1 * 4 + 1] = t(F_M11, j) + q;
    a.v[2 * 4 + 1] = 0.0 + q;
    a.v[0 * 4 + 1] = m10 + dc1 * ft.fr2 * ft.w;
  }
  return a;
}

// ---------------------------------------------------------------------------------------
// Boundary rows and the closing elimination of the discriminant
// (reference determinant_explicit, src/solver.rs:435-542, with row_pivot!/sweep!, :163-269).
// B[r][c]: system row r (2 inner rows, then the 4 rows [accum | -I], later the 2 outer rows),
// system column c.  All indices are compile-time; the data-dependent row swap is done with
// selects so nothing is runtime-indexed.
// ---------------------------------------------------------------------------------------
template <int ROWS, int COLS, int K>
SDEV void pivot_and_sweep(double (&B)[6][8], double& det) {
  int max_idx = K;
  double max_val = __builtin_fabs(B[K][K]);
#pragma unroll
  for (int i = K + 1; i < ROWS; ++i) {
    const double v = __builtin_fabs(B[i][K]);
    if (v > max_val) { max_idx = i; max_val = v; }
  }
  double prow[COLS];
#pragma unroll
  for (int j = 0; j < COLS; ++j) prow[j] = B[K][j];
#pragma unroll
  for (int i = K + 1; i < ROWS; ++i) {
    const bool sel = (i == max_idx);
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
      const double old = B[i][j];
      prow[j] = sel ? old : prow[j];
      B[i][j] = sel ? B[K][j] : old;
    }
  }
  const double pivot = prow[K];
  if (max_idx != K) det = det * -1.0;
  det = det * pivot;
  const double pinv = 1.0 / pivot;
#pragma unroll
  for (int r = K + 1; r < ROWS; ++r) {
    const double m = B[r][K] * pinv;
#pragma unroll
    for (int j = 0; j < COLS; ++j) B[r][j] = B[r][j] - prow[j] * m;
  }
}

template <bool ELL0>
SDEV double inner_00(const SystemDev& s, double f) {  // reference src/system/adiabatic.rs:164-173
  if constexpr (ELL0) {
    return s.inner_c1 * f * f;
  } else {
    const double rot = s.inner_rotm;
    const double omega2 = (f - rot + 2.0 * rot / s.ell) * (f - rot);
    return s.inner_c1 * omega2;
  }
}

template <bool ELL0>
SDEV double close_determinant(const SystemDev& s, double f, const M4& accum) {
  double B[6][8];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) B[r][c] = 0.0;
  B[0][0] = inner_00<ELL0>(s, f);
  B[0][1] = -s.ell; B[0][2] = -s.ell;
  B[1][2] = s.ell;  B[1][3] = -1.0;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      B[r + 2][c] = accum.v[c * 4 + r];
      B[r + 2][c + 4] = (r == c) ? -1.0 : 0.0;
    }
  double det = 1.0;
  pivot_and_sweep<6, 8, 0>(B, det);
  pivot_and_sweep<6, 8, 1>(B, det);
  pivot_and_sweep<6, 8, 2>(B, det);
  pivot_and_sweep<6, 8, 3>(B, det);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { B[i][j] = B[i + 4][j + 4]; B[i][j + 4] = 0.0; }
  // outer boundary (reference src/system/adiabatic.rs:185-208)
  B[2][0] = 1.0; B[2][1] = -1.0; B[2][2] = 0.0; B[2][3] = 0.0;
  B[3][0] = s.outer_u; B[3][1] = 0.0; B[3][2] = s.ell + 1.0; B[3][3] = 1.0;
  pivot_and_sweep<4, 4, 0>(B, det);
  pivot_and_sweep<4, 4, 1>(B, det);
  pivot_and_sweep<4, 4, 2>(B, det);
  return det * B[3][3];
}

}  // namespace STORM_NS
