// scheme_kernels.hpp -- the two kernels that evaluate step matrices, templated on the
// difference scheme.  Each scheme is instantiated in its own translation unit
// (scheme_tu.hip compiled 8 times with -DSTORM_SCHEME=n) so the heavy, fully unrolled
// bodies build in parallel.
//
//   det_fused_kernel   lane = frequency.  Walks all intervals, builds the step matrix in
//                      registers and multiplies it into the lane's running product
//                      (reference determinant_explicit, src/solver.rs:486-492), then closes
//                      with the boundary elimination.  Coefficients are wave-uniform reads.
//                      Used when there are enough frequencies to fill the chip.
//   build_steps_kernel lane = (frequency, interval) pair, intervals fastest so the coefficient
//                      planes are read coalesced.  Writes the 16 entries of the step matrix,
//                      row-major, to steps[f][i][16].  Feeds the ordered-product kernel (small
//                      frequency batches: bracket refinement) and the eigenvector elimination.
#pragma once
#include "device_system.hpp"
#include "steppers.hpp"

namespace storm {

template <int SCHEME, bool ELL0, bool UNIFORM>
SDEV M4 interval_step(const SystemDev& s, size_t i, double f, const FreqTerms& ut, bool& ok) {
  constexpr int P = scheme_points(SCHEME);
  M4 a[P];
#pragma unroll
  for (int j = 0; j < P; ++j) a[j] = point_matrix<ELL0, UNIFORM>(s, (size_t)j * s.n_int + i, f, ut);
  return step_matrix<SCHEME>(a, ok);
}

template <int SCHEME, bool ELL0, bool UNIFORM>
__global__ __launch_bounds__(64) void det_fused_kernel(SystemDev s, const double* __restrict__ freq, size_t n,
                                                       double* __restrict__ det_out, int* __restrict__ flags) {
  const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x;
  const double f = freq[t < n ? t : n - 1];
  const FreqTerms ut = freq_terms<ELL0>(f, s.rotm_uniform, s.lambda);
  bool ok = true;
  M4 acc = m4_identity();
  for (size_t i = 0; i < s.n_int; ++i) {
    const M4 l = interval_step<SCHEME, ELL0, UNIFORM>(s, i, f, ut, ok);
    acc = m4_mul(l, acc);
  }
  const double det = close_determinant<ELL0>(s, f, acc);
  if (t < n) {
    det_out[t] = ok ? det : __longlong_as_double(0x7ff8000000000000LL);
    if (!ok) atomicOr(flags, 1);
  }
}

// quad-cooperative variant: 4 lanes per frequency.  Lane q of a quad builds the step matrix of
// interval 4t+q, then the four matrices are multiplied into the running product in interval order;
// lane q owns column q of the product and receives each step matrix from its quad over DPP.
// A column of L*acc is evaluated exactly like one column of the reference's gemm, so the result is
// bit-identical to the lane-per-frequency kernel while exposing 4x more waves.
template <int K>
SDEV double quad_bcast_d(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, K * 0x55, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, K * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

template <int K>
SDEV void quad_apply(const M4& mine, double (&col)[4]) {
  double l[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) l[e] = quad_bcast_d<K>(mine.v[e]);
  double c[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = l[r] * col[0];
#pragma unroll
  for (int k = 1; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = l[k * 4 + r] * col[k] + c[r];
#pragma unroll
  for (int r = 0; r < 4; ++r) col[r] = c[r];
}

template <int SCHEME, bool ELL0, bool UNIFORM>
__global__ __launch_bounds__(64) void det_quad_kernel(SystemDev s, const double* __restrict__ freq, size_t n,
                                                      double* __restrict__ det_out, int* __restrict__ flags) {
  const int q = threadIdx.x & 3;
  const size_t t = (size_t)blockIdx.x * 16 + (threadIdx.x >> 2);
  const double f = freq[t < n ? t : n - 1];
  const FreqTerms ut = freq_terms<ELL0>(f, s.rotm_uniform, s.lambda);
  bool ok = true;
  double col[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) col[r] = (r == q) ? 1.0 : 0.0;
  const size_t n_int = s.n_int;
  for (size_t i0 = 0; i0 < n_int; i0 += 4) {
    const size_t i = i0 + q;
    const M4 l = interval_step<SCHEME, ELL0, UNIFORM>(s, i < n_int ? i : n_int - 1, f, ut, ok);
    quad_apply<0>(l, col);
    if (i0 + 1 < n_int) quad_apply<1>(l, col);
    if (i0 + 2 < n_int) quad_apply<2>(l, col);
    if (i0 + 3 < n_int) quad_apply<3>(l, col);
  }
  M4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    acc.v[0 * 4 + r] = quad_bcast_d<0>(col[r]);
    acc.v[1 * 4 + r] = quad_bcast_d<1>(col[r]);
    acc.v[2 * 4 + r] = quad_bcast_d<2>(col[r]);
    acc.v[3 * 4 + r] = quad_bcast_d<3>(col[r]);
  }
  // a singular stage matrix anywhere in the quad poisons the frequency
  const unsigned long long bad = __ballot(!ok);
  const bool quad_ok = ((bad >> (threadIdx.x & 60)) & 0xfULL) == 0;
  const double det = close_determinant<ELL0>(s, f, acc);
  if (q == 0 && t < n) {
    det_out[t] = quad_ok ? det : __longlong_as_double(0x7ff8000000000000LL);
    if (!quad_ok) atomicOr(flags, 1);
  }
}

template <int SCHEME, bool ELL0, bool UNIFORM>
__global__ __launch_bounds__(256) void build_steps_kernel(SystemDev s, const double* __restrict__ freq, size_t n,
                                                          double* __restrict__ steps, int* __restrict__ flags) {
  // grid.x covers intervals in tiles of 256, grid.y = frequency
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t fi = blockIdx.y;
  if (fi >= n) return;
  const double f = freq[fi];
  const FreqTerms ut = freq_terms<ELL0>(f, s.rotm_uniform, s.lambda);
  if (i >= s.n_int) return;
  bool ok = true;
  const M4 l = interval_step<SCHEME, ELL0, UNIFORM>(s, i, f, ut, ok);
  double2* out = reinterpret_cast<double2*>(steps + (fi * s.n_int + i) * 16);
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // row-major: row r = {L(r,0..3)}
    out[r * 2 + 0] = ok ? make_double2(l.v[0 * 4 + r], l.v[1 * 4 + r]) : make_double2(nan, nan);
    out[r * 2 + 1] = ok ? make_double2(l.v[2 * 4 + r], l.v[3 * 4 + r]) : make_double2(nan, nan);
  }
  if (!ok) atomicOr(flags, 1);
}

// Host-side launchers, one pair per scheme TU (defined in scheme_tu.hip).
struct LaunchArgs {
  SystemDev sys;
  const double* freq;  // device
  size_t n;
  double* out;         // det_out (fused) or steps (build), device
  int* flags;          // device
  bool ell0, uniform;
  hipStream_t stream;
};
typedef hipError_t (*launch_fn)(const LaunchArgs&);
extern launch_fn g_launch_fused[kNumSchemes];
extern launch_fn g_launch_quad[kNumSchemes];
extern launch_fn g_launch_build[kNumSchemes];

}  // namespace storm
