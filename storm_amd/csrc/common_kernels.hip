// common_kernels.hip -- scheme-independent kernels.
//
//   chain_kernel        ordered product of the step matrices of one frequency followed by the
//                       boundary elimination.  16 lanes per frequency: lane (r, c) owns element
//                       (r, c) of the running product; the 4 values of column c it needs are
//                       fetched from its quad with DPP.  Every element is accumulated in exactly
//                       the reference's order (reference src/solver.rs:486-492 + nalgebra gemm),
//                       so the result is bit-identical to the lane-per-frequency kernel.
//   eigen_forward_kernel  block-bidiagonal elimination with row pivoting that records the packed
//                       upper factor (reference determinant_inner/UpperResult, src/solver.rs:
//                       86-161,271-433).  One lane per mode.
//   eigen_backsub_kernel  back-substitution to the eigenvector (reference src/solver.rs:100-131).
#include "common_kernels.hpp"

namespace storm {

template <int K>
SDEV double quad_bcast(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, K * 0x55, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, K * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

constexpr int kChainUnroll = 4;

template <bool ELL0>
__global__ __launch_bounds__(64) void chain_kernel(SystemDev s, const double* __restrict__ freq, size_t n,
                                                   const double* __restrict__ steps, double* __restrict__ det_out) {
  __shared__ double sh[64];
  const int tid = threadIdx.x;
  const int sub = tid & 15;
  const int r = sub & 3;   // row of the running product owned by this lane
  const int c = sub >> 2;  // column (= quad index inside the 16-lane group)
  const size_t fi_raw = (size_t)blockIdx.x * 4 + (tid >> 4);
  const size_t fi = fi_raw < n ? fi_raw : n - 1;
  const double2* __restrict__ rows = reinterpret_cast<const double2*>(steps + fi * s.n_int * 16) + r * 2;
  double acc = (r == c) ? 1.0 : 0.0;
  const size_t n_int = s.n_int;
  size_t i = 0;
  for (; i + kChainUnroll <= n_int; i += kChainUnroll) {
    double2 lo[kChainUnroll], hi[kChainUnroll];
#pragma unroll
    for (int u = 0; u < kChainUnroll; ++u) {
      lo[u] = rows[(i + u) * 8];
      hi[u] = rows[(i + u) * 8 + 1];
    }
#pragma unroll
    for (int u = 0; u < kChainUnroll; ++u) {
      const double a0 = quad_bcast<0>(acc), a1 = quad_bcast<1>(acc), a2 = quad_bcast<2>(acc), a3 = quad_bcast<3>(acc);
      double t = lo[u].x * a0;
      t = lo[u].y * a1 + t;
      t = hi[u].x * a2 + t;
      t = hi[u].y * a3 + t;
      acc = t;
    }
  }
  for (; i < n_int; ++i) {
    const double2 lo = rows[i * 8], hi = rows[i * 8 + 1];
    const double a0 = quad_bcast<0>(acc), a1 = quad_bcast<1>(acc), a2 = quad_bcast<2>(acc), a3 = quad_bcast<3>(acc);
    double t = lo.x * a0;
    t = lo.y * a1 + t;
    t = hi.x * a2 + t;
    t = hi.y * a3 + t;
    acc = t;
  }
  sh[tid] = acc;  // element (r, c) at group*16 + c*4 + r: column-major, as M4 wants it
  __syncthreads();
  if (sub == 0 && fi_raw < n) {
    M4 a;
#pragma unroll
    for (int e = 0; e < 16; ++e) a.v[e] = sh[(tid & 48) + e];
    det_out[fi_raw] = close_determinant<ELL0>(s, freq[fi_raw], a);
  }
}

hipError_t launch_chain(const SystemDev& s, bool ell0, const double* freq, size_t n, const double* steps,
                        double* det_out, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 3) / 4);
  if (ell0) chain_kernel<true><<<blocks, 64, 0, stream>>>(s, freq, n, steps, det_out);
  else chain_kernel<false><<<blocks, 64, 0, stream>>>(s, freq, n, steps, det_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Eigenvector: forward elimination.
// ---------------------------------------------------------------------------------------
// One pivot of the 6x8 band (or of the closing 4x4).  COLPIV selects the column-pivot variant
// used for the first two pivots of the first interval (reference column_pivot!, solver.rs:192-230).
// `up` receives prow[i]*pinv for i = K+1..COLS-1 (reference sweep! first arm, solver.rs:164-174).
template <int ROWS, int COLS, int K, bool COLPIV>
SDEV void pivot_sweep_upper(double (&B)[6][8], double& det, double* __restrict__ up, int* swap_out) {
  double prow[COLS];
  if constexpr (COLPIV) {
    int max_idx = K;
    double max_val = __builtin_fabs(B[K][K]);
#pragma unroll
    for (int i = K + 1; i < COLS; ++i) {
      const double v = __builtin_fabs(B[K][i]);
      if (v > max_val) { max_idx = i; max_val = v; }
    }
    // swap system columns max_idx and K in rows K..ROWS-1
#pragma unroll
    for (int i = K + 1; i < COLS; ++i) {
      const bool sel = (i == max_idx);
#pragma unroll
      for (int rr = K; rr < ROWS; ++rr) {
        const double a = B[rr][i], b = B[rr][K];
        B[rr][i] = sel ? b : a;
        B[rr][K] = sel ? a : b;
      }
    }
    if (max_idx != K) det = det * -1.0;
    *swap_out = max_idx;  // == K when no swap happened
#pragma unroll
    for (int j = 0; j < COLS; ++j) prow[j] = B[K][j];
  } else {
    int max_idx = K;
    double max_val = __builtin_fabs(B[K][K]);
#pragma unroll
    for (int i = K + 1; i < ROWS; ++i) {
      const double v = __builtin_fabs(B[i][K]);
      if (v > max_val) { max_idx = i; max_val = v; }
    }
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
    if (max_idx != K) det = det * -1.0;
  }
  const double pivot = prow[K];
  det = det * pivot;
  const double pinv = 1.0 / pivot;
#pragma unroll
  for (int i = K + 1; i < COLS; ++i) up[i - K - 1] = prow[i] * pinv;
#pragma unroll
  for (int rr = K + 1; rr < ROWS; ++rr) {
    const double m = B[rr][K] * pinv;
#pragma unroll
    for (int j = 0; j < COLS; ++j) B[rr][j] = B[rr][j] - prow[j] * m;
  }
}

template <bool FIRST>
SDEV void eliminate_interval(double (&B)[6][8], double& det, const double* __restrict__ lrow,
                             double* __restrict__ up, int* swaps) {
  // rows 2..5 <- [L | -I]; lrow is the row-major step matrix
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      B[r + 2][c] = lrow[r * 4 + c];
      B[r + 2][c + 4] = (r == c) ? -1.0 : 0.0;
    }
  int dummy;
  pivot_sweep_upper<6, 8, 0, FIRST>(B, det, up + 0, FIRST ? &swaps[0] : &dummy);
  pivot_sweep_upper<6, 8, 1, FIRST>(B, det, up + 7, FIRST ? &swaps[1] : &dummy);
  pivot_sweep_upper<6, 8, 2, false>(B, det, up + 13, &dummy);
  pivot_sweep_upper<6, 8, 3, false>(B, det, up + 18, &dummy);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { B[i][j] = B[i + 4][j + 4]; B[i][j + 4] = 0.0; }
}

template <bool ELL0>
__global__ __launch_bounds__(64) void eigen_forward_kernel(SystemDev s, const double* __restrict__ freq, size_t n,
                                                           const double* __restrict__ steps,
                                                           double* __restrict__ upper, int* __restrict__ swaps,
                                                           double* __restrict__ det_out) {
  const size_t m = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (m >= n) return;
  const double f = freq[m];
  const size_t n_int = s.n_int;
  const size_t ustride = n_int * 22 + 6;
  double* __restrict__ up = upper + m * ustride;
  const double* __restrict__ st = steps + m * n_int * 16;
  double B[6][8];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) B[r][c] = 0.0;
  B[0][0] = inner_00<ELL0>(s, f);
  B[0][1] = -s.ell; B[0][2] = -s.ell;
  B[1][2] = s.ell;  B[1][3] = -1.0;
  double det = 1.0;
  int sw[2] = {0, 1};
  if (n_int > 0) {
    double l[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) l[e] = st[e];
    eliminate_interval<true>(B, det, l, up, sw);
  }
  for (size_t i = 1; i < n_int; ++i) {
    double l[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) l[e] = st[i * 16 + e];
    eliminate_interval<false>(B, det, l, up + i * 22, sw);
  }
  swaps[m * 2 + 0] = sw[0];
  swaps[m * 2 + 1] = sw[1];
  // outer boundary rows and the closing 4x4
  B[2][0] = 1.0; B[2][1] = -1.0; B[2][2] = 0.0; B[2][3] = 0.0;
  B[3][0] = s.outer_u; B[3][1] = 0.0; B[3][2] = s.ell + 1.0; B[3][3] = 1.0;
  double* __restrict__ uf = up + n_int * 22;
  int dummy;
  pivot_sweep_upper<4, 4, 0, false>(B, det, uf + 0, &dummy);
  pivot_sweep_upper<4, 4, 1, false>(B, det, uf + 3, &dummy);
  pivot_sweep_upper<4, 4, 2, false>(B, det, uf + 5, &dummy);
  det_out[m] = det * B[3][3];
}

// Back-substitution.  `upper` is walked backwards exactly like the reference's reversed
// iterator; e[] is the eigenvector, length (n_int+1)*4, last component fixed to 1.
__global__ __launch_bounds__(64) void eigen_backsub_kernel(size_t n_int, size_t n, const double* __restrict__ upper,
                                                           const int* __restrict__ swaps, double* __restrict__ vec) {
  const size_t m = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (m >= n) return;
  const size_t ustride = n_int * 22 + 6;
  const size_t len = (n_int + 1) * 4;
  const double* __restrict__ d = upper + m * ustride;
  double* __restrict__ e = vec + m * len;
  size_t it = ustride;  // reverse cursor
  // w[0] is the most recently produced component, w[k] the one produced k steps earlier
  double w[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) w[k] = 0.0;
  w[0] = 1.0;
  e[len - 1] = 1.0;
  // closing 4x4: components len-2, len-3, len-4
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    double nv = 0.0;
#pragma unroll
    for (int j = 0; j < i; ++j) nv = nv - d[--it] * w[i - 1 - j];
    // e[len-j-1] for j = 0..i-1 are w[i-1], w[i-2], ..., w[0] (oldest first)
#pragma unroll
    for (int k = 6; k > 0; --k) w[k] = w[k - 1];
    w[0] = nv;
    e[len - i - 1] = nv;
  }
  // now w[0..3] = e[len-4], e[len-3], e[len-2], e[len-1]
  for (size_t k = 0; k < n_int; ++k) {
    // the 4 components of grid point (n_int - 1 - k); they depend on the last 4..7 produced
    // reference: e[len - i - (k+1)*4 - 1] = -sum_{j < i+4} data.next() * e[len - j - k*4 - 1]
    // e[len - j - k*4 - 1], j = 0..: the first component of the previous block is j = 3.
    double base[7];
    // base[j] = e[len - j - k*4 - 1]; before this block w holds the previous block newest-first:
    // w[0] = e[len - 3 - k*4 - 1] ... w[3] = e[len - 0 - k*4 - 1]
    base[0] = w[3]; base[1] = w[2]; base[2] = w[1]; base[3] = w[0];
    double nv0 = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) nv0 = nv0 - d[--it] * base[j];
    base[4] = nv0;
    double nv1 = 0.0;
#pragma unroll
    for (int j = 0; j < 5; ++j) nv1 = nv1 - d[--it] * base[j];
    base[5] = nv1;
    double nv2 = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) nv2 = nv2 - d[--it] * base[j];
    base[6] = nv2;
    double nv3 = 0.0;
#pragma unroll
    for (int j = 0; j < 7; ++j) nv3 = nv3 - d[--it] * base[j];
    const size_t o = len - (k + 1) * 4 - 1;
    e[o] = nv0; e[o - 1] = nv1; e[o - 2] = nv2; e[o - 3] = nv3;
    w[3] = nv0; w[2] = nv1; w[1] = nv2; w[0] = nv3;
  }
  // undo the column pivots of the first interval, last recorded first
  const int s1 = swaps[m * 2 + 1], s0 = swaps[m * 2 + 0];
  if (s1 != 1) { const double a = e[s1], b = e[1]; e[s1] = b; e[1] = a; }
  if (s0 != 0) { const double a = e[s0], b = e[0]; e[s0] = b; e[0] = a; }
}

hipError_t launch_eigen(const SystemDev& s, bool ell0, const double* freq, size_t n, const double* steps,
                        double* upper, int* swaps, double* det_out, double* vec, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 63) / 64);
  if (ell0) eigen_forward_kernel<true><<<blocks, 64, 0, stream>>>(s, freq, n, steps, upper, swaps, det_out);
  else eigen_forward_kernel<false><<<blocks, 64, 0, stream>>>(s, freq, n, steps, upper, swaps, det_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  eigen_backsub_kernel<<<blocks, 64, 0, stream>>>(s.n_int, n, upper, swaps, vec);
  return hipGetLastError();
}

}  // namespace storm
