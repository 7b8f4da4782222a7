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
//                       86-161,271-433).  One wave per mode.
//   eigen_backsub_kernel  back-substitution to the eigenvector (reference src/solver.rs:100-131).
#include "common_kernels.hpp"
#include "device_system.hpp"

namespace STORM_NS {
using namespace storm;

template <int K>
SDEV double quad_bcast(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, K * 0x55, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, K * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

constexpr int kChainBlock = 16;  // steps per register block; one block is prefetched while one is consumed

struct RowBlock {
  double2 lo[kChainBlock], hi[kChainBlock];
};

SDEV void chain_load(RowBlock& b, const double2* __restrict__ rows, size_t i0) {
#pragma unroll
  for (int u = 0; u < kChainBlock; ++u) {
    b.lo[u] = rows[(i0 + u) * 8];
    b.hi[u] = rows[(i0 + u) * 8 + 1];
  }
}

SDEV double chain_step(double acc, double2 lo, double2 hi) {
  const double a0 = quad_bcast<0>(acc), a1 = quad_bcast<1>(acc), a2 = quad_bcast<2>(acc), a3 = quad_bcast<3>(acc);
  double t = lo.x * a0;
  t = lo.y * a1 + t;
  t = hi.x * a2 + t;
  t = hi.y * a3 + t;
  return t;
}

// SCALED: the running product is renormalised by an exact power of two after every block of 16 steps (largest entry
// brought to [1, 2)) and the exponents are summed, so the discriminant comes out as mantissa * 2^exponent and cannot
// overflow where the reference's raw product (src/solver.rs:486-492, det *= pivot :504,531) does.  The closing
// elimination runs on the renormalised product; with acc = 2^K acc' the discriminant is 2^(2K) times the one of acc'
// (the two outer-boundary rows carry the factor: scale rows 2..5 by 2^-K, columns 4..7 by 2^K, the last two rows by
// 2^-K).  Mathematically the reference's value; the pivot order may differ from the raw run's, so it agrees with the
// raw discriminant to rounding, not bit for bit.  every_step: renormalise after every step instead (the "steps" are segment
// transfer matrices, each the product of a hundred step matrices and up to ~2^200 in magnitude: sixteen of them between
// two renormalisations could overflow); the scalings are exact, so the result does not depend on where they happen.
template <bool ELL0, bool SCALED>
__global__ __launch_bounds__(64) void chain_kernel(SystemDev s, const double* __restrict__ freq, size_t n,
                                                   const double* __restrict__ steps, double* __restrict__ det_out,
                                                   long long* __restrict__ exp_out, const SystemDev* __restrict__ table,
                                                   const int* __restrict__ sys_index, bool every_step) {
  __shared__ double sh[64];
  const int tid = threadIdx.x;
  const int sub = tid & 15;
  const int r = sub & 3;   // row of the running product owned by this lane
  const int c = sub >> 2;  // column (= quad index inside the 16-lane group)
  const size_t fi_raw = (size_t)blockIdx.x * 4 + (tid >> 4);
  const size_t fi = fi_raw < n ? fi_raw : n - 1;
  const double2* __restrict__ rows = reinterpret_cast<const double2*>(steps + fi * s.n_int * 16) + r * 2;
  double acc = (r == c) ? 1.0 : 0.0;
  long long scale_exp = 0;
  auto renormalise = [&]() {
    if constexpr (SCALED) {
      unsigned long long m = (unsigned long long)__double_as_longlong(__builtin_fabs(acc));
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const unsigned long long o = __shfl_xor(m, d, 64);
        m = o > m ? o : m;
      }
      const int biased = (int)((m >> 52) & 0x7ff);
      if (biased != 0 && biased != 0x7ff) {  // (zero/subnormal or non-finite: leave it)
        const int k = biased - 1023;
        acc = __builtin_ldexp(acc, -k);
        scale_exp += k;
      }
    }
  };
  const size_t n_int = s.n_int;
  const size_t n_blocks = n_int / kChainBlock;
  size_t i = 0;
  if (n_blocks > 0) {
    RowBlock cur, nxt;
    chain_load(nxt, rows, 0);
    for (size_t b = 0; b < n_blocks; ++b) {
      cur = nxt;
      if (b + 1 < n_blocks) chain_load(nxt, rows, (b + 1) * kChainBlock);
#pragma unroll
      for (int u = 0; u < kChainBlock; ++u) {
        acc = chain_step(acc, cur.lo[u], cur.hi[u]);
        if (SCALED && every_step) renormalise();
      }
      renormalise();
    }
    i = n_blocks * kChainBlock;
  }
  for (; i < n_int; ++i) {
    acc = chain_step(acc, rows[i * 8], rows[i * 8 + 1]);
    if (SCALED && every_step) renormalise();
  }
  renormalise();
  sh[tid] = acc;  // element (r, c) at group*16 + c*4 + r: column-major, as M4 wants it
  __syncthreads();
  if (sub == 0 && fi_raw < n) {
    M4 a;
#pragma unroll
    for (int e = 0; e < 16; ++e) a.v[e] = sh[(tid & 48) + e];
    // (several systems per launch: the boundary rows are those of the frequency's own system; `s` carries the common
    // number of steps, which for a segment-parallel evaluation is the number of segments, not the system's intervals)
    const double det = table ? close_determinant<ELL0>(table[sys_index[fi_raw]], freq[fi_raw], a) : close_determinant<ELL0>(s, freq[fi_raw], a);
    if constexpr (SCALED) {
      // mantissa in [0.5, 1) like frexp; zero, inf and NaN keep exponent 0
      const unsigned long long bits = (unsigned long long)__double_as_longlong(det);
      const int biased = (int)((bits >> 52) & 0x7ff);
      if (biased != 0 && biased != 0x7ff) {
        det_out[fi_raw] = __builtin_ldexp(det, -(biased - 1022));
        exp_out[fi_raw] = 2 * scale_exp + (biased - 1022);
      } else {
        det_out[fi_raw] = det;
        exp_out[fi_raw] = det == 0.0 || det != det || biased == 0x7ff ? 0 : 2 * scale_exp;  // (subnormal: as is)
        if (biased == 0 && det != 0.0) exp_out[fi_raw] = 2 * scale_exp;
      }
    } else {
      det_out[fi_raw] = det;
    }
  }
}

hipError_t launch_chain(const SystemDev& s, bool ell0, const double* freq, size_t n, const double* steps,
                        double* det_out, hipStream_t stream, const SystemDev* table, const int* sys_index) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 3) / 4);
  if (ell0) chain_kernel<true, false><<<blocks, 64, 0, stream>>>(s, freq, n, steps, det_out, nullptr, table, sys_index, false);
  else chain_kernel<false, false><<<blocks, 64, 0, stream>>>(s, freq, n, steps, det_out, nullptr, table, sys_index, false);
  return hipGetLastError();
}
hipError_t launch_chain_scaled(const SystemDev& s, bool ell0, const double* freq, size_t n, const double* steps,
                               double* mantissa_out, long long* exponent_out, hipStream_t stream, bool every_step) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 3) / 4);
  if (ell0) chain_kernel<true, true><<<blocks, 64, 0, stream>>>(s, freq, n, steps, mantissa_out, exponent_out, nullptr, nullptr, every_step);
  else chain_kernel<false, true><<<blocks, 64, 0, stream>>>(s, freq, n, steps, mantissa_out, exponent_out, nullptr, nullptr, every_step);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Bracketing on the device (reference src/bracket.rs): the sign-swap filter over the scan values with an ORDERED
// compaction of the brackets, and the per-bracket state machine of InverseQuadratic::optimize advanced one evaluation
// per round in device memory, so that a refinement round is "propose -> evaluate -> update" without the abscissae or the
// discriminants crossing PCIe.  Integer ULP logic on the raw bit patterns like the reference (f64::to_bits).  The host
// versions of the same functions live in storm_hip.hip; this file is compiled -ffp-contract=off in namespace storm,
// which is the one the host uses for these kernels in BOTH arithmetic modes (the proposals are the reference's).
// ---------------------------------------------------------------------------------------
SDEV bool is_finite(double x) { return (((unsigned long long)__double_as_longlong(x) >> 52) & 0x7ff) != 0x7ff; }
SDEV bool is_nan(double x) { return x != x; }
// f64::signum compared for inequality: by sign bit; NaN.signum() is NaN and NaN != anything.
SDEV bool signum_differs(double a, double b) {
  if (is_nan(a) || is_nan(b)) return true;
  return (__double_as_longlong(a) < 0) != (__double_as_longlong(b) < 0);
}

constexpr int kSwapBlock = 256;  // threads
constexpr int kSwapPer = 4;      // adjacent pairs per thread

// 0: no sign swap, 1: a bracket, 2: a sign swap with a non-finite end (dropped, raises the "invalid numbers" flag)
SDEV int pair_kind(const double* __restrict__ det, size_t i) {
  const double a = det[i], b = det[i + 1];
  if (!signum_differs(a, b)) return 0;
  return (is_finite(a) && is_finite(b)) ? 1 : 2;
}

__global__ __launch_bounds__(kSwapBlock) void swap_count_kernel(const double* __restrict__ det, size_t n,
                                                                unsigned* __restrict__ block_count, int* __restrict__ flags) {
  __shared__ unsigned sh[kSwapBlock / 64];
  const size_t first = ((size_t)blockIdx.x * kSwapBlock + threadIdx.x) * kSwapPer;
  unsigned cnt = 0;
  bool invalid = false;
#pragma unroll
  for (int j = 0; j < kSwapPer; ++j)
    if (first + j + 1 < n) {
      const int k = pair_kind(det, first + j);
      cnt += (k == 1);
      invalid = invalid || (k == 2);
    }
  if (invalid) atomicOr(flags, 2);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
#pragma unroll
    for (int w = 0; w < kSwapBlock / 64; ++w) t += sh[w];
    block_count[blockIdx.x] = t;
  }
}

// The exclusive sum of the block counts, in place, by ONE workgroup (a scan has n / 1024 blocks: 98 for 10^5 points); the
// total goes to block_count[n_blocks].  On the device so that count -> offsets -> emit is one stream of launches: round 2
// summed on the host, a D2H copy, a synchronisation and an H2D copy in the middle of every scan.
__global__ __launch_bounds__(kSwapBlock) void swap_offsets_kernel(unsigned* __restrict__ block_count, unsigned n_blocks) {
  __shared__ unsigned sh[kSwapBlock];
  __shared__ unsigned carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (unsigned base = 0; base < n_blocks; base += kSwapBlock) {
    const unsigned k = base + threadIdx.x;
    const unsigned cnt = k < n_blocks ? block_count[k] : 0;
    sh[threadIdx.x] = cnt;
    __syncthreads();
    unsigned incl = cnt;
    for (int d = 1; d < kSwapBlock; d <<= 1) {
      const unsigned add = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      incl += add;
      sh[threadIdx.x] = incl;
      __syncthreads();
    }
    const unsigned before = carry;
    if (k < n_blocks) block_count[k] = before + incl - cnt;
    __syncthreads();
    if (threadIdx.x == kSwapBlock - 1) carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) block_count[n_blocks] = carry;
}

// block_offset[b] = number of brackets before block b (swap_offsets_kernel).  Brackets from position `cap` on are not
// written (the caller learns the total from the offsets and reports the overflow).
__global__ __launch_bounds__(kSwapBlock) void swap_emit_kernel(const double* __restrict__ freq, const double* __restrict__ det,
                                                               size_t n, const unsigned* __restrict__ block_offset, size_t cap,
                                                               BracketDev* __restrict__ out, unsigned long long* __restrict__ index_out) {
  __shared__ unsigned sh[kSwapBlock];
  const size_t first = ((size_t)blockIdx.x * kSwapBlock + threadIdx.x) * kSwapPer;
  int kind[kSwapPer];
  unsigned cnt = 0;
#pragma unroll
  for (int j = 0; j < kSwapPer; ++j) {
    kind[j] = (first + j + 1 < n) ? pair_kind(det, first + j) : 0;
    cnt += (kind[j] == 1);
  }
  sh[threadIdx.x] = cnt;
  __syncthreads();
  // exclusive scan over the block's threads (Hillis-Steele, 8 steps)
  unsigned incl = cnt;
  for (int d = 1; d < kSwapBlock; d <<= 1) {
    const unsigned add = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    incl += add;
    sh[threadIdx.x] = incl;
    __syncthreads();
  }
  size_t at = (size_t)block_offset[blockIdx.x] + (incl - cnt);
#pragma unroll
  for (int j = 0; j < kSwapPer; ++j)
    if (kind[j] == 1 && at >= cap) ++at;
    else if (kind[j] == 1) {
      const size_t i = first + j;
      BracketDev b;
      b.lo_x = freq[i]; b.lo_f = det[i]; b.hi_x = freq[i + 1]; b.hi_f = det[i + 1];
      b.pv_x = 0.0; b.pv_f = 0.0; b.root = 0.0; b.pending = 0.0; b.evals = 0; b.has_prev = 0; b.done = 0;
      out[at] = b;
      index_out[at] = i;
      ++at;
    }
}

// evaluate_secant / evaluate_inverse_quadratic at y = 0, src/bracket.rs:73-81 (same association as the host versions)
SDEV double dev_secant(double ax, double af, double bx, double bf) {
  const double y = 0.0;
  return (y - bf) / (af - bf) * ax + (y - af) / (bf - af) * bx;
}
SDEV double dev_inverse_quadratic(double ax, double af, double bx, double bf, double cx, double cf) {
  const double y = 0.0;
  return (y - bf) * (y - cf) / (af - bf) / (af - cf) * ax + (y - af) * (y - cf) / (bf - af) / (bf - cf) * bx +
         (y - af) * (y - bf) / (cf - bf) / (cf - af) * cx;
}

// One round of InverseQuadratic::optimize (src/bracket.rs:172-221) for every bracket: consume the discriminant of the
// pending abscissa (f_in, null in the first round), then propose the next one.  x_out[k] is what the discriminant
// kernels evaluate next; a finished bracket re-submits its root (the value is ignored).  live_out counts the brackets
// that still want an evaluation.  flags: 4 = non-finite discriminant (the reference asserts), 8 = bracket precondition.
__global__ __launch_bounds__(64) void refine_step_kernel(BracketDev* __restrict__ br, unsigned n, const double* __restrict__ f_in,
                                                         double* __restrict__ x_out, unsigned long long ulp,
                                                         unsigned* __restrict__ live_out, int* __restrict__ flags) {
  const unsigned k = blockIdx.x * 64 + threadIdx.x;
  if (k >= n) return;
  BracketDev b = br[k];
  if (!b.done && f_in != nullptr) {
    const double nx = b.pending, nf = f_in[k];
    if (!is_finite(nx) || !is_finite(nf)) {
      atomicOr(flags, 4);
      b.done = 1;
    } else {
      b.evals += 1;
      if (!signum_differs(b.hi_f, nf)) { b.pv_x = b.hi_x; b.pv_f = b.hi_f; b.hi_x = nx; b.hi_f = nf; }
      else { b.pv_x = b.lo_x; b.pv_f = b.lo_f; b.lo_x = nx; b.lo_f = nf; }
      b.has_prev = 1;
    }
  }
  if (!b.done) {
    double x;
    if (!b.has_prev) {
      x = dev_secant(b.lo_x, b.lo_f, b.hi_x, b.hi_f);
    } else {
      x = dev_inverse_quadratic(b.lo_x, b.lo_f, b.hi_x, b.hi_f, b.pv_x, b.pv_f);
      if (x <= b.lo_x || x >= b.hi_x) x = dev_secant(b.lo_x, b.lo_f, b.hi_x, b.hi_f);
    }
    // ensure_maximal_bracket, src/bracket.rs:96-132
    const double upper = b.hi_x, lower = b.lo_x;
    if (!(upper > lower) || !is_finite(upper) || !is_finite(lower) || !(upper >= 0.0) || !(lower >= 0.0)) {
      atomicOr(flags, 8);
      b.done = 1;
    } else {
      const unsigned long long ub = (unsigned long long)__double_as_longlong(upper), lb = (unsigned long long)__double_as_longlong(lower);
      if (ub - lb <= ulp) {
        b.done = 1;
        b.root = x;
      } else {
        unsigned long long xb = (unsigned long long)__double_as_longlong(x);
        if ((ub - lb) <= 2 * ulp) xb = lb + (ub - lb) / 2;
        else if ((ub < xb) || (ub - xb) < ulp) xb = ub - ulp;
        else if ((lb > xb) || (xb - lb) < ulp) xb = lb + ulp;
        b.pending = __longlong_as_double((long long)xb);
      }
    }
  }
  x_out[k] = b.done ? b.lo_x : b.pending;
  if (!b.done) atomicAdd(live_out, 1u);
  br[k] = b;
}

// One thread per interval: reads of a plane row are coalesced across the threads, each thread writes its interval's row.
__global__ __launch_bounds__(256) void aos_copy_kernel(double* __restrict__ tab, size_t sys_elems, size_t stride, int rows, int pitch) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= stride) return;
  const double* __restrict__ soa = tab + (size_t)blockIdx.y * sys_elems;
  double* __restrict__ dst = tab + (size_t)blockIdx.y * sys_elems + (size_t)rows * stride + i * (size_t)pitch;
  for (int r = 0; r < rows; ++r) dst[r] = soa[(size_t)r * stride + i];
  for (int r = rows; r < pitch; ++r) dst[r] = 0.0;
}
hipError_t launch_aos_copy(double* tab, size_t n_sys, size_t sys_elems, size_t stride, int p, hipStream_t stream) {
  if (n_sys == 0 || stride == 0) return hipSuccess;
  for (size_t k0 = 0; k0 < n_sys; k0 += 65535) {  // grid.y limit
    const unsigned cnt = (unsigned)std::min<size_t>(65535, n_sys - k0);
    aos_copy_kernel<<<dim3((unsigned)((stride + 255) / 256), cnt, 1), 256, 0, stream>>>(tab + k0 * sys_elems, sys_elems, stride,
                                                                                         kNumFields * p, aos_pitch(p));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_swap_count(const double* det, size_t n, unsigned* block_count, int* flags, hipStream_t stream) {
  if (n < 2) return hipSuccess;
  swap_count_kernel<<<(unsigned)swap_blocks(n), kSwapBlock, 0, stream>>>(det, n, block_count, flags);
  return hipGetLastError();
}
hipError_t launch_swap_offsets(unsigned* block_count, size_t n_blocks, hipStream_t stream) {
  swap_offsets_kernel<<<1, kSwapBlock, 0, stream>>>(block_count, (unsigned)n_blocks);
  return hipGetLastError();
}
hipError_t launch_swap_emit(const double* freq, const double* det, size_t n, const unsigned* block_offset, size_t cap, BracketDev* out,
                            unsigned long long* index_out, hipStream_t stream) {
  if (n < 2) return hipSuccess;
  swap_emit_kernel<<<(unsigned)swap_blocks(n), kSwapBlock, 0, stream>>>(freq, det, n, block_offset, cap, out, index_out);
  return hipGetLastError();
}
hipError_t launch_refine_step(BracketDev* br, size_t n, const double* f_in, double* x_out, unsigned long long ulp,
                              unsigned* live_out, int* flags, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  refine_step_kernel<<<(unsigned)((n + 63) / 64), 64, 0, stream>>>(br, (unsigned)n, f_in, x_out, ulp, live_out, flags);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Eigenvector: forward elimination, one wave per mode.
// Lane j < 8 owns system column j of the 6x8 band (6 values in registers).  A lone wave issues one instruction
// (vector OR scalar) per ~4 cycles, so the kernel is shaped by instruction count, not by lanes:
//   * pivot search: every lane searches its own column with 4 instructions per row; the owner's answer (lane K)
//     is made wave-uniform with ONE v_readlane (instead of reading the whole column into scalar registers first);
//   * row exchange: the two rows are addressed by the uniform row index (indexed register access; no branches, no
//     scalar bookkeeping of a second copy of the pivot column);
//   * the packed factors go to LDS through per-lane offsets fixed at kernel start (lanes without a factor write to a
//     dump slot), so no execution-mask juggling sits between two pivots;
//   * the pivot and the multipliers' column are read with v_readlane AFTER the exchange and used straight from
//     scalar registers;
//   * the carry of rows 4,5 into the next interval is a DPP row shift (not a ds_bpermute round trip);
//   * step matrices arrive 8 intervals (1 KB) per LDS-DMA instruction, one block ahead; the 22 packed factors per
//     interval are parked in LDS and flushed 8 intervals at a time with coalesced stores, so no wait on the
//     vector-memory counter sits inside an interval.
// Arithmetic per entry is exactly the reference's (row_pivot!/column_pivot!/sweep!, src/solver.rs:163-269):
//   m = B[r][K] * (1/pivot);  B[r][j] -= prow[j] * m.
// ---------------------------------------------------------------------------------------
SDEV double lane_value(double x, int lane) {  // wave-uniform copy of lane `lane`'s value
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}

SDEV double row_shift_left4(double x) {  // lane i <- lane i + 4 (inside each row of 16 lanes; 0 past the row)
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xf, 0xf, true);  // row_shl:4
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// ROWS x COLS elimination step K on column-distributed data.  `b` = this lane's column.
// `up_slot` is where THIS lane puts prow*pinv: [j-K-1] of the packed upper row of this pivot for K < j < COLS, a dump
// slot for every other lane (so the store needs no execution mask).
template <int ROWS, int COLS, int K, bool SEARCH>
SDEV void wave_pivot(double (&b)[6], double& det, double* __restrict__ up_slot) {
  double prow = b[K];
  if constexpr (SEARCH) {
    int idx = K;
    double best = __builtin_fabs(b[K]);
#pragma unroll
    for (int r = K + 1; r < ROWS; ++r) {
      const double v = __builtin_fabs(b[r]);
      const bool gt = v > best;  // strict: the first maximum wins, NaN never does
      idx = gt ? r : idx;
      best = gt ? v : best;
    }
    const int max_idx = __builtin_amdgcn_readlane(idx, K);  // the owner of column K decides for the wave
    prow = b[max_idx];   // wave-uniform register index (s_set_gpr_idx): cheaper than selects on every row
    b[max_idx] = b[K];
    det = det * (max_idx != K ? -1.0 : 1.0);
  }
  const double pivot = lane_value(prow, K);
  det = det * pivot;
  const double pinv = 1.0 / pivot;
  *up_slot = prow * pinv;
#pragma unroll
  for (int r = K + 1; r < ROWS; ++r) {
    const double m = lane_value(b[r], K) * pinv;
    b[r] = b[r] - prow * m;
  }
}

// column_pivot! (src/solver.rs:192-230) for pivot K of the first interval: pick the largest entry
// of system row K among columns K..7, exchange the two columns in rows K..5.
template <int K>
SDEV int wave_column_swap(double (&b)[6], int j, double& det) {
  int max_idx = K;
  double max_val = __builtin_fabs(lane_value(b[K], K));
#pragma unroll
  for (int i = K + 1; i < 8; ++i) {
    const double v = __builtin_fabs(lane_value(b[K], i));
    if (v > max_val) { max_idx = i; max_val = v; }
  }
  max_idx = __builtin_amdgcn_readfirstlane(max_idx);
  if (max_idx != K) {
#pragma unroll
    for (int r = K; r < 6; ++r) {
      const double from_max = lane_value(b[r], max_idx), from_k = lane_value(b[r], K);
      b[r] = (j == K) ? from_max : ((j == max_idx) ? from_k : b[r]);
    }
    det = det * -1.0;
  }
  return max_idx;
}

constexpr int kEigBlock = 8;                    // intervals per staged block
constexpr int kEigStepTile = kEigBlock * 16;    // doubles: 8 step matrices, row-major
constexpr int kEigUpTile = kEigBlock * 22;      // doubles: 8 x 22 packed factors

typedef __attribute__((address_space(3))) void* eig_lds_ptr;
typedef __attribute__((address_space(1))) const void* eig_glb_ptr;

template <bool ELL0>
__global__ __launch_bounds__(64) void eigen_forward_kernel(SystemDev s0, const double* __restrict__ freq, size_t n,
                                                           const double* __restrict__ steps,
                                                           double* __restrict__ upper, int* __restrict__ swaps,
                                                           double* __restrict__ det_out,
                                                           const SystemDev* __restrict__ table,
                                                           const int* __restrict__ sys_index) {
  // one shared object (a second one beside an LDS-DMA target makes hipcc drain the DMA before every LDS read)
  __shared__ __attribute__((aligned(16))) double lds[2 * kEigStepTile + 2 * kEigUpTile + 64];
  double* const up_tiles = lds + 2 * kEigStepTile;
  double* const dump = lds + 2 * kEigStepTile + 2 * kEigUpTile;  // one slot per lane for factors nobody needs
  const int lane = threadIdx.x;
  const int j = lane;          // lanes 0..7 carry the band; the others idle along (their values are never used)
  const size_t m = blockIdx.x;  // one wave per mode
  const SystemDev s = pick_system(s0, table, sys_index, m);  // several systems per launch: see LaunchArgs in types.hpp
  const double f = freq[m];
  const size_t n_int = s.n_int;
  const size_t ustride = n_int * 22 + 6;
  double* __restrict__ up = upper + m * ustride;
  const char* __restrict__ st_bytes = reinterpret_cast<const char*>(steps + m * n_int * 16);
  const size_t st_last = n_int * 128 - 16;  // last 16-B piece of this mode's step matrices
  auto stage_steps = [&](size_t i0, double* tile) {  // 8 intervals = 1 KB = one DMA instruction
    size_t off = i0 * 128 + (size_t)lane * 16;
    off = off < st_last ? off : st_last;
    __builtin_amdgcn_global_load_lds((eig_glb_ptr)(st_bytes + off), (eig_lds_ptr)tile, 16, 0, 0);
  };
  double b[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) b[r] = 0.0;
  // inner boundary rows (reference src/system/adiabatic.rs:154-183)
  {
    const double i00 = inner_00<ELL0>(s, f);
    b[0] = (j == 0) ? i00 : ((j == 1 || j == 2) ? -s.ell : 0.0);
    b[1] = (j == 2) ? s.ell : ((j == 3) ? -1.0 : 0.0);
  }
  auto flush_factors = [&](size_t i0, const double* tile) {  // block [i0, i0+8): coalesced, 8 B per lane and instruction
    const size_t cnt = (n_int - i0 < (size_t)kEigBlock ? n_int - i0 : (size_t)kEigBlock) * 22;
#pragma unroll
    for (int qd = 0; qd < (kEigUpTile + 63) / 64; ++qd) {
      const size_t e = (size_t)qd * 64 + lane;
      if (e < cnt) up[i0 * 22 + e] = tile[e];
    }
  };
  // where this lane's factor of pivot K goes, relative to the interval's 22 packed values (or to the dump slots)
  int slot[4];
  {
    constexpr int kRowStart[4] = {0, 7, 13, 18};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      slot[k] = (j > k && j < 8) ? kRowStart[k] + j - k - 1 : (int)(dump - up_tiles) + lane;
  }
  double det = 1.0;
  int sw0 = 0, sw1 = 1;
  if (n_int > 0) stage_steps(0, lds);
  int buf = 0;
  for (size_t i0 = 0; i0 < n_int; i0 += kEigBlock) {
    // Everything the vector-memory counter still holds was issued a whole block ago (this block's step matrices,
    // the flush of the block before the previous one), so this wait does not stall.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (i0 + kEigBlock < n_int) stage_steps(i0 + kEigBlock, lds + (buf ^ 1) * kEigStepTile);
    if (i0 > 0) flush_factors(i0 - kEigBlock, up_tiles + (buf ^ 1) * kEigUpTile);  // the previous block's factors
    double* const up_tile = up_tiles + buf * kEigUpTile;
    // lane j < 4 reads column j of each step matrix; the values of interval u+1 are fetched while u is eliminated
    const double* __restrict__ tile = lds + buf * kEigStepTile + (j & 3);
    double nl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) nl[r] = tile[r * 4];
#pragma unroll
    for (int u = 0; u < kEigBlock; ++u) {
      const size_t i = i0 + u;
      double cl[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) cl[r] = nl[r];
      if (u + 1 < kEigBlock) {
#pragma unroll
        for (int r = 0; r < 4; ++r) nl[r] = tile[(u + 1) * 16 + r * 4];
      }
      if (i < n_int) {
        // rows 2..5 <- [L | -I]
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r + 2] = (j < 4) ? cl[r] : ((j - 4 == r) ? -1.0 : 0.0);
        double* __restrict__ upi = up_tile + u * 22;
        // dump slots are addressed relative to up_tiles: undo the per-buffer, per-interval offset for them
        auto slot_ptr = [&](int k) { return (j > k && j < 8) ? upi + slot[k] : up_tiles + slot[k]; };
        if (u == 0 && i0 == 0) {
          sw0 = wave_column_swap<0>(b, j, det);
          wave_pivot<6, 8, 0, false>(b, det, slot_ptr(0));
          sw1 = wave_column_swap<1>(b, j, det);
          wave_pivot<6, 8, 1, false>(b, det, slot_ptr(1));
        } else {
          wave_pivot<6, 8, 0, true>(b, det, slot_ptr(0));
          wave_pivot<6, 8, 1, true>(b, det, slot_ptr(1));
        }
        wave_pivot<6, 8, 2, true>(b, det, slot_ptr(2));
        wave_pivot<6, 8, 3, true>(b, det, slot_ptr(3));
        // the two untouched rows move up: B[i][j] = B[i+4][j+4], right half cleared
        const double t4 = row_shift_left4(b[4]), t5 = row_shift_left4(b[5]);
        b[0] = (j < 4) ? t4 : 0.0;
        b[1] = (j < 4) ? t5 : 0.0;
      }
    }
    buf ^= 1;
  }
  if (n_int > 0) {
    __syncthreads();
    flush_factors((n_int - 1) / kEigBlock * kEigBlock, up_tiles + (buf ^ 1) * kEigUpTile);
  }
  if (lane == 0) { swaps[m * 2 + 0] = sw0; swaps[m * 2 + 1] = sw1; }
  // outer boundary rows (reference src/system/adiabatic.rs:185-208) and the closing 4x4
  b[2] = (j == 0) ? 1.0 : ((j == 1) ? -1.0 : 0.0);
  b[3] = (j == 0) ? s.outer_u : ((j == 2) ? s.ell + 1.0 : ((j == 3) ? 1.0 : 0.0));
  double* __restrict__ uf = up + n_int * 22;
  wave_pivot<4, 4, 0, true>(b, det, (j > 0 && j < 4) ? uf + 0 + j - 1 : dump + lane);
  wave_pivot<4, 4, 1, true>(b, det, (j > 1 && j < 4) ? uf + 3 + j - 2 : dump + lane);
  wave_pivot<4, 4, 2, true>(b, det, (j > 2 && j < 4) ? uf + 5 + j - 3 : dump + lane);
  const double last = lane_value(b[3], 3);
  if (lane == 0) det_out[m] = det * last;
}

// Back-substitution.  `upper` is walked backwards exactly like the reference's reversed iterator
// (src/solver.rs:100-131); e[] is the eigenvector, length (n_int+1)*4, last component fixed to 1.
// One wave per mode, every lane runs the same short recurrence (44 multiply/subtract instructions per interval) on
// wave-uniform data.  The packed factors arrive kBsTile intervals (5.5 KB) at a time by LDS-DMA, one tile ahead; the
// 22 factors of interval k-1 are read from LDS into registers while interval k is being computed, so neither global
// nor LDS latency sits on the serial chain.  Finished components are parked in LDS (every lane writes the same
// value) and flushed one tile at a time with coalesced stores.
constexpr int kBsTile = 32;                    // intervals per tile
constexpr int kBsTileDoubles = kBsTile * 22;   // 704 doubles = 5632 B = 5.5 DMA instructions
constexpr int kBsTileStride = 6 * 128;         // doubles: room for the 6 whole DMA instructions of a tile

__global__ __launch_bounds__(64) void eigen_backsub_kernel(size_t n_int, size_t n, const double* __restrict__ upper,
                                                           const int* __restrict__ swaps, double* __restrict__ vec) {
  // one shared object: two factor tiles + the parking area for finished components
  __shared__ __attribute__((aligned(16))) double lds[2 * kBsTileStride + kBsTile * 4];
  double* const park = lds + 2 * kBsTileStride;
  const size_t m = blockIdx.x;
  const int lane = threadIdx.x;
  const size_t ustride = n_int * 22 + 6;
  const size_t len = (n_int + 1) * 4;
  const double* __restrict__ d = upper + m * ustride;
  double* __restrict__ e = vec + m * len;
  // closing 4x4 (reference src/solver.rs:107-113): packed rows {k=0: 3, k=1: 2, k=2: 1 entries}
  const double* __restrict__ df = d + n_int * 22;
  const double e3 = 1.0;
  const double e2 = 0.0 - df[5] * e3;
  const double e1 = (0.0 - df[4] * e3) - df[3] * e2;
  const double e0 = ((0.0 - df[2] * e3) - df[1] * e2) - df[0] * e1;
  if (lane == 0) { e[len - 1] = e3; e[len - 2] = e2; e[len - 3] = e1; e[len - 4] = e0; }
  double b0 = e3, b1 = e2, b2 = e1, b3 = e0;
  // tiles are taken from the END of the interval range: tile t covers intervals [hi - cnt, hi), hi = n_int - t*kBsTile
  const size_t n_tiles = (n_int + kBsTile - 1) / kBsTile;
  auto tile_bounds = [&](size_t t, size_t& lo, size_t& cnt) {
    const size_t hi = n_int - t * kBsTile;
    cnt = hi < (size_t)kBsTile ? hi : (size_t)kBsTile;
    lo = hi - cnt;
  };
  const char* __restrict__ dbytes = reinterpret_cast<const char*>(d);
  const size_t last_piece = n_int * 176 - 16;  // (n_int > 0 whenever this is used)
  auto stage_tile = [&](size_t t, double* dst) {  // interval lo + q of the tile lands at dst[q*22 ...]
    size_t lo, cnt;
    tile_bounds(t, lo, cnt);
    static_assert((kBsTileDoubles * 8 + 1023) / 1024 * 128 <= kBsTileStride, "tile stride");
#pragma unroll
    for (int qd = 0; qd < (kBsTileDoubles * 8 + 1023) / 1024; ++qd) {
      size_t off = lo * 176 + (size_t)qd * 1024 + (size_t)lane * 16;
      off = off < last_piece ? off : last_piece;
      __builtin_amdgcn_global_load_lds((eig_glb_ptr)(dbytes + off), (eig_lds_ptr)(dst + qd * 128), 16, 0, 0);
    }
  };
  struct Factors { double2 v[11]; };
  auto read_factors = [](const double* src) {
    Factors f;
#pragma unroll
    for (int q = 0; q < 11; ++q) f.v[q] = *reinterpret_cast<const double2*>(src + q * 2);
    return f;
  };
  if (n_tiles > 0) stage_tile(0, lds);
  for (size_t t = 0; t < n_tiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t (requested one tile ago) has landed
    __syncthreads();
    if (t + 1 < n_tiles) stage_tile(t + 1, lds + ((t + 1) & 1) * kBsTileStride);
    size_t lo, cnt;
    tile_bounds(t, lo, cnt);
    const double* __restrict__ cur_tile = lds + (t & 1) * kBsTileStride;
    // One interval.  Written so that the in-order issue of a lone wave is not held up by FP64 latency: first the 16
    // products that only need the previous point (independent of each other), then the four partial sums side by
    // side, then the short triangular tail that really is serial.  Same operations per component, same order of the
    // subtractions within each component as the reference's reversed iterator.
    auto interval = [&](const Factors& c, size_t k) {
      // entry 2q, 2q+1 of the interval's 22 = c.v[q].x, .y
      const double p00 = c.v[10].y * b0, p10 = c.v[8].y * b0, p20 = c.v[6].x * b0, p30 = c.v[3].x * b0;
      const double p01 = c.v[10].x * b1, p11 = c.v[8].x * b1, p21 = c.v[5].y * b1, p31 = c.v[2].y * b1;
      const double p02 = c.v[9].y * b2, p12 = c.v[7].y * b2, p22 = c.v[5].x * b2, p32 = c.v[2].x * b2;
      const double p03 = c.v[9].x * b3, p13 = c.v[7].x * b3, p23 = c.v[4].y * b3, p33 = c.v[1].y * b3;
      double v0 = 0.0 - p00, v1 = 0.0 - p10, v2 = 0.0 - p20, v3 = 0.0 - p30;
      v0 = v0 - p01; v1 = v1 - p11; v2 = v2 - p21; v3 = v3 - p31;
      v0 = v0 - p02; v1 = v1 - p12; v2 = v2 - p22; v3 = v3 - p32;
      v0 = v0 - p03; v1 = v1 - p13; v2 = v2 - p23; v3 = v3 - p33;
      const double q10 = c.v[6].y * v0, q20 = c.v[4].x * v0, q30 = c.v[1].x * v0;
      v1 = v1 - q10; v2 = v2 - q20; v3 = v3 - q30;
      const double q21 = c.v[3].y * v1, q31 = c.v[0].y * v1;
      v2 = v2 - q21; v3 = v3 - q31;
      v3 = v3 - c.v[0].x * v2;
      // components of grid point (lo + k): y1..y4 = v3, v2, v1, v0 (all lanes hold the same values)
      double2* __restrict__ dst = reinterpret_cast<double2*>(park + k * 4);
      dst[0] = make_double2(v3, v2);
      dst[1] = make_double2(v1, v0);
      b0 = v0; b1 = v1; b2 = v2; b3 = v3;
    };
    // two register sets take turns (no copies): while one interval is computed the next one's factors are in flight
    Factors fa = read_factors(cur_tile + (cnt - 1) * 22), fb;
    size_t k = cnt;
    while (k >= 2) {
      fb = read_factors(cur_tile + (k - 2) * 22);
      interval(fa, k - 1);
      if (k >= 3) fa = read_factors(cur_tile + (k - 3) * 22);
      interval(fb, k - 2);
      k -= 2;
    }
    if (k == 1) interval(fa, 0);
    __syncthreads();
#pragma unroll
    for (int qd = 0; qd < kBsTile * 4 / 64; ++qd) {
      const size_t at = (size_t)qd * 64 + lane;
      if (at < cnt * 4) e[lo * 4 + at] = park[at];
    }
  }
  // undo the column pivots of the first interval, last recorded first
  __syncthreads();
  if (lane == 0) {
    __threadfence_block();
    const int s1 = swaps[m * 2 + 1], s0 = swaps[m * 2 + 0];
    if (s1 != 1) { const double a = e[s1], b = e[1]; e[s1] = b; e[1] = a; }
    if (s0 != 0) { const double a = e[s0], b = e[0]; e[s0] = b; e[0] = a; }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Lane-per-mode variants of the two eigenvector kernels, for launches with thousands of modes (storm_sweep: every mode of
// every system of a rotation sweep).  The wave-per-mode kernels above spend a whole wave's issue slot on 8 useful lanes
// and a lone wave already saturates its SIMD (one FP64 VALU instruction per 4 cycles), so beyond ~1 wave per SIMD their
// time grows linearly with the number of modes.  Here every lane carries one mode completely in registers (the 6 x 8
// band, 48 doubles): no cross-lane traffic, ~700 instructions per interval for 64 modes instead of ~350 for one.  Same
// operations in the same order per mode (row_pivot!/column_pivot!/sweep!, src/solver.rs:163-269), so the same bits.
// ---------------------------------------------------------------------------------------------------------------------

// One pivot step of the band held by this lane: rows K..ROWS-1, columns K..COLS-1 (entries left of the pivot column are
// never read again).  fac[0 .. COLS-K-2] receives the normalised pivot row.
template <int ROWS, int COLS, int K, bool SEARCH>
SDEV void lane_pivot(double (&b)[6][8], double& det, double* __restrict__ fac) {
  double prow[COLS];
  if constexpr (SEARCH) {
    int idx = K;
    double best = __builtin_fabs(b[K][K]);
#pragma unroll
    for (int r = K + 1; r < ROWS; ++r) {
      const double v = __builtin_fabs(b[r][K]);
      const bool gt = v > best;  // strict: the first maximum wins, NaN never does
      idx = gt ? r : idx;
      best = gt ? v : best;
    }
    // the row exchange, by selects (the row index differs from lane to lane)
#pragma unroll
    for (int c = K; c < COLS; ++c) {
      double pv = b[K][c];
#pragma unroll
      for (int r = K + 1; r < ROWS; ++r) pv = (idx == r) ? b[r][c] : pv;
      prow[c] = pv;
    }
#pragma unroll
    for (int r = K + 1; r < ROWS; ++r)
#pragma unroll
      for (int c = K; c < COLS; ++c) b[r][c] = (idx == r) ? b[K][c] : b[r][c];
    det = det * (idx != K ? -1.0 : 1.0);
  } else {
#pragma unroll
    for (int c = K; c < COLS; ++c) prow[c] = b[K][c];
  }
  const double pivot = prow[K];
  det = det * pivot;
  const double pinv = 1.0 / pivot;
#pragma unroll
  for (int c = K + 1; c < COLS; ++c) fac[c - K - 1] = prow[c] * pinv;
#pragma unroll
  for (int r = K + 1; r < ROWS; ++r) {
    const double m = b[r][K] * pinv;
#pragma unroll
    for (int c = K + 1; c < COLS; ++c) b[r][c] = b[r][c] - prow[c] * m;
  }
}

// column_pivot! for pivot K of the first interval, lane-local.  Returns the column exchanged with K (K if none).
template <int K>
SDEV int lane_column_swap(double (&b)[6][8], double& det) {
  int idx = K;
  double best = __builtin_fabs(b[K][K]);
#pragma unroll
  for (int c = K + 1; c < 8; ++c) {
    const double v = __builtin_fabs(b[K][c]);
    const bool gt = v > best;
    idx = gt ? c : idx;
    best = gt ? v : best;
  }
#pragma unroll
  for (int r = K; r < 6; ++r) {
    const double from_k = b[r][K];
    double from_max = from_k;
#pragma unroll
    for (int c = K + 1; c < 8; ++c) from_max = (idx == c) ? b[r][c] : from_max;
    b[r][K] = from_max;
#pragma unroll
    for (int c = K + 1; c < 8; ++c) b[r][c] = (idx == c) ? from_k : b[r][c];
  }
  det = det * (idx != K ? -1.0 : 1.0);
  return idx;
}

// ELL_CLASS: 0 = every mode has ell != 0, 1 = every mode has ell == 0, 2 = per lane (a sweep's modes of all systems in
// one launch; only the inner boundary depends on it).
template <int ELL_CLASS>
__global__ __launch_bounds__(64) void eigen_forward_lanes_kernel(SystemDev s0, const double* __restrict__ freq, size_t n,
                                                                 const double* __restrict__ steps,
                                                                 double* __restrict__ upper, int* __restrict__ swaps,
                                                                 double* __restrict__ det_out,
                                                                 const SystemDev* __restrict__ table,
                                                                 const int* __restrict__ sys_index) {
  const size_t m_raw = (size_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = m_raw < n;
  const size_t m = live ? m_raw : n - 1;  // surplus lanes repeat the last mode and store nothing
  const size_t n_int = s0.n_int;          // shared by every system of the launch
  SystemDev s = s0;
  if (table) {  // only the boundary scalars differ from lane to lane
    const SystemDev* __restrict__ mine = table + sys_index[m];
    s.ell = mine->ell; s.inner_c1 = mine->inner_c1; s.inner_rotm = mine->inner_rotm; s.outer_u = mine->outer_u;
  }
  const double f = freq[m];
  // factors go out mode-fastest, [interval*22 + entry][mode] (row length n_pad = the grid's lane count): one coalesced
  // 512-B store per entry instead of 64 lines touched by every store; surplus lanes write their own (unused) columns
  const size_t n_pad = (size_t)gridDim.x * 64;
  double* __restrict__ up = upper + m_raw;
  const double2* __restrict__ st = reinterpret_cast<const double2*>(steps + m * n_int * 16);
  double b[6][8];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) b[r][c] = 0.0;
  // inner boundary rows (reference src/system/adiabatic.rs:154-183)
  if constexpr (ELL_CLASS == 2) {
    const double radial = inner_00<true>(s, f), other = inner_00<false>(s, f);  // (the unused one may be inf/NaN)
    b[0][0] = (s.ell == 0.0) ? radial : other;
  } else {
    b[0][0] = inner_00<ELL_CLASS == 1>(s, f);
  }
  b[0][1] = -s.ell; b[0][2] = -s.ell;
  b[1][2] = s.ell;  b[1][3] = -1.0;
  double det = 1.0;
  int sw0 = 0, sw1 = 1;
  double2 nxt[8];  // the step matrix of the next interval, row-major, fetched one interval (~3000 cycles) ahead
  if (n_int > 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) nxt[q] = st[q];
  }
  for (size_t i = 0; i < n_int; ++i) {
    // rows 2..5 <- [L | -I]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      b[r + 2][0] = nxt[r * 2].x; b[r + 2][1] = nxt[r * 2].y; b[r + 2][2] = nxt[r * 2 + 1].x; b[r + 2][3] = nxt[r * 2 + 1].y;
#pragma unroll
      for (int c = 0; c < 4; ++c) b[r + 2][c + 4] = (r == c) ? -1.0 : 0.0;
    }
    {
      const size_t ni = i + 1 < n_int ? i + 1 : i;
#pragma unroll
      for (int q = 0; q < 8; ++q) nxt[q] = st[ni * 8 + q];
    }
    double fac[22];
    if (i == 0) {
      sw0 = lane_column_swap<0>(b, det);
      lane_pivot<6, 8, 0, false>(b, det, fac + 0);
      sw1 = lane_column_swap<1>(b, det);
      lane_pivot<6, 8, 1, false>(b, det, fac + 7);
    } else {
      lane_pivot<6, 8, 0, true>(b, det, fac + 0);
      lane_pivot<6, 8, 1, true>(b, det, fac + 7);
    }
    lane_pivot<6, 8, 2, true>(b, det, fac + 13);
    lane_pivot<6, 8, 3, true>(b, det, fac + 18);
    {
      double* __restrict__ dst = up + i * 22 * n_pad;
#pragma unroll
      for (int q = 0; q < 22; ++q) dst[q * n_pad] = fac[q];
    }
    // the two untouched rows move up: B[i][j] = B[i+4][j+4], right half cleared
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      b[0][c] = b[4][c + 4]; b[1][c] = b[5][c + 4];
      b[0][c + 4] = 0.0; b[1][c + 4] = 0.0;
    }
  }
  // outer boundary rows (reference src/system/adiabatic.rs:185-208) and the closing 4x4
  b[2][0] = 1.0; b[2][1] = -1.0; b[2][2] = 0.0; b[2][3] = 0.0;
  b[3][0] = s.outer_u; b[3][1] = 0.0; b[3][2] = s.ell + 1.0; b[3][3] = 1.0;
  double fin[6];
  lane_pivot<4, 4, 0, true>(b, det, fin + 0);
  lane_pivot<4, 4, 1, true>(b, det, fin + 3);
  lane_pivot<4, 4, 2, true>(b, det, fin + 5);
  {
    double* __restrict__ dst = up + n_int * 22 * n_pad;
#pragma unroll
    for (int q = 0; q < 6; ++q) dst[q * n_pad] = fin[q];
  }
  if (live) {
    swaps[m * 2 + 0] = sw0; swaps[m * 2 + 1] = sw1;
    det_out[m] = det * b[3][3];
  }
}

// Back-substitution, one lane per mode: 22 factors per interval straight from global memory into registers, four
// intervals ahead (HBM latency is several intervals long), 44 multiply/subtract instructions per interval for 64 modes.
// The result layout is [mode][point][4], i.e. lane-strided by the whole eigenvector: written directly, every store
// touches 64 lines (measured: half the kernel's time).  So finished points are parked in LDS, 16 points x 4 components
// per mode, and flushed transposed: one coalesced 512-B store per mode and tile.
constexpr int kBlTilePoints = 16;
constexpr int kBlTileRows = kBlTilePoints * 4;  // 64: one row per (point, component), one column per mode
constexpr int kBlTileStride = 65;               // doubles per row (odd: the transposed read is conflict-free)

__global__ __launch_bounds__(64) void eigen_backsub_lanes_kernel(size_t n_int, size_t n, const double* __restrict__ upper,
                                                                 const int* __restrict__ swaps, double* __restrict__ vec) {
  __shared__ double tile[kBlTileRows * kBlTileStride];
  const int lane = threadIdx.x;
  const size_t m0 = (size_t)blockIdx.x * 64;
  const size_t m_raw = m0 + lane;
  const bool live = m_raw < n;
  const size_t m = live ? m_raw : n - 1;
  const size_t len = (n_int + 1) * 4;
  const size_t n_pad = (size_t)gridDim.x * 64;  // factors are mode-fastest, see eigen_forward_lanes_kernel
  const double* __restrict__ d = upper + m_raw;
  const int sw1 = swaps[m * 2 + 1], sw0 = swaps[m * 2 + 0];
  const size_t modes_here = n - m0 < 64 ? n - m0 : 64;
  // Point p = (y1..y4)(x_p) of this lane's mode goes to column `lane` of the tile; the tile holding point 0 is the
  // last one, and the column exchanges of the first interval (entries 0..7 = points 0 and 1) are undone in it.
  auto flush = [&](size_t p0) {
    const size_t top = n_int - p0 < (size_t)(kBlTilePoints - 1) ? n_int - p0 : (size_t)(kBlTilePoints - 1);  // last point held
    const int rows = (int)(top + 1) * 4;
    if (p0 == 0 && n_int > 0) {  // last recorded first
      double* col = tile + lane;
      if (sw1 != 1) { const double a = col[sw1 * kBlTileStride], bb = col[1 * kBlTileStride]; col[sw1 * kBlTileStride] = bb; col[1 * kBlTileStride] = a; }
      if (sw0 != 0) { const double a = col[sw0 * kBlTileStride], bb = col[0]; col[sw0 * kBlTileStride] = bb; col[0] = a; }
    }
    __syncthreads();
    for (size_t jm = 0; jm < modes_here; ++jm) {
      const double v = tile[lane * kBlTileStride + jm];
      if (lane < rows) vec[(m0 + jm) * len + p0 * 4 + lane] = v;
    }
    __syncthreads();
  };
  auto emit = [&](size_t p, double y1, double y2, double y3, double y4) {
    double* col = tile + (p % kBlTilePoints) * 4 * kBlTileStride + lane;
    col[0] = y1; col[kBlTileStride] = y2; col[2 * kBlTileStride] = y3; col[3 * kBlTileStride] = y4;
    if (p % kBlTilePoints == 0) flush(p);
  };
  // closing 4x4 (reference src/solver.rs:107-113)
  const double* __restrict__ df = d + n_int * 22 * n_pad;
  const double e3 = 1.0;
  const double e2 = 0.0 - df[5 * n_pad] * e3;
  const double e1 = (0.0 - df[4 * n_pad] * e3) - df[3 * n_pad] * e2;
  const double e0 = ((0.0 - df[2 * n_pad] * e3) - df[1 * n_pad] * e2) - df[0] * e1;
  emit(n_int, e0, e1, e2, e3);
  double b0 = e3, b1 = e2, b2 = e1, b3 = e0;
  struct Factors { double2 v[11]; };
  auto load = [&](long long k) {  // (only called with n_int > 0; indices below 0 re-read interval 0)
    Factors fc;
    const double* __restrict__ src = d + (size_t)(k < 0 ? 0 : k) * 22 * n_pad;
#pragma unroll
    for (int q = 0; q < 11; ++q) fc.v[q] = make_double2(src[(2 * q) * n_pad], src[(2 * q + 1) * n_pad]);
    return fc;
  };
  auto interval = [&](const Factors& c, size_t k) {  // see eigen_backsub_kernel
    const double p00 = c.v[10].y * b0, p10 = c.v[8].y * b0, p20 = c.v[6].x * b0, p30 = c.v[3].x * b0;
    const double p01 = c.v[10].x * b1, p11 = c.v[8].x * b1, p21 = c.v[5].y * b1, p31 = c.v[2].y * b1;
    const double p02 = c.v[9].y * b2, p12 = c.v[7].y * b2, p22 = c.v[5].x * b2, p32 = c.v[2].x * b2;
    const double p03 = c.v[9].x * b3, p13 = c.v[7].x * b3, p23 = c.v[4].y * b3, p33 = c.v[1].y * b3;
    double v0 = 0.0 - p00, v1 = 0.0 - p10, v2 = 0.0 - p20, v3 = 0.0 - p30;
    v0 = v0 - p01; v1 = v1 - p11; v2 = v2 - p21; v3 = v3 - p31;
    v0 = v0 - p02; v1 = v1 - p12; v2 = v2 - p22; v3 = v3 - p32;
    v0 = v0 - p03; v1 = v1 - p13; v2 = v2 - p23; v3 = v3 - p33;
    const double q10 = c.v[6].y * v0, q20 = c.v[4].x * v0, q30 = c.v[1].x * v0;
    v1 = v1 - q10; v2 = v2 - q20; v3 = v3 - q30;
    const double q21 = c.v[3].y * v1, q31 = c.v[0].y * v1;
    v2 = v2 - q21; v3 = v3 - q31;
    v3 = v3 - c.v[0].x * v2;
    emit(k, v3, v2, v1, v0);
    b0 = v0; b1 = v1; b2 = v2; b3 = v3;
  };
  if (n_int > 0) {
    long long k = (long long)n_int;
    Factors f0 = load(k - 1), f1 = load(k - 2), f2 = load(k - 3), f3 = load(k - 4);
    while (k >= 4) {
      // (scheduling barriers: left alone, the compiler sinks the refills towards their first use to save registers,
      // which is exactly the latency this ring is there to hide)
      interval(f0, (size_t)(k - 1)); f0 = load(k - 5); __builtin_amdgcn_sched_barrier(0);
      interval(f1, (size_t)(k - 2)); f1 = load(k - 6); __builtin_amdgcn_sched_barrier(0);
      interval(f2, (size_t)(k - 3)); f2 = load(k - 7); __builtin_amdgcn_sched_barrier(0);
      interval(f3, (size_t)(k - 4)); f3 = load(k - 8); __builtin_amdgcn_sched_barrier(0);
      k -= 4;
    }
    if (k >= 1) interval(f0, (size_t)(k - 1));
    if (k >= 2) interval(f1, (size_t)(k - 2));
    if (k >= 3) interval(f2, (size_t)(k - 3));
  }
}

// `layout`: 0 = by mode count, 1 = one wave per mode, 2 = one lane per mode.  `ell_class`: 0 = ell != 0 for every mode,
// 1 = ell == 0 for every mode, 2 = mixed (lane-per-mode kernels only: their one ell-dependent expression is selected
// per lane, so the modes of every system of a sweep go out in ONE launch).
bool eigen_uses_lanes(size_t n, int layout) {
  // wave per mode: ~4.6 + 1.1 ms per 6199 intervals while every wave has a SIMD to itself (1024 modes), then linear in
  // n (13.1 ms at 2061 modes); lane per mode: 9.6 + 2.5 ms for any count up to 64 per SIMD: they cross near 1900
  return layout == 2 || (layout == 0 && n >= 1920);
}
hipError_t launch_eigen(const SystemDev& s, int ell_class, const double* freq, size_t n, const double* steps,
                        double* upper, int* swaps, double* det_out, double* vec, hipStream_t stream,
                        const SystemDev* table, const int* sys_index, int layout) {
  if (n == 0) return hipSuccess;
  const bool lanes = eigen_uses_lanes(n, layout);  // (the two layouts keep `upper` differently: both kernels follow it)
  if (ell_class == 2 && !lanes) return hipErrorInvalidValue;
  const unsigned wblocks = (unsigned)n, lblocks = (unsigned)((n + 63) / 64);
  if (lanes) {
    if (ell_class == 2) eigen_forward_lanes_kernel<2><<<lblocks, 64, 0, stream>>>(s, freq, n, steps, upper, swaps, det_out, table, sys_index);
    else if (ell_class == 1) eigen_forward_lanes_kernel<1><<<lblocks, 64, 0, stream>>>(s, freq, n, steps, upper, swaps, det_out, table, sys_index);
    else eigen_forward_lanes_kernel<0><<<lblocks, 64, 0, stream>>>(s, freq, n, steps, upper, swaps, det_out, table, sys_index);
  } else {
    if (ell_class == 1) eigen_forward_kernel<true><<<wblocks, 64, 0, stream>>>(s, freq, n, steps, upper, swaps, det_out, table, sys_index);
    else eigen_forward_kernel<false><<<wblocks, 64, 0, stream>>>(s, freq, n, steps, upper, swaps, det_out, table, sys_index);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (lanes) eigen_backsub_lanes_kernel<<<lblocks, 64, 0, stream>>>(s.n_int, n, upper, swaps, vec);
  else eigen_backsub_kernel<<<wblocks, 64, 0, stream>>>(s.n_int, n, upper, swaps, vec);
  return hipGetLastError();
}

}  // namespace STORM_NS
