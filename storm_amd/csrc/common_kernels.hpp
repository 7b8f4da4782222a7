#pragma once
#include "types.hpp"

namespace storm {
// State of InverseQuadratic::optimize for one bracket (reference src/bracket.rs:134-146, BracketResult :6-15), as the
// device keeps it between refinement rounds.
struct BracketDev {
  double lo_x, lo_f, hi_x, hi_f;   // lower / upper (Point, src/bracket.rs:63-68)
  double pv_x, pv_f;               // previous (valid if has_prev)
  double root, pending;            // result once done; the abscissa whose discriminant is being evaluated
  unsigned long long evals;
  int has_prev, done;
};
// pairs (i, i+1), i < n - 1, in blocks of 256 threads x 4 pairs
inline size_t swap_blocks(size_t n) { return n < 2 ? 0 : (n - 1 + 1023) / 1024; }
}  // namespace storm

// Scheme-independent launchers, one set per arithmetic build (see types.hpp).
// launch_aos_copy fills the interval-major copy behind the planes of n_sys systems (types.hpp, aos_pitch): system k's
// planes start at tab + k * sys_elems, its copy kNumFields * p * stride doubles further.  HBM to HBM, so the upload
// carries the planes only.
#define STORM_DECL_COMMON(ns)                                                                                     \
  namespace ns {                                                                                                  \
  hipError_t launch_chain(const storm::SystemDev& s, bool ell0, const double* freq, size_t n, const double* steps, \
                          double* det_out, hipStream_t stream, const storm::SystemDev* table = nullptr,           \
                          const int* sys_index = nullptr);                                                        \
  hipError_t launch_chain_scaled(const storm::SystemDev& s, bool ell0, const double* freq, size_t n,               \
                                 const double* steps, double* mantissa_out, long long* exponent_out, hipStream_t stream, \
                                 bool every_step = false);                                                        \
  hipError_t launch_aos_copy(double* tab, size_t n_sys, size_t sys_elems, size_t stride, int p, hipStream_t stream); \
  hipError_t launch_swap_count(const double* det, size_t n, unsigned* block_count, int* flags, hipStream_t stream); \
  hipError_t launch_swap_offsets(unsigned* block_count, size_t n_blocks, hipStream_t stream);                      \
  hipError_t launch_swap_emit(const double* freq, const double* det, size_t n, const unsigned* block_offset,       \
                              size_t cap, storm::BracketDev* out, unsigned long long* index_out, hipStream_t stream); \
  hipError_t launch_refine_step(storm::BracketDev* br, size_t n, const double* f_in, double* x_out,                \
                                unsigned long long ulp, unsigned* live_out, int* flags, hipStream_t stream);       \
  bool eigen_uses_lanes(size_t n, int layout);                                                                    \
  hipError_t launch_eigen(const storm::SystemDev& s, int ell_class, const double* freq, size_t n, const double* steps, \
                          double* upper, int* swaps, double* det_out, double* vec, hipStream_t stream,            \
                          const storm::SystemDev* table, const int* sys_index, int layout);                       \
  }
STORM_DECL_COMMON(storm)
STORM_DECL_COMMON(storm_fast)
