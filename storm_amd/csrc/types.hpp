// types.hpp -- plain types shared by the host code and by both arithmetic builds of the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// The device code is compiled twice from the same sources:
//   namespace storm       -ffp-contract=off   separate mul and add: bit-identical to the reference
//   namespace storm_fast  -ffp-contract=fast  v_fma_f64 allowed: ~2x arithmetic ceiling, results agree
//                                              with the reference to rounding (tolerances in DESIGN.md)
#ifndef STORM_NS
#define STORM_NS storm
#endif
// STORM_FAST_BUILD selects the FMA build's kernels and launch choices; STORM_FAST_MATH its re-formulated arithmetic
// (mat4.hpp, steppers.hpp, device_system.hpp).  The two only differ in the variant libraries of the attribution
// experiment (tests/attribution_config3.py), which switch the re-formulations off -- all of them with
// -DSTORM_FAST_PARITY_FORMULAS (contraction only), or one at a time with -DSTORM_FAST_IEEE_DIV, -DSTORM_FAST_NO_LU,
// -DSTORM_FAST_REF_INVERSE, -DSTORM_FAST_NO_FOLD.
#if defined(STORM_FAST_BUILD) && !defined(STORM_FAST_PARITY_FORMULAS)
#define STORM_FAST_MATH 1
#endif

namespace storm {

enum Scheme : int {
  kColloc2 = 0, kColloc4 = 1, kColloc6 = 2, kColloc8 = 3,
  kMagnus2 = 4, kMagnus4 = 5, kMagnus6 = 6, kMagnus8 = 7,
  kNumSchemes = 8
};
__host__ __device__ constexpr int scheme_points(int s) { return (s & 3) + 1; }
__host__ __device__ constexpr bool scheme_is_magnus(int s) { return s >= kMagnus2; }

enum Field : int {
  F_M00 = 0, F_M01, F_M10, F_M11, F_M22, F_M30, F_M31, F_M32 /* = dx*lambda */, F_M33,
  F_D /* dx = delta/x */, F_DC1 /* dx*c1 */, F_DL2 /* (dx*lambda)^2 */, F_ROTM /* m*rot */,
  kNumFields
};

// A single solver (not the systems of a sweep) keeps a second copy of the planes behind them, interval-major: the
// kNumFields * p values of interval i in consecutive doubles at tab + kNumFields * plane + i * aos_pitch(p), row order
// field * p + j.  The column-owner kernels read it: one address per lane and interval, immediate offsets for the rest.
__host__ __device__ constexpr int aos_pitch(int p) { return (kNumFields * p + 1) / 2 * 2; }  // doubles; rows 16-B aligned

struct SystemDev {
  const double* tab;   // kNumFields planes
  size_t plane;        // doubles per plane (= p * stride)
  size_t stride;       // doubles per (field, point) row: n_int rounded up to 16, plus 16 (rows are 128-B
                       // aligned and may be read one 16-interval round past the end; the padding repeats
                       // the last interval)
  size_t n_int;        // grid intervals
  double ell, lambda;  // lambda = ell*(ell+1)
  double inner_c1, inner_rotm, outer_u;
  double rotm_uniform;  // value of every m*rot entry when they are all bitwise equal
};

// The system of a wave when several systems share a launch: table[sys_index[at]] with a wave-uniform `at`, read through
// the constant address space so that the 80 bytes arrive by scalar loads and stay in SGPRs like the kernel argument
// `s0` does.  (A plain `table ? table[...] : s0` went through vector loads and selects and kept the struct in VGPRs:
// +28 to +78 registers per kernel, one wave per SIMD less for colloc2 and colloc6.)
__device__ __forceinline__ SystemDev pick_system(const SystemDev& s0, const SystemDev* table, const int* sys_index, size_t at) {
  if (table == nullptr) return s0;
  typedef const int __attribute__((address_space(4))) * c_int_ptr;
  typedef const unsigned long long __attribute__((address_space(4))) * c_u64_ptr;
  const int si = ((c_int_ptr)sys_index)[at];
  const c_u64_ptr src = (c_u64_ptr)(table + si);
  unsigned long long w[sizeof(SystemDev) / 8];
#pragma unroll
  for (unsigned k = 0; k < sizeof(SystemDev) / 8; ++k) w[k] = src[k];
  SystemDev s;
  __builtin_memcpy(&s, w, sizeof(SystemDev));
  return s;
}

struct LaunchArgs {
  SystemDev sys;
  const double* freq;  // device
  size_t n;
  double* out;         // det_out (fused kernels) or steps (build), device
  int* flags;          // device
  bool ell0, uniform;
  int hex_lanes;       // det_hex only: lanes per frequency, 16 or 32; 0 = pick by wave-slot packing
  int hex_rounds_per_launch = 1;  // det_hex only: a batch goes out in pieces of this many wave rounds; 0 = one launch
  hipStream_t stream;
  // Several systems in one launch (storm_sweep): frequency i belongs to system
  // table[sys_index[i]] (device arrays).  All systems of a launch share n_int, the scheme and the (ell0, uniform)
  // class; det_hex additionally needs every aligned group of 4 frequencies to belong to one system (the host pads).
  // table == nullptr: the one system `sys`.
  const SystemDev* table = nullptr;
  const int* sys_index = nullptr;
  // Segment-parallel kernels (seg_product, seg_fill; FMA build): n_seg segments (a multiple of 4) of seg_len intervals (a
  // multiple of 16) per frequency; `out` = the segment transfer matrices [n][n_seg][16]; growth (may be null) =
  // max |T_s| per frequency as raw bits; nodes / vec: seg_fill's boundary values in, eigenvectors out.
  int n_seg = 0, seg_len = 0;
  unsigned long long* growth = nullptr;
  const double* nodes = nullptr;
  double* vec = nullptr;
};
typedef hipError_t (*launch_fn)(const LaunchArgs&);

}  // namespace storm
