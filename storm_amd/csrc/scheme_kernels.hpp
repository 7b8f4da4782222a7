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

namespace STORM_NS {
using namespace storm;

template <int SCHEME, bool ELL0, bool UNIFORM, class Coef>
SDEV M4 interval_step_from(const Coef& c, double lambda, double f, const FreqTerms& ut, bool& ok) {
  constexpr int P = scheme_points(SCHEME);
  M4 a[P];
#pragma unroll
  for (int j = 0; j < P; ++j) a[j] = point_matrix<ELL0, UNIFORM>(c, j, lambda, f, ut);
  return step_matrix<SCHEME>(a, ok);
}

template <int SCHEME, bool ELL0, bool UNIFORM>
SDEV M4 interval_step(const SystemDev& s, size_t i, double f, const FreqTerms& ut, bool& ok) {
  const GlobalCoef c{s.tab + i, s.plane, s.stride};
  return interval_step_from<SCHEME, ELL0, UNIFORM>(c, s.lambda, f, ut, ok);
}

// The same from the interval-major copy of the coefficients (types.hpp, aos_pitch): one address per lane, 16-byte loads
// at immediate offsets.  Same values, same bits.
template <int SCHEME, bool ELL0, bool UNIFORM>
SDEV M4 interval_step_aos(const SystemDev& s, size_t i, double f, const FreqTerms& ut, bool& ok) {
  constexpr int kP = scheme_points(SCHEME);
  const AosCoef<kP> c{static_cast<const double*>(
      __builtin_assume_aligned(s.tab + kNumFields * s.plane + i * (size_t)aos_pitch(kP), 16))};
  return interval_step_from<SCHEME, ELL0, UNIFORM>(c, s.lambda, f, ut, ok);
}

// Where the column-owner body takes its coefficients from the interval-major copy (AosCoef: 23 address instructions and
// half the load instructions less per interval for Colloc4) instead of the planes.  Same values, same bits.  Measured on
// one box, both trees alternating, 10^5 frequencies on tams (ms, planes -> copy): parity Colloc2 10.49 -> 10.28, Colloc4
// 23.56 -> 22.92, Magnus2 24.47 -> 24.72, Magnus6 40.7 -> 40.6, Magnus8 73.1 -> 83.9 (!); FMA build Colloc2 5.58 -> 5.42,
// Colloc4 10.71 -> 10.27, Magnus2 14.95 -> 14.73, Magnus6 25.57 -> 24.95, Magnus8 49.4 -> 45.8.
__host__ __device__ constexpr bool col_reads_aos(int scheme) {
#ifdef STORM_FAST_BUILD
  return true;
#else
  return scheme == kColloc2 || scheme == kColloc4;
#endif
}

// The 16/32-lane body of det_hex (without LDS prefetch; single solvers only -- for a sweep the second copy of every
// system's planes costs more in allocation and upload than the kernels gain): same-box A/B at 12 288 and 49 152
// frequencies, FMA build: Colloc4 1.65 -> 1.54 ms and 6.05 -> 5.73, Colloc2 -3 %, Magnus6 -3 / -5 %, Magnus8 -7 %, Magnus2
// +-0; parity build: Colloc2 -4 %, Colloc4 +-0, Magnus2 +5 %, Magnus6 +2 %, Magnus8 +9 %.
__host__ __device__ constexpr bool hex_reads_aos(int scheme) {
#ifdef STORM_FAST_BUILD
  return true;
#else
  return scheme == kColloc2;
#endif
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

// LPF (16 or 32) lanes per frequency ("hex" kernel, 64/LPF frequencies per wave).  Round t: lane s of a
// group builds the step matrix of interval LPF*t+s (phase A), parks it row-major in LDS, then the group
// multiplies the LPF matrices into the running product in interval order (phase B): lane (r, c) owns
// element (r, c) of the product, reads row r of each step matrix from LDS and gets column c of the
// product from its quad by DPP (with LPF = 32 the two 16-lane halves of a group carry the same product).
// Element by element this is the reference's gemm (same order as the ordered-product kernel in
// common_kernels.hip), so results stay bit-identical while one launch exposes n*LPF/64 waves: the host
// picks the LPF whose wave count packs the chip's wave slots best (see launch_hex in scheme_tu.hip).
constexpr int kHexLaneStride = 18;  // doubles; 16 + 2 of padding (LDS banks)
__host__ __device__ constexpr int hex_group_stride(int lpf) { return lpf * kHexLaneStride + 4; }  // doubles
// (Measured with the MFMA phase B: a group stride of 128 B past a multiple of 512 B, which gives the four blocks of an
// MFMA four different bank windows for their operand reads, changes nothing -- +-0.5 %, Magnus8 +5 % at 12 288
// frequencies: the LDS is not what these kernels wait for.)

SDEV double hex_chain_step(double acc, double2 lo, double2 hi) {
  const double a0 = quad_bcast_d<0>(acc), a1 = quad_bcast_d<1>(acc), a2 = quad_bcast_d<2>(acc), a3 = quad_bcast_d<3>(acc);
  double t = lo.x * a0;
  t = lo.y * a1 + t;
  t = hi.x * a2 + t;
  t = hi.y * a3 + t;
  return t;
}

// FMA build, 16 lanes per frequency: phase B as ONE v_mfma_f64_4x4x4 per interval and wave (four independent 4x4x4
// products per instruction = the wave's four frequencies) instead of 4 multiply-adds + 8 DPP moves + two 16-byte LDS
// reads per lane.  Operand layout measured with tools/mfma64_microbench.hip: block b = lanes {4b + q + 16 r};
// A[i][k] in lane 4b + i + 16k, B[k][j] in lane 4b + j + 16k, D[i][j] in lane 4b + j + 16i -- D's layout is B's, so the
// running product P <- S P chains from D into B with no shuffle (12 ns per dependent step; the DPP/FMA step: 56 ns).
// FP64 MFMA runs through the same FP64 pipe as v_fma_f64 on gfx950 (the microbench: VALU + MFMA together never pass
// 78 TFLOP/s), so this buys issue slots and LDS traffic (8 B instead of 32 B per lane and interval), not flops.  The
// sum over k is fused in the hardware's order: not offered in the parity build.
// Same-box A/B against the multiply-add phase B (tams, FMA build, 12 288 / 10^5 frequencies, 16 lanes forced): Colloc4
// -9 / -8 %, Colloc8 -5 / -6 %, Magnus2 -8 / -5 %, Magnus8 0 / -4 %; Colloc2 -1 / +4 %, Colloc6 +1 / +1 %, Magnus6 0 / +1 %:
// those three keep the multiply-adds in det_hex (the column-owner body: col_uses_mfma below).
__host__ __device__ constexpr bool hex_uses_mfma(int scheme, int lpf) {
#ifdef STORM_FAST_BUILD
  return (lpf == 16 || lpf == 32) && scheme != kColloc2 && scheme != kColloc6 && scheme != kMagnus6;
#else
  return false;
#endif
}

// Magnus 2/4/6 need just over 256 registers when left alone; capping them at 256 buys a second wave
// per SIMD (better issue utilisation, phase-B latency hidden behind the other wave's phase A).
__host__ __device__ constexpr int hex_min_waves(int scheme) {
#ifdef STORM_FAST_BUILD
  if (scheme == kColloc4) return 3;  // FMA build: 172 registers left alone, 168 buy the third wave per SIMD
#endif
  return scheme == kMagnus2 ? 2 : 1;
}

// Coefficient staging for det_hex: the 13*P plane rows of the NEXT round (LPF intervals per row) are
// copied global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, no VGPRs) while
// the current round computes, so a wave that is alone on its SIMD does not wait for L2 at the top of a
// round. Measured A/B on the box (LPF 16, n = 10^4, tams): Colloc8 +5.5 %, Colloc6 +2 %, Colloc4 -1.6 %,
// Magnus4 -10 %, Magnus6 0 %, Magnus8 -3 %  ->  enabled where it pays (and where two tiles per wave
// leave the LDS room for the kernel's occupancy: Colloc6 runs two waves per SIMD).
__host__ __device__ constexpr bool hex_prefetch(int scheme, int lpf) {
  return scheme == kColloc8 || (scheme == kColloc6 && lpf == 16);
}
__host__ __device__ constexpr int hex_rows_per_dma(int lpf) { return 128 / lpf; }  // 1 KB / (lpf * 8 B)
__host__ __device__ constexpr int hex_coef_rows(int scheme, int lpf) {
  return (kNumFields * scheme_points(scheme) + hex_rows_per_dma(lpf) - 1) / hex_rows_per_dma(lpf) * hex_rows_per_dma(lpf);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* glb_ptr_t;

template <int SCHEME, int LPF>
SDEV void hex_stage_coefficients(const SystemDev& s, size_t i0, double* tile, int lane) {
  constexpr int kRows = kNumFields * scheme_points(SCHEME);
  constexpr int kRpd = hex_rows_per_dma(LPF);
  constexpr int kPieces = LPF / 2;  // 16-byte pieces per row
#pragma unroll
  for (int q = 0; q < hex_coef_rows(SCHEME, LPF) / kRpd; ++q) {
    int row = q * kRpd + lane / kPieces;
    row = row < kRows ? row : kRows - 1;  // surplus lanes re-read the last row into the tile's slack rows
    const double* src = s.tab + (size_t)row * s.stride + i0 + (lane % kPieces) * 2;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(tile + q * 128), 16, 0, 0);
  }
}

// doubles of LDS the det_hex body needs
template <int SCHEME, int LPF>
__host__ __device__ constexpr int hex_lds_doubles() {
  constexpr int kHandOff = ((64 / LPF) * hex_group_stride(LPF) + 1) / 2 * 2;
  constexpr int kTile = hex_prefetch(SCHEME, LPF) ? hex_coef_rows(SCHEME, LPF) * LPF : 2;
  return kHandOff + 2 * kTile;
}

// The work of workgroup `block` (one wave) of det_hex; `lds`: hex_lds_doubles<SCHEME, LPF>() doubles, 16-B aligned.
// AOS: the system is a single solver's, which has the interval-major copy of the planes (a sweep's systems have not).
template <int SCHEME, bool ELL0, bool UNIFORM, int LPF, bool AOS>
SDEV void det_hex_body(const SystemDev& s0, const double* __restrict__ freq, size_t n, double* __restrict__ det_out,
                       int* __restrict__ flags, const SystemDev* __restrict__ table, const int* __restrict__ sys_index,
                       size_t block, double* __restrict__ lds) {
  static_assert(LPF == 16 || LPF == 32 || LPF == 64, "lanes per frequency");
  constexpr int kFreqs = 64 / LPF;  // frequencies per wave
  constexpr int kGroup = hex_group_stride(LPF);
  constexpr bool kPrefetch = hex_prefetch(SCHEME, LPF);
  constexpr int kTile = kPrefetch ? hex_coef_rows(SCHEME, LPF) * LPF : 2;  // doubles per coefficient tile
  // ONE shared array for the step-matrix hand-off and the two coefficient tiles: with a second __shared__ object
  // beside the LDS-DMA target hipcc waits vmcnt(0) before the first ds_read after every DMA issue, which drains the
  // prefetch at once (seen in the .s; cdna_hip_programming.md, "three .s-level traps").
  constexpr int kHandOff = (kFreqs * kGroup + 1) / 2 * 2;  // doubles, keeps the tiles 16-B aligned
  double* const coef = lds + kHandOff;
  const int tid = threadIdx.x;
  constexpr bool kMfma = hex_uses_mfma(SCHEME, LPF);
  // (MFMA phase B: MFMA block b = lanes {4b + q + 16 r}.  16 lanes per frequency: block b is frequency b.  32 lanes: blocks
  // 2g and 2g + 1 both carry the product of frequency g, as the two 16-lane halves of a group do in the multiply-add form.)
  static_assert(!kMfma || LPF == 16 || LPF == 32, "MFMA phase B: 16 or 32 lanes per frequency");
  const int g = kMfma ? (LPF == 16 ? (tid >> 2) & 3 : (tid >> 3) & 1) : tid / LPF;  // frequency within the wave
  const int sub = kMfma ? (tid & 3) + 4 * (tid >> 4) + (LPF == 32 ? 4 * (tid & 4) : 0) : tid % LPF;  // interval within the round (phase A)
  const int e = kMfma ? (tid & 3) * 4 + (tid >> 4) : tid & 15;      // element c*4+r of the product (phase B)
  const int r = e & 3;
  const int c = e >> 2;
  const size_t t = block * kFreqs + g;
  // the wave's system: its own when several systems share the launch (wave-uniform, read through the scalar unit)
  const SystemDev s = pick_system(s0, table, sys_index, block * kFreqs);
  const double f = freq[t < n ? t : n - 1];
  const FreqTerms ut = freq_terms<ELL0>(f, s.rotm_uniform, s.lambda);
  bool ok = true;
  double acc = (r == c) ? 1.0 : 0.0;
  const size_t n_int = s.n_int;
  double* __restrict__ mine = lds + g * kGroup + sub * kHexLaneStride;
  // (MFMA: this lane's A operand is S(i, k) with i = tid % 4, k = tid / 16, row-major in the hand-off)
  const double* __restrict__ rows = kMfma ? lds + g * kGroup + (tid & 3) * 4 + (tid >> 4) : lds + g * kGroup + r * 4;
  if constexpr (kPrefetch) {
    if (n_int > 0) hex_stage_coefficients<SCHEME, LPF>(s, 0, coef, tid);
  }
  int buf = 0;
  for (size_t i0 = 0; i0 < n_int; i0 += LPF) {
    M4 l;
    if constexpr (kPrefetch) {
      // tile `buf` was requested one round ago (or in the prologue): make sure it has landed
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (i0 + LPF < n_int) hex_stage_coefficients<SCHEME, LPF>(s, i0 + LPF, coef + (buf ^ 1) * kTile, tid);
      const LdsCoef<scheme_points(SCHEME), LPF> cf{coef + buf * kTile + sub};
      l = interval_step_from<SCHEME, ELL0, UNIFORM>(cf, s.lambda, f, ut, ok);
      buf ^= 1;
    } else {
      // rows are padded past n_int with copies of the last interval, so no clamping is needed
      if constexpr (AOS && hex_reads_aos(SCHEME)) l = interval_step_aos<SCHEME, ELL0, UNIFORM>(s, i0 + sub, f, ut, ok);
      else l = interval_step<SCHEME, ELL0, UNIFORM>(s, i0 + sub, f, ut, ok);
    }
    __syncthreads();  // previous round's rows have been consumed
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {  // row-major: row rr = {L(rr,0..3)}
      *reinterpret_cast<double2*>(mine + rr * 4) = make_double2(l.v[0 * 4 + rr], l.v[1 * 4 + rr]);
      *reinterpret_cast<double2*>(mine + rr * 4 + 2) = make_double2(l.v[2 * 4 + rr], l.v[3 * 4 + rr]);
    }
    __syncthreads();
    if constexpr (kMfma) {
#pragma unroll
      for (int h = 0; h < LPF; h += 16) {
        double av[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) av[k] = rows[(h + k) * kHexLaneStride];
        if (i0 + h + 16 <= n_int) {
#pragma unroll
          for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av[k], acc, 0.0, 0, 0, 0);
        } else {
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if (i0 + h + k < n_int) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(av[k], acc, 0.0, 0, 0, 0);
        }
      }
    } else
#pragma unroll
    for (int h = 0; h < LPF; h += 16) {
      double2 lo[16], hi[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        lo[k] = *reinterpret_cast<const double2*>(rows + (h + k) * kHexLaneStride);
        hi[k] = *reinterpret_cast<const double2*>(rows + (h + k) * kHexLaneStride + 2);
      }
      if (i0 + h + 16 <= n_int) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = hex_chain_step(acc, lo[k], hi[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (i0 + h + k < n_int) acc = hex_chain_step(acc, lo[k], hi[k]);
      }
    }
  }
  __syncthreads();
  if (kMfma ? sub < 16 || LPF == 16 : sub < 16) lds[g * kGroup + e] = acc;  // column-major element c*4+r
  const unsigned long long bad = __ballot(!ok);
  __syncthreads();
  if (sub == 0 && t < n) {
    M4 a;
#pragma unroll
    for (int k = 0; k < 16; ++k) a.v[k] = lds[g * kGroup + k];
    constexpr unsigned long long kMask = LPF == 64 ? ~0ULL : ((1ULL << (LPF & 63)) - 1ULL);
    const bool group_ok = !kMfma      ? ((bad >> (g * LPF)) & kMask) == 0
                          : LPF == 16 ? ((bad >> (4 * g)) & 0x000F000F000F000FULL) == 0
                                      : ((bad >> (8 * g)) & 0x00FF00FF00FF00FFULL) == 0;
    const double det = close_determinant<ELL0>(s, f, a);
    det_out[t] = group_ok ? det : __longlong_as_double(0x7ff8000000000000LL);
    if (!group_ok) atomicOr(flags, 1);
  }
}

template <int SCHEME, bool ELL0, bool UNIFORM, int LPF, bool AOS = false>
__global__ __launch_bounds__(64, hex_min_waves(SCHEME)) void det_hex_kernel(SystemDev s0, const double* __restrict__ freq, size_t n,
                                                     double* __restrict__ det_out, int* __restrict__ flags,
                                                     const SystemDev* __restrict__ table, const int* __restrict__ sys_index) {
  __shared__ __attribute__((aligned(16))) double lds[hex_lds_doubles<SCHEME, LPF>()];
  det_hex_body<SCHEME, ELL0, UNIFORM, LPF, AOS>(s0, freq, n, det_out, flags, table, sys_index, (size_t)blockIdx.x, lds);
}

// Column-owner variant ("det_col"): 4 lanes per frequency, 16 frequencies per wave.  Phase A as in det_hex (lane q builds the
// step matrix of interval 4t+q and parks it row-major in LDS); in phase B lane c owns COLUMN c of the running product and
// reads every step matrix whole from LDS (all four lanes of a group read the same addresses: broadcast), so the ordered
// product needs no cross-lane traffic at all: 16 multiply-adds per step and lane, against det_hex's 4 multiply-adds + 8 DPP
// moves per step and lane for a quarter of the product -- 64 instead of 192 VALU instructions per evaluation spent on the
// 112-flop product (the kernels are VALU-issue bound: rocprofv3 SQ_ACTIVE_INST_VALU is 94-98 % of the launch).  A column of
// L*acc is evaluated term by term in the reference's gemm order, so the bits are the same.  It has a quarter of det_hex's
// waves, each four times as long: the launcher uses it for whole wave rounds of large batches and sends the remainder
// through det_hex (launch_hex in scheme_tu.hip).
// FMA build: phase B through v_mfma_f64_4x4x4 as in det_hex (hex_uses_mfma above): MFMA (Q, k) applies the step matrices of
// interval slot k to the products of frequencies 4Q .. 4Q+3, one product per MFMA block, so a round is four independent
// chains of four dependent MFMAs (det_hex: one chain of sixteen) and 16 MFMAs + 16 eight-byte LDS reads per lane instead
// of 64 multiply-adds + 32 sixteen-byte reads.  640 B per frequency: the four blocks of an MFMA read four different
// 128-B bank windows.  Same-box A/B against the multiply-add form (column owners alone, 12 288 / 10^5 frequencies): Colloc4
// -7 / -6 %, Colloc6 -4 / -2 %, Magnus2 -5 / -4 %, Magnus6 -1 / -3 %, Magnus8 -2 / -1 %; not Colloc2, whose det_mix came out
// 5 % slower at 10^5 frequencies with it.
__host__ __device__ constexpr bool col_uses_mfma(int scheme) {
#ifdef STORM_FAST_BUILD
  return scheme != kColloc2;
#else
  return false;
#endif
}
// doubles per frequency: four matrices of 18 (+2: bank spread; MFMA: +8)
__host__ __device__ constexpr int col_group_stride(int scheme) { return 4 * kHexLaneStride + (col_uses_mfma(scheme) ? 8 : 2); }

template <int SCHEME, bool ELL0, bool UNIFORM>
SDEV void det_col_body(const SystemDev& s, const double* __restrict__ freq, size_t n, double* __restrict__ det_out,
                       int* __restrict__ flags, size_t block, double* __restrict__ lds) {
  constexpr bool kColMfma = col_uses_mfma(SCHEME);
  constexpr int kColGroup = col_group_stride(SCHEME);
  const int tid = threadIdx.x;
  const int g = tid >> 2;  // frequency within the wave
  const int q = tid & 3;   // interval within the round (phase A), column of the product (phase B)
  const size_t t = block * 16 + g;
  const double f = freq[t < n ? t : n - 1];
  const FreqTerms ut = freq_terms<ELL0>(f, s.rotm_uniform, s.lambda);
  bool ok = true;
  // VALU phase B: column q of this frequency's product.  MFMA phase B: element (row tid / 16, column tid % 4) of the
  // products of frequencies 4Q + (tid / 4) % 4, Q = 0..3 (the D/B operand layout, hex_uses_mfma above).
  double col[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) col[r] = kColMfma ? (((tid >> 4) == q) ? 1.0 : 0.0) : ((r == q) ? 1.0 : 0.0);
  const size_t n_int = s.n_int;
  double* __restrict__ mine = lds + g * kColGroup + q * kHexLaneStride;
  [[maybe_unused]] const double* __restrict__ mats = lds + g * kColGroup;
  // MFMA A operand: S(i, k) with i = tid % 4, k = tid / 16 of frequency 4Q + (tid / 4) % 4, row-major in the hand-off
  [[maybe_unused]] const double* __restrict__ aop = lds + ((tid >> 2) & 3) * kColGroup + q * 4 + (tid >> 4);
  for (size_t i0 = 0; i0 < n_int; i0 += 4) {
    // rows are padded past n_int with copies of the last interval, so no clamping is needed
    M4 l;
    if constexpr (col_reads_aos(SCHEME)) {
      // the coefficients come from the interval-major copy: one address per lane, 16-byte loads at immediate offsets
      l = interval_step_aos<SCHEME, ELL0, UNIFORM>(s, i0 + q, f, ut, ok);
    } else {
      l = interval_step<SCHEME, ELL0, UNIFORM>(s, i0 + q, f, ut, ok);
    }
    __syncthreads();  // previous round's matrices have been consumed
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {  // row-major: row rr = {L(rr,0..3)}
      *reinterpret_cast<double2*>(mine + rr * 4) = make_double2(l.v[0 * 4 + rr], l.v[1 * 4 + rr]);
      *reinterpret_cast<double2*>(mine + rr * 4 + 2) = make_double2(l.v[2 * 4 + rr], l.v[3 * 4 + rr]);
    }
    __syncthreads();
    if constexpr (kColMfma) {
      double av[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) av[k][qq] = aop[qq * 4 * kColGroup + k * kHexLaneStride];
      if (i0 + 4 <= n_int) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) col[qq] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[k][qq], col[qq], 0.0, 0, 0, 0);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (i0 + k < n_int) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) col[qq] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[k][qq], col[qq], 0.0, 0, 0, 0);
          }
        }
      }
      continue;
    }
    double2 m[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int p = 0; p < 8; ++p) m[k][p] = *reinterpret_cast<const double2*>(mats + k * kHexLaneStride + p * 2);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < n_int) {
        double nc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // row r of L = m[k][2r] (columns 0,1), m[k][2r+1] (columns 2,3)
          double v = m[k][2 * r].x * col[0];
          v = m[k][2 * r].y * col[1] + v;
          v = m[k][2 * r + 1].x * col[2] + v;
          v = m[k][2 * r + 1].y * col[3] + v;
          nc[r] = v;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) col[r] = nc[r];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // column-major element c*4+r
    if constexpr (kColMfma) lds[(r * 4 + ((tid >> 2) & 3)) * kColGroup + q * 4 + (tid >> 4)] = col[r];
    else lds[g * kColGroup + q * 4 + r] = col[r];
  }
  const unsigned long long bad = __ballot(!ok);
  __syncthreads();
  if (q == 0 && t < n) {
    M4 a;
#pragma unroll
    for (int k = 0; k < 16; ++k) a.v[k] = lds[g * kColGroup + k];
    const bool group_ok = ((bad >> (g * 4)) & 0xfULL) == 0;
    const double det = close_determinant<ELL0>(s, f, a);
    det_out[t] = group_ok ? det : __longlong_as_double(0x7ff8000000000000LL);
    if (!group_ok) atomicOr(flags, 1);
  }
}

template <int SCHEME, bool ELL0, bool UNIFORM>
__global__ __launch_bounds__(64, hex_min_waves(SCHEME)) void det_col_kernel(SystemDev s, const double* __restrict__ freq, size_t n,
                                                                           double* __restrict__ det_out, int* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) double lds[16 * col_group_stride(SCHEME)];
  det_col_body<SCHEME, ELL0, UNIFORM>(s, freq, n, det_out, flags, (size_t)blockIdx.x, lds);
}

// Both in one launch: workgroups [0, col_blocks) run the column-owner body on the first n_col frequencies, the others the
// 16-lane det_hex body on the rest.  The dispatcher hands out workgroups in order, so the long column-owner waves start
// first and the short det_hex waves fill the SIMDs as those finish: the launch ends on short waves (a launch of
// column-owner waves alone loses its 22 % instruction saving to its ragged end).
template <int SCHEME, bool ELL0, bool UNIFORM>
__global__ __launch_bounds__(64, hex_min_waves(SCHEME)) void det_mix_kernel(SystemDev s, const double* __restrict__ freq, size_t n,
                                                                           double* __restrict__ det_out, int* __restrict__ flags,
                                                                           unsigned col_blocks) {
  constexpr int kLds = hex_lds_doubles<SCHEME, 16>() > 16 * col_group_stride(SCHEME) ? hex_lds_doubles<SCHEME, 16>() : 16 * col_group_stride(SCHEME);
  __shared__ __attribute__((aligned(16))) double lds[kLds];
  const size_t n_col = (size_t)col_blocks * 16;  // (the launcher keeps n_col <= n)
  if (blockIdx.x < col_blocks)
    det_col_body<SCHEME, ELL0, UNIFORM>(s, freq, n_col, det_out, flags, (size_t)blockIdx.x, lds);
  else
    det_hex_body<SCHEME, ELL0, UNIFORM, 16, true>(s, freq + n_col, n - n_col, det_out + n_col, flags, nullptr, nullptr,
                                            (size_t)(blockIdx.x - col_blocks), lds);
}

// ---------------------------------------------------------------------------------------------------------------------
// Segment-parallel kernels (FMA build only; the host never picks them in the parity build).
//
// One det(omega) is a chain of n_int dependent 4x4 products (reference src/solver.rs:486-492); a batch of a few
// frequencies (a bracket-refinement round, the eigenvectors of a scan's modes) leaves the GPU idle while a handful of
// 16-lane groups walk that chain (56 ns per interval: 0.56 ms for 10^4 intervals whatever the batch).  Here the interval
// axis is cut into n_seg contiguous segments of seg_len intervals and every (frequency, segment) pair gets its own
// 16-lane group, built like det_hex's: phase A = 16 step matrices per round in registers -> LDS, phase B = ordered
// product over the round.  seg_product writes the segment's transfer matrix T_s = L_{hi-1} ... L_{lo} (row-major, the
// layout of `steps`), so the ordered-product kernel (chain, common_kernels.hip) finishes the discriminant on n_seg
// "intervals" instead of n_int, and the eigenvector elimination runs on the n_seg-interval system whose unknowns are the
// solution at the segment boundaries (multiple shooting); seg_fill then carries the boundary values through each segment,
// y_{i+1} = L_i y_i, all segments at once.  Associating the product by segments changes rounding (not the algorithm),
// which is why this is not offered in the parity build.  n_seg is a multiple of 4 so that the four groups of a wave share
// their frequency (and, in a sweep, their system); segments past the end of the grid are empty (T_s = I).
// The FIRST interval is a segment of its own: the reference's elimination treats its block specially (column pivoting,
// src/solver.rs:192-230,334-339, whose second exchange leaves the already stored factors of pivot 0 in the old column
// order), so the value it returns at grid point 0 does not satisfy y_1 = L_0 y_0 in general.  With interval 0 alone in
// segment 0 the boundary-value system starts with the reference's own first block (point 0 comes out like the
// reference's) and nothing is propagated from point 0.
// ---------------------------------------------------------------------------------------------------------------------
struct SegSetup {
  size_t fi;       // frequency (mode) index, wave-uniform
  int seg;         // segment of this 16-lane group
  size_t lo, hi;   // its intervals [lo, hi)
};
SDEV SegSetup seg_setup(size_t n_int, int n_seg, int seg_len) {
  SegSetup u;
  const size_t t0 = (size_t)blockIdx.x * 4;
  u.fi = t0 / (size_t)n_seg;
  u.seg = (int)(t0 - u.fi * (size_t)n_seg) + (int)(threadIdx.x >> 4);
  // segment 0 is the first interval alone, segment s >= 1 covers [1 + (s-1) seg_len, 1 + s seg_len)
  if (u.seg == 0) {
    u.lo = 0;
    u.hi = n_int < 1 ? n_int : 1;
  } else {
    u.lo = 1 + (size_t)(u.seg - 1) * (size_t)seg_len;
    u.lo = u.lo < n_int ? u.lo : n_int;
    u.hi = u.lo + (size_t)seg_len < n_int ? u.lo + (size_t)seg_len : n_int;
  }
  return u;
}

// One round of a segment group: builds the step matrices of intervals i0 .. i0+15 (clamped to the grid: the surplus
// ones are copies that phase B never applies), parks them in LDS and returns this lane's row of each.
template <int SCHEME, bool ELL0, bool UNIFORM>
SDEV void seg_round_build(const SystemDev& s, size_t i0, int sub, double f, const FreqTerms& ut, bool& ok,
                          double* __restrict__ mine, const double* __restrict__ rows, double2 (&lo)[16], double2 (&hi)[16]) {
  size_t i = i0 + sub;
  i = i < s.n_int ? i : s.n_int - 1;
  const M4 l = interval_step<SCHEME, ELL0, UNIFORM>(s, i, f, ut, ok);
  __syncthreads();  // previous round's rows have been consumed
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {  // row-major: row rr = {L(rr,0..3)}
    *reinterpret_cast<double2*>(mine + rr * 4) = make_double2(l.v[0 * 4 + rr], l.v[1 * 4 + rr]);
    *reinterpret_cast<double2*>(mine + rr * 4 + 2) = make_double2(l.v[2 * 4 + rr], l.v[3 * 4 + rr]);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    lo[k] = *reinterpret_cast<const double2*>(rows + k * kHexLaneStride);
    hi[k] = *reinterpret_cast<const double2*>(rows + k * kHexLaneStride + 2);
  }
}

template <int SCHEME, bool ELL0, bool UNIFORM>
__global__ __launch_bounds__(64, hex_min_waves(SCHEME)) void seg_product_kernel(SystemDev s0, const double* __restrict__ freq, int n_seg, int seg_len,
                                                         double* __restrict__ partial, unsigned long long* __restrict__ growth,
                                                         int* __restrict__ flags, const SystemDev* __restrict__ table,
                                                         const int* __restrict__ sys_index) {
  constexpr int kGroup = hex_group_stride(16);
  __shared__ __attribute__((aligned(16))) double lds[4 * kGroup];
  const int tid = threadIdx.x, g = tid >> 4, sub = tid & 15, r = sub & 3, c = sub >> 2;
  const SystemDev s = pick_system(s0, table, sys_index, (size_t)blockIdx.x * 4 / (size_t)n_seg);
  const SegSetup u = seg_setup(s.n_int, n_seg, seg_len);
  const double f = freq[u.fi];
  const FreqTerms ut = freq_terms<ELL0>(f, s.rotm_uniform, s.lambda);
  bool ok = true;
  double acc = (r == c) ? 1.0 : 0.0;
  double* __restrict__ mine = lds + g * kGroup + sub * kHexLaneStride;
  const double* __restrict__ rows = lds + g * kGroup + r * 4;
  for (int rd = 0; rd < seg_len / 16; ++rd) {
    const size_t i0 = u.lo + (size_t)rd * 16;
    double2 lo[16], hi[16];
    seg_round_build<SCHEME, ELL0, UNIFORM>(s, i0, sub, f, ut, ok, mine, rows, lo, hi);
    if (i0 + 16 <= u.hi) {
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = hex_chain_step(acc, lo[k], hi[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k)
        if (i0 + k < u.hi) acc = hex_chain_step(acc, lo[k], hi[k]);
    }
  }
  const unsigned long long bad = __ballot(!ok);
  const bool group_ok = ((bad >> (g * 16)) & 0xffffULL) == 0;
  const size_t t = (size_t)blockIdx.x * 4 + g;
  partial[t * 16 + r * 4 + c] = group_ok ? acc : __longlong_as_double(0x7ff8000000000000LL);
  if (!group_ok && sub == 0) atomicOr(flags, 1);
  if (growth != nullptr) {  // max |T_s| over the segments of a mode (NaN counts as huge): the eigenvector path's guard
    unsigned long long m = (unsigned long long)__double_as_longlong(__builtin_fabs(acc));
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const unsigned long long o = __shfl_xor(m, d, 64);
      m = o > m ? o : m;
    }
    if (sub == 0) atomicMax(growth + u.fi, m);
  }
}

// y at the segment boundaries -> y at every grid point.  nodes: [mode][(n_seg + 1) * 4], vec: [mode][(n_int + 1) * 4].
// Point lo of a segment is its boundary value; points lo+1 .. hi-1 are propagated; the last grid point comes from the
// last boundary (whose fourth component the elimination fixed to 1, like the reference's back-substitution).
template <int SCHEME, bool ELL0, bool UNIFORM>
__global__ __launch_bounds__(64) void seg_fill_kernel(SystemDev s0, const double* __restrict__ freq, int n_seg, int seg_len,
                                                      const double* __restrict__ nodes, double* __restrict__ vec,
                                                      const SystemDev* __restrict__ table, const int* __restrict__ sys_index) {
  constexpr int kGroup = hex_group_stride(16);
  __shared__ __attribute__((aligned(16))) double lds[4 * kGroup];
  const int tid = threadIdx.x, g = tid >> 4, sub = tid & 15, r = sub & 3, q = sub >> 2;
  const SystemDev s = pick_system(s0, table, sys_index, (size_t)blockIdx.x * 4 / (size_t)n_seg);
  const SegSetup u = seg_setup(s.n_int, n_seg, seg_len);
  const double f = freq[u.fi];
  const FreqTerms ut = freq_terms<ELL0>(f, s.rotm_uniform, s.lambda);
  bool ok = true;
  const size_t len = (s.n_int + 1) * 4;
  double* __restrict__ e = vec + u.fi * len;
  const double* __restrict__ nd = nodes + u.fi * (size_t)(n_seg + 1) * 4;
  double y = nd[(size_t)u.seg * 4 + r];  // every quad of the group carries the same vector
  if (q == 0 && u.lo < u.hi) e[u.lo * 4 + r] = y;
  if (q == 0 && u.seg == n_seg - 1) e[s.n_int * 4 + r] = nd[(size_t)n_seg * 4 + r];
  double* __restrict__ mine = lds + g * kGroup + sub * kHexLaneStride;
  const double* __restrict__ rows = lds + g * kGroup + r * 4;
  for (int rd = 0; rd < seg_len / 16; ++rd) {
    const size_t i0 = u.lo + (size_t)rd * 16;
    // (rounds past the segment's end compute values that are never stored: the trip count stays wave-uniform, so the
    // barriers inside seg_round_build are reached by all four groups together)
    double2 lo[16], hi[16];
    seg_round_build<SCHEME, ELL0, UNIFORM>(s, i0, sub, f, ut, ok, mine, rows, lo, hi);
    // step k gives y at point i0 + k + 1; lane (q, r) keeps the steps k = 4j + q, so that for fixed j the 16 lanes of the
    // group hold 16 consecutive doubles of the eigenvector (one 128-B store)
    double out[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      y = hex_chain_step(y, lo[k], hi[k]);
      if ((k & 3) == q) out[k >> 2] = y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t point = i0 + 1 + 4 * j + q;
      if (point < u.hi) e[point * 4 + r] = out[j];
    }
  }
}

// One wave per workgroup: a workgroup of four waves needs room on all four SIMDs of a CU at once, which it does not
// find for tens of milliseconds while another handle's det_hex<colloc8> waves (one per SIMD, the whole register file)
// are in flight (kernel trace of bench.py: a 0.5 ms refinement launch took 35 ms).
constexpr int kBuildBlock = 64;
template <int SCHEME, bool ELL0, bool UNIFORM>
__global__ __launch_bounds__(kBuildBlock) void build_steps_kernel(SystemDev s0, const double* __restrict__ freq, size_t n,
                                                          double* __restrict__ steps, int* __restrict__ flags,
                                                          const SystemDev* __restrict__ table, const int* __restrict__ sys_index) {
  // grid.x covers intervals in tiles of kBuildBlock, grid.y = frequency
  const size_t i = (size_t)blockIdx.x * kBuildBlock + threadIdx.x;
  const size_t fi = blockIdx.y;
  if (fi >= n) return;
  const SystemDev s = pick_system(s0, table, sys_index, fi);
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

}  // namespace STORM_NS
