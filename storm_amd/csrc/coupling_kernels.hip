// coupling_kernels.hip -- the mode-coupling integrals of the deformed star (perturb_deformed, reference
// src/perturbed.rs:110-263) on the device.  Compiled twice, like common_kernels.hip: namespace storm with
// -ffp-contract=off (every accumulator bit-identical to the reference's sequential `+=` loop) and namespace storm_fast
// with FMA contraction allowed (option "arithmetic" = 1 of the handle that asks; tolerance in DESIGN.md section 10).
#include "postprocess.hpp"

#include <algorithm>
#include <cmath>
#include <mutex>
#include <type_traits>
#include "shared_div.hpp"

#ifndef STORM_NS
#define STORM_NS storm
#endif
#define PDEV __device__ __forceinline__

namespace STORM_NS {
using namespace storm;
namespace {
// ---------------------------------------------------------------------------------------
// Mode-coupling integrals (reference src/perturbed.rs:110-263): six M x M matrices, each element a sum over the
// radial grid of 148 flops of products of the left and right modes' post-processing fields.
// Lane = one (left, right) pair, wave = 8 x 8 tile of pairs; every lane walks the grid in index order with its
// six accumulators (the reference's `+=` loop: a sequential sum), so results are bit-identical.
// Staging: per chunk of 8 grid points the wave needs 4 fields of its 8 left modes and 5 of its 8 right modes; they
// are copied global -> LDS by LDS-DMA (global_load_lds_dword: no VGPRs, any 4-byte alignment) one chunk ahead of
// the arithmetic, so the serial walk never waits for memory.  The 19 mode-independent weights of a point have a
// wave-uniform address: they are read through the constant address space (s_load, scalar cache) into SGPRs and feed
// the FP64 instructions as scalar operands -- no LDS broadcast (that was 20 of 29 LDS reads per point), 28 fewer
// VGPRs (121: four waves per SIMD).  FP64 VALU bound (148 flops per (pair, point); fields are read once per tile).
// No MFMA: the contraction could be phrased as ~20 weighted GEMMs, but MFMA accumulates with fused multiply-adds
// in its own order and would change the last bits of every element; FP64 MFMA has the same peak as the FP64
// vector pipe on MI355X, so the only cost of staying on the VALU is the un-fused mul/add split.
// ---------------------------------------------------------------------------------------
constexpr int kCpChunk = 8;                                      // grid points per staged chunk
// Geometry of a workgroup's tile of TILE x TILE (left, right) pairs: TILE = 8 (one wave) or 16 (four waves, every wave
// an 8 x 8 block of the tile).
template <int TILE>
struct CpGeom {
  static_assert(TILE == 8 || TILE == 16, "tile of 8 or 16 modes per side");
  static constexpr int kSide = TILE / 8;                         // waves per side
  static constexpr int kWaves = kSide * kSide;
  static constexpr int kPlane = kCpChunk * TILE;                 // doubles: [point][mode of the tile]
  static constexpr int kFieldTile = 9 * kPlane;                  // 4 left + 5 right planes
  static constexpr int kDmaPerPlane = kPlane / 32;               // 256-byte DMA instructions per plane and chunk
  static constexpr int kPtsPerDma = 32 / TILE;
  static_assert(kDmaPerPlane % kWaves == 0, "the DMA instructions of a plane divide evenly over the waves");
};

typedef __attribute__((address_space(3))) void* cp_lds_ptr;
typedef __attribute__((address_space(1))) const void* cp_glb_ptr;

// Requests this wave's share of chunk [rc0, rc0 + 8) of the fields the workgroup needs into `tile` (points past the
// grid are clamped to the last one).  LDS image per plane: [point][mode], so the modes of one point sit in consecutive
// doubles (conflict-free reads, each shared by 8 lanes).  One DMA instruction moves 256 B = 32 / kCpTile points x
// kCpTile modes; lane -> (point, mode, half of the double); wave w issues instructions w, w + kCpWaves, .. of each plane.
// `lrow` / `rrow`: the fields row of the left / right mode this LANE stages (see coupling_body).
// NULL_TILE: only the four planes the shortcut of coupling_body reads (left psi, p; right xi_r, rho).
template <int TILE, bool NULL_TILE>
PDEV void coupling_stage(const CouplingArgs& a, size_t lrow, size_t rrow, size_t rc0, double* tile, int lane, int wave) {
  constexpr int kCpTile = TILE, kCpWaves = CpGeom<TILE>::kWaves, kCpPlane = CpGeom<TILE>::kPlane;
  constexpr int kCpDmaPerPlane = CpGeom<TILE>::kDmaPerPlane, kCpPtsPerDma = CpGeom<TILE>::kPtsPerDma;
  const size_t np = a.n_pts;
  const int pt_in = lane / (2 * kCpTile), word = lane & 1;
  constexpr int kPlanes[9] = {XI_R, XI_H, PSI, PRS, XI_R, XI_H, XI_TP, XI_TN, RHO};  // 4 left, 5 right
#pragma unroll
  for (int pl = 0; pl < 9; ++pl) {
    if (NULL_TILE && !(pl == 2 || pl == 3 || pl == 4 || pl == 8)) continue;
    const double* row = a.fields + ((pl < 4 ? lrow : rrow) * kPostFields + kPlanes[pl]) * np;
#pragma unroll
    for (int k = 0; k < kCpDmaPerPlane / kCpWaves; ++k) {
      const int h = wave + k * kCpWaves;  // wave-uniform
      size_t pt = rc0 + h * kCpPtsPerDma + pt_in;
      pt = pt < np ? pt : np - 1;
      const char* src = reinterpret_cast<const char*>(row + pt) + word * 4;
      __builtin_amdgcn_global_load_lds((cp_glb_ptr)src, (cp_lds_ptr)(tile + pl * kCpPlane + h * 32), 4, 0, 0);
    }
  }
}

// weights: uniform address, constant address space -> scalar loads
typedef const double __attribute__((address_space(4))) * cp_const_ptr;

#ifdef STORM_FAST_BUILD
// The matrix-pipe walk (FMA build) works on the ORDERED mode list of a problem (CouplingArgs::order, made by the host:
// coupling_order in storm_hip.hip): the caller's modes grouped by degree, every group padded to a multiple of 16
// positions, so that the 16 left modes of a tile share one degree and its 16 right modes share one -- the ten Gaunt
// factors are then constants of the tile.  order[pos] = the caller's index of the mode at padded position pos, or -1 for
// padding; the first position of a tile is never padding.  A padding position computes with its tile's first mode and
// stores nothing.  (Round 2 left the tiles that straddle two degrees to the vector walk, side by side with this one: 3 % of
// the tiles of M = 1024 modes of two degrees took 9.5 ms there, one wave per SIMD walking the grid point by point at ~3000
// cycles per point, next to 7.9 ms for the other 97 % here.)
PDEV size_t cp_mode_at(const CouplingArgs& a, size_t pos, size_t tile_first) {
  const int k = a.order[pos];
  return size_t(k >= 0 ? k : a.order[tile_first]);
}

constexpr int kCpRing = 3;  // chunks staged at once in the matrix-pipe walk (kCpRing - 1 requested ahead).  Measured, M = 1024 of one
// degree | 80 modes of three degrees repeated to M = 512 (few eligible tiles: latency counts): ring of 2: 8.2 | 7.7 ms, 3: 8.7 | 5.2 ms,
// 4: 10.8 | 5.3 ms (42 KB of LDS: three workgroups per CU instead of five).

// LDS image of a staged plane in the matrix-pipe walks: [mode][point] in groups of four modes (one LDS-DMA instruction: 4
// modes x 8 points x 8 B) that start kCpGroup = 34 doubles apart, not 32: lane (mode, point) reads mode * 64 B + point * 8 B,
// and with groups 256 B apart the four groups of a side land on the same banks (a 4-way conflict on every operand read:
// SQ_LDS_BANK_CONFLICT 8.8e8 cycles per launch at M = 1024, profiles/r03_pmc_coupling_matrix.txt); two doubles of padding
// turn the group index into a bank offset of 4, and the 32 lanes of a half-wave hit 32 distinct bank pairs.
constexpr int kCpGroup = 4 * kCpChunk + 2;
PDEV int cp_slot(int mode) { return (mode >> 2) * kCpGroup + (mode & 3) * kCpChunk; }

PDEV void coupling_matrix_body(const CouplingArgs& a, unsigned tile_l, unsigned tile_r, unsigned piece) {
  constexpr int kCpTile = 16, kCpPlane = 4 * kCpGroup, kCpFields = 9 * kCpPlane, kCpBufs = kCpRing;
  constexpr int kCpTileDoubles = kCpFields + kCpChunk * kCouplingWeights;  // fields, then the points' weights
  __shared__ __attribute__((aligned(16))) double tiles[kCpBufs][kCpTileDoubles];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const size_t M = a.n_modes, np = a.n_pts;
  const size_t l0 = (size_t)tile_l * kCpTile, r0 = (size_t)tile_r * kCpTile;  // positions in the ordered, padded list
  const size_t lfirst = size_t(a.order[l0]), rfirst = size_t(a.order[r0]);
  const double* __restrict__ q = a.consts + ((size_t)a.ell_index[lfirst] * a.n_ell + a.ell_index[rfirst]) * kCouplingConsts;
  // Degrees more than 2 apart: all ten factors are zero and, if the 32 modes are tame (see coupling_body), what is left of
  // l_zero is two instructions per four points on four of the nine planes.  Wave-uniform, the same for all four waves.
  bool null_walk;
  {
    bool all_zero = true;
#pragma unroll
    for (int k = 0; k < kCouplingConsts; ++k) all_zero = all_zero && q[k] == 0.0;
    const size_t lt = cp_mode_at(a, l0 + (lane & 15), l0), rt = cp_mode_at(a, r0 + (lane & 15), r0);
    null_walk = all_zero && a.tame != nullptr && a.weights_tame && __all(a.tame[lt] != 0 && a.tame[rt] != 0);
  }
  const double q2 = q[CQ_Q2], q2_hrd = q[CQ_Q2_HRD], q2_hld = q[CQ_Q2_HLD], q2_h = q[CQ_Q2_H];
  const double q_rt_p = q[CQ_RT_P], q_rt_n = q[CQ_RT_N], q_ht_p = q[CQ_HT_P], q_ht_n = q[CQ_HT_N];
  const double ipr = q[CQ_IPR], iph = q[CQ_IPH];
  const double k6 = ipr + 6.0 * q2;
  const int mi = lane & 15;
  // ---- Matrix-pipe walk (FMA build, tiles whose 16 left modes share one degree and whose 16 right modes share one) ----
  // With the ten factors constant over the tile every integrand is  sum_p (w_p(point) * left field_p) * right field:
  // grouped by RIGHT field, an output is one v_mfma_f64_16x16x4 per right field it involves (K = four grid points):
  //   d_squared, d_linear, d_zero: xi_r, xi_h            (3 x 2)
  //   r_linear, r_zero:            xi_r, xi_h, xi_t+, xi_t-  (2 x 4)
  //   l_zero = lz0 + F lz1 + F^2 lz2 (F = the right mode's frequency, a column scaling applied at the end):
  //            lz0: xi_r, xi_h, rho';  lz1, lz2: xi_r, xi_h   (3 + 2 + 2)
  // = 21 instructions per four points and tile, against ~60 vector instructions per (pair, point) of the walk below.
  // Operand layout (tools/_probe, measured): A[i][k] in lane i + 16k, B[k][j] in lane j + 16k, D[i][j] in lane
  // j + 16 (i % 4), element i / 4 -- lane = (mode, point) builds its own weighted left operand from the weights of ITS
  // point.  A chunk of 8 staged points = two groups of four: wave w takes group w & 1 and output half w >> 1 (0: d_*,
  // lz1, lz2; 1: r_*, lz0); the partial sums of the two groups meet in LDS at the end.  Same integrands, another
  // association of the sum over the grid: the FMA build's tolerance (tests/test_hip_coupling.py).
  // Degrees more than 2 apart (all ten factors zero, `null_walk`): what is left of l_zero is two instructions per four
  // points (rho' and xi_r) on four of the nine planes.
  {
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int grp = wave & 1, half = wave >> 1, kpt = lane >> 4;
    v4d acc[5];
#pragma unroll
    for (int o = 0; o < 5; ++o) acc[o] = v4d{0.0, 0.0, 0.0, 0.0};
    const double q2x2 = 2.0 * q2, q2hx2 = 2.0 * q2_h, k6x2 = 2.0 * k6;
    // the weights of a chunk's points (160 B each, consecutive in memory) ride along with the fields: 1280 B = five
    // LDS-DMA instructions of 256 B, issued by wave 0..3 and wave 0 again; every lane needs the weights of ITS point,
    // so the scalar loads of the vector walk (one wave-uniform point at a time) do not fit here
    auto stage_weights = [&](size_t rc0, double* tile) {
#pragma unroll
      for (int h = 0; h < 5; ++h) {
        if ((h & 3) != wave) continue;  // wave-uniform
        const int word = h * 64 + lane;                       // 4-byte word of the chunk's 8 x 40 words
        size_t pt = rc0 + (size_t)(word / (2 * kCouplingWeights));
        pt = pt < np ? pt : np - 1;
        const char* src = reinterpret_cast<const char*>(a.weights + pt * kCouplingWeights) + (word % (2 * kCouplingWeights)) * 4;
        __builtin_amdgcn_global_load_lds((cp_glb_ptr)src, (cp_lds_ptr)(tile + kCpFields + h * 32), 4, 0, 0);
      }
    };
    // A ring of kCpBufs staged chunks, kCpBufs - 1 of them requested ahead: a chunk of this walk is ~500 cycles of work
    // per wave, less than one trip to L2, so one chunk of lookahead (enough for the vector walk) leaves every chunk
    // waiting for its data.  Every chunk is requested -- past the grid's end as clamped copies of the last point, which
    // `valid` keeps out of the sums -- so the count of requests in flight is the same constant at every wait.
    constexpr int kAhead = kCpBufs - 1;
    const size_t n_chunks_all = (np - 1 + kCpChunk - 1) / kCpChunk;
    // this workgroup's piece of the grid (CouplingArgs::n_split; requests past its end fetch the next piece's points or
    // clamped copies: never used)
    const size_t chunk0 = n_chunks_all * piece / a.n_split, n_chunks = n_chunks_all * (piece + 1) / a.n_split;
    // Field staging of this walk: LDS image [plane][mode][point], one DMA instruction = 4 modes x 8 points, 16
    // consecutive lanes on the 64 consecutive bytes of one mode's chunk (the vector walk's image is [point][mode]: 2
    // points x 16 modes per instruction, i.e. 16 cache lines touched for 256 bytes -- at this walk's pace the address
    // path of those requests was the bound: 14 ms at M = 1024 of one degree whatever the lookahead).  Wave w stages
    // modes 4w .. 4w+3 of every plane.
    const size_t lstage = cp_mode_at(a, l0 + 4 * wave + (lane >> 4), l0), rstage = cp_mode_at(a, r0 + 4 * wave + (lane >> 4), r0);
    const size_t lrow_m = a.rows ? (size_t)a.rows[lstage] : lstage;
    const size_t rrow_m = a.rows ? (size_t)a.rows[rstage] : rstage;
    auto stage_fields = [&](size_t rc0, double* tile) {
      constexpr int kPlanes[9] = {XI_R, XI_H, PSI, PRS, XI_R, XI_H, XI_TP, XI_TN, RHO};  // 4 left, 5 right
      size_t pt = rc0 + (size_t)((lane & 15) >> 1);
      pt = pt < np ? pt : np - 1;
#pragma unroll
      for (int pl = 0; pl < 9; ++pl) {
        if (null_walk && !(pl == 2 || pl == 3 || pl == 4 || pl == 8)) continue;
        const double* row = a.fields + ((pl < 4 ? lrow_m : rrow_m) * kPostFields + kPlanes[pl]) * np;
        const char* src = reinterpret_cast<const char*>(row + pt) + (lane & 1) * 4;
        __builtin_amdgcn_global_load_lds((cp_glb_ptr)src, (cp_lds_ptr)(tile + pl * kCpPlane + wave * kCpGroup), 4, 0, 0);
      }
    };
    auto request = [&](size_t chunk) {
      double* tile = tiles[chunk % kCpBufs];
      stage_fields(1 + chunk * kCpChunk, tile);
      stage_weights(1 + chunk * kCpChunk, tile);
    };
    for (int k = 0; k < kAhead; ++k) request(chunk0 + k);
    for (size_t chunk = chunk0; chunk < n_chunks; ++chunk) {
      const size_t rc0 = 1 + chunk * kCpChunk;
      // requests of the kAhead - 1 later chunks may stay in flight: 9 field + 1 weight instructions each, wave 0 one more
      // (the null walk: 4 field instructions)
      if (null_walk) {
        if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kAhead - 1) * 6) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kAhead - 1) * 5) : "memory");
      } else {
        if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kAhead - 1) * 11) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kAhead - 1) * 10) : "memory");
      }
      __syncthreads();  // chunk's data is in LDS for every wave, and every wave is done with the buffer of chunk - 1
      request(chunk + kAhead);
      const double* __restrict__ t = tiles[chunk % kCpBufs];
      const int c = grp * 4 + kpt;
      const bool valid = rc0 + c < np;
      const double* __restrict__ u = t + kCpFields + c * kCouplingWeights;
      const int at = cp_slot(mi) + c;
      // a point past the grid contributes nothing: every left operand is linear in the left fields
      const double lxr = valid ? t[0 * kCpPlane + at] : 0.0, lxh = valid ? t[1 * kCpPlane + at] : 0.0;
      const double lpsi = valid ? t[2 * kCpPlane + at] : 0.0, lp = valid ? t[3 * kCpPlane + at] : 0.0;
      const double rxr = t[4 * kCpPlane + at], rxh = t[5 * kCpPlane + at];
      const double eb = u[CW_EB], rho = u[CW_RHO], mrot = u[CW_MROT], c1 = u[CW_C1], tr2 = u[CW_TR2];
      if (null_walk) {
        // l_zero += trap r^2 (psi_l rho'_r tea - (rho psi_l + p_l) xi_r adtea / x): one right field per output half
        if (half == 0) {
          acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(tr2 * (lpsi * u[CW_TEA]), t[8 * kCpPlane + at], acc[2], 0, 0, 0);
        } else {
          const double a0r = -(tr2 * ((u[CW_RX] * u[CW_ADTEA]) * (rho * lpsi + lp)));
          acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0r, rxr, acc[2], 0, 0, 0);
        }
      } else if (half == 0) {
        const double vol = u[CW_VOL];
        const double c_rr = ipr * u[CW_U1] + q2x2 * u[CW_U3], c_hh = iph * u[CW_U2] + q2hx2 * eb;
        const double adr = vol * (lxr * c_rr + lxh * (eb * q2_hld)), adh = vol * (lxh * c_hh + lxr * (eb * q2_hrd));
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(adr, rxr, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(adh, rxh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(c1 * adr, rxr, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(c1 * adh, rxh, acc[1], 0, 0, 0);
        const double c2 = u[CW_C2];
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(c2 * adr, rxr, acc[2], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(c2 * adh, rxh, acc[2], 0, 0, 0);
        // l_zero, the terms with F and F^2 of (sig rho) (sig (ipr rr + iph hh) + c1 ipr (s + hh)), sig = F - mrot
        const double tw = tr2 * rho, ci = c1 * ipr, m2 = -2.0 * mrot;
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(tw * (lxr * (m2 * ipr) + lxh * ci), rxr, acc[3], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(tw * (lxh * (m2 * iph + ci) + lxr * ci), rxh, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(tw * (lxr * ipr), rxr, acc[4], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(tw * (lxh * iph), rxh, acc[4], 0, 0, 0);
      } else {
        const double rtp = t[6 * kCpPlane + at], rtn = t[7 * kCpPlane + at];
        const double rrho = t[8 * kCpPlane + at];
        const double vol2 = u[CW_VOL2], vrot = u[CW_VOL2ROT];
        const double c_s = ipr * u[CW_U4] + q2 * u[CW_U5], c_h2 = ipr * u[CW_U2] + k6x2 * eb;
        const double bs = mrot * (vol2 * c_s), bh2 = mrot * (vol2 * c_h2);
        const double arr = lxh * bs, arh = lxr * bs + lxh * bh2;
        const double atp = vrot * (lxr * q_rt_p + lxh * q_ht_p), atn = vrot * (lxr * q_rt_n + lxh * q_ht_n);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(arr, rxr, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(arh, rxh, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(atp, rtp, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(atn, rtn, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(mrot * arr, rxr, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(mrot * arh, rxh, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(mrot * atp, rtp, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(mrot * atn, rtn, acc[1], 0, 0, 0);
        // l_zero, the terms without F
        const double ca = u[CW_ADTEA] + q2 * u[CW_ADTEB], cb = u[CW_TEA] + q2 * u[CW_TEB], rx = u[CW_RX];
        const double pp = rho * lpsi, mc = mrot * (c1 * ipr), mm = mrot * mrot;
        const double a0r = tr2 * (rho * (lxr * (mm * ipr) - lxh * mc) - (rx * ca) * (pp + lp));
        const double a0h = tr2 * (rho * (lxh * (mm * iph - mc) - lxr * mc) + (rx * (u[CW_TEB] * q2_hrd)) * (pp - lp));
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0r, rxr, acc[2], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0h, rxh, acc[2], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(tr2 * (lpsi * cb), rrho, acc[2], 0, 0, 0);
      }
    }
    // the two groups' partial sums meet in LDS, then lz0 crosses to the wave that holds lz1 and lz2.  The staging tiles
    // are free once the requests past the grid's end (issued to keep the count in flight constant) have landed.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double* __restrict__ red = &tiles[0][0];  // half 0: 5 outputs x 256 doubles, then half 1: 3 x 256
    static_assert(kCpBufs * kCpTileDoubles >= 8 * 256, "reduction scratch fits the staging tiles");
    const int n_out = half == 0 ? 5 : 3, red_base = half == 0 ? 0 : 5 * 256;
    if (grp == 1) {
#pragma unroll
      for (int o = 0; o < 5; ++o)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (o < n_out) red[red_base + (o * 4 + v) * 64 + lane] = acc[o][v];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int o = 0; o < 5; ++o)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (o < n_out) acc[o][v] += red[red_base + (o * 4 + v) * 64 + lane];
    }
    __syncthreads();
    double* __restrict__ lz0 = red;  // (every partial sum has been read)
    if (grp == 0 && half == 1) {
#pragma unroll
      for (int v = 0; v < 4; ++v) lz0[v * 64 + lane] = acc[2][v];
    }
    __syncthreads();
    if (grp == 0) {
      const size_t mm2 = M * M;
      double* __restrict__ out = a.n_split > 1 ? a.partial + size_t(piece) * a.partial_stride : a.out;
      const int rj = a.order[r0 + (size_t)(lane & 15)];        // the caller's indices; -1: padding, nothing to store
      const double F = a.freq[rj >= 0 ? rj : 0];
      bool ok = true;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int li = a.order[l0 + (size_t)((lane >> 4) + 4 * v)];
        if (rj < 0 || li < 0) continue;
        const size_t at = size_t(li) + size_t(rj) * M;
        if (null_walk) {
          if (half == 0) {
            const double lz = lz0[v * 64 + lane] + acc[2][v];
            out[0 * mm2 + at] = 0.0; out[1 * mm2 + at] = 0.0; out[2 * mm2 + at] = 0.0;
            out[5 * mm2 + at] = lz;
            ok = ok && isfinite(lz);
          } else {
            out[3 * mm2 + at] = 0.0; out[4 * mm2 + at] = 0.0;
          }
        } else if (half == 0) {
          const double lz = lz0[v * 64 + lane] + F * (acc[3][v] + F * acc[4][v]);
          out[0 * mm2 + at] = acc[0][v]; out[1 * mm2 + at] = acc[1][v]; out[2 * mm2 + at] = acc[2][v];
          out[5 * mm2 + at] = lz;
          ok = ok && isfinite(acc[0][v]) && isfinite(lz);
        } else {
          out[3 * mm2 + at] = acc[0][v]; out[4 * mm2 + at] = acc[1][v];
          ok = ok && isfinite(acc[0][v]);
        }
      }
      if (!ok) atomicOr(a.flags, 1);
    }
    return;
  }
}
// ---- The same walk on 32 x 32 tiles (round 3) ----
// A workgroup stages the fields of 32 left and 32 right positions once per chunk and each of its four waves owns one
// 16 x 16 sub-tile with ALL eight accumulators and all 21 instructions per four points (both groups of a chunk): half the
// bytes per (pair, point) of the 16 x 16 walk (its bound when the modes are all distinct: 5 B per pair and point for 0.66
// SIMD-cycles of matrix pipe), no cross-wave reduction at the end, and the addresses of the staging requests are a
// per-lane constant plus a wave-uniform chunk offset (the 16 x 16 walk recomputes and clamps them per chunk: ~6 of its ~9
// vector instructions per matrix instruction).  The two 16-blocks of a side may belong to different degrees (groups are
// padded to 16 positions, not 32): every wave has its own Gaunt factors and its own choice of the two-instruction form.
// A side's last block may be all padding (odd number of blocks): its waves stage their share and compute nothing.
constexpr int kCp32Plane = 8 * kCpGroup;                               // doubles per plane: [mode of the side][point], padded groups
constexpr int kCp32Fields = 9 * kCp32Plane;                            // 4 left + 5 right planes
constexpr int kCp32TileDoubles = kCp32Fields + kCpChunk * kCouplingWeights + 32;   // + the points' weights (+ pad to 256 B)
constexpr int kCp32Ring = 3;   // (a ring of 2 with four workgroups per CU was measured: 8.7 vs 8.2 ms at M = 1024 -- the compiler
// does not reach four waves per SIMD at 116 registers, and with two a chunk of lookahead is not enough)

PDEV void coupling_matrix32_body(const CouplingArgs& a, unsigned tile_l, unsigned tile_r, unsigned piece) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) double tiles[kCp32Ring][kCp32TileDoubles];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const int bl = wave >> 1, br = wave & 1;
  const size_t M = a.n_modes, np = a.n_pts, n_pad = a.n_pad;
  const size_t side_l = (size_t)tile_l * 32, side_r = (size_t)tile_r * 32;   // positions in the ordered, padded list
  const size_t l0 = side_l + 16 * bl, r0 = side_r + 16 * br;
  // a block past the list's end (odd number of 16-blocks) does not exist; the first position of an existing block is a mode
  const bool exists = l0 < n_pad && r0 < n_pad;
  const size_t lfirst = size_t(a.order[exists ? l0 : side_l]), rfirst = size_t(a.order[exists ? r0 : side_r]);
  const double* __restrict__ q = a.consts + ((size_t)a.ell_index[lfirst] * a.n_ell + a.ell_index[rfirst]) * kCouplingConsts;
  const double q2 = q[CQ_Q2], q2_hrd = q[CQ_Q2_HRD], q2_hld = q[CQ_Q2_HLD], q2_h = q[CQ_Q2_H];
  const double q_rt_p = q[CQ_RT_P], q_rt_n = q[CQ_RT_N], q_ht_p = q[CQ_HT_P], q_ht_n = q[CQ_HT_N];
  const double ipr = q[CQ_IPR], iph = q[CQ_IPH];
  const double k6 = ipr + 6.0 * q2;
  const double q2x2 = 2.0 * q2, q2hx2 = 2.0 * q2_h, k6x2 = 2.0 * k6;
  bool null_walk = false;
  if (exists) {
    bool all_zero = true;
#pragma unroll
    for (int k = 0; k < kCouplingConsts; ++k) all_zero = all_zero && q[k] == 0.0;
    const size_t lt = cp_mode_at(a, l0 + (lane & 15), l0), rt = cp_mode_at(a, r0 + (lane & 15), r0);
    null_walk = all_zero && a.tame != nullptr && a.weights_tame && __all(a.tame[lt] != 0 && a.tame[rt] != 0);
  }
  // the workgroup stages the four planes of the two-instruction form only if none of its waves needs more
  const bool null_stage = __syncthreads_and(null_walk || !exists) != 0;
  // ---- staging: wave w requests slots 8w .. 8w + 7 of each side, two instructions per plane (4 slots x 8 points x 8 B
  // each: 16 consecutive lanes on the 64 consecutive bytes of one mode's chunk).  Per-lane constants: the byte address
  // of point 0 of the lane's two left and two right modes' field blocks plus its 4 bytes of the 64; per chunk a
  // wave-uniform offset.
  const char* lbase[2];
  const char* rbase[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const size_t slot = size_t(8 * wave + 4 * j + (lane >> 4));
    size_t lp = side_l + slot, rp = side_r + slot;
    // (a slot past the list's end or a padding position stages its side's first mode: never read into a stored result)
    const int lk = lp < n_pad ? a.order[lp] : -1, rk = rp < n_pad ? a.order[rp] : -1;
    const size_t lmode = size_t(lk >= 0 ? lk : a.order[side_l]), rmode = size_t(rk >= 0 ? rk : a.order[side_r]);
    const size_t lrow = a.rows ? (size_t)a.rows[lmode] : lmode, rrow = a.rows ? (size_t)a.rows[rmode] : rmode;
    lbase[j] = reinterpret_cast<const char*>(a.fields + lrow * kPostFields * np) + (lane & 15) * 4;
    rbase[j] = reinterpret_cast<const char*>(a.fields + rrow * kPostFields * np) + (lane & 15) * 4;
  }
  const char* wbase = reinterpret_cast<const char*>(a.weights) + lane * 4;
  const size_t n_chunks_all = (np - 1 + kCpChunk - 1) / kCpChunk;
  const size_t chunk0 = n_chunks_all * piece / a.n_split, n_chunks = n_chunks_all * (piece + 1) / a.n_split;  // this piece
  constexpr int kAhead = kCp32Ring - 1;
  auto request = [&](size_t chunk) {
    // past the grid's end the last chunk is requested again (its data is never used): the count of requests in flight
    // stays the same constant at every wait.  The last chunk itself may reach up to 7 points past the grid: those bytes
    // belong to the next plane of the same mode (none of the staged planes is a mode's last) or to the 16 spare rows of
    // the weights, and `valid` keeps them out of the sums.
    const size_t ch = chunk < n_chunks_all ? chunk : n_chunks_all - 1;
    const size_t byte0 = (1 + ch * kCpChunk) * sizeof(double);   // wave-uniform
    double* tile = tiles[chunk % kCp32Ring];
    constexpr int kPlanes[9] = {XI_R, XI_H, PSI, PRS, XI_R, XI_H, XI_TP, XI_TN, RHO};  // 4 left, 5 right
#pragma unroll
    for (int pl = 0; pl < 9; ++pl) {
      if (null_stage && !(pl == 2 || pl == 3 || pl == 4 || pl == 8)) continue;
      const size_t off = (size_t)kPlanes[pl] * np * sizeof(double) + byte0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* src = (pl < 4 ? lbase[j] : rbase[j]) + off;
        __builtin_amdgcn_global_load_lds((cp_glb_ptr)src, (cp_lds_ptr)(tile + pl * kCp32Plane + (2 * wave + j) * kCpGroup), 4, 0, 0);
      }
    }
    // the chunk's weights: 8 points x 160 B = five instructions of 256 B, waves 0..3 and wave 0 again
    const char* wsrc = wbase + (1 + ch * kCpChunk) * (kCouplingWeights * sizeof(double));
#pragma unroll
    for (int h = 0; h < 5; ++h) {
      if ((h & 3) != wave) continue;  // wave-uniform
      __builtin_amdgcn_global_load_lds((cp_glb_ptr)(wsrc + h * 256), (cp_lds_ptr)(tile + kCp32Fields + h * 32), 4, 0, 0);
    }
  };
  v4d acc[8];  // d_squared, d_linear, d_zero, lz1, lz2, r_linear, r_zero, lz0
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = v4d{0.0, 0.0, 0.0, 0.0};
  const int mi = lane & 15, kpt = lane >> 4;
  for (int k = 0; k < kAhead; ++k) request(chunk0 + k);
  for (size_t chunk = chunk0; chunk < n_chunks; ++chunk) {
    const size_t rc0 = 1 + chunk * kCpChunk;
    if (null_stage) {
      if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kAhead - 1) * 10) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kAhead - 1) * 9) : "memory");
    } else {
      if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kAhead - 1) * 20) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kAhead - 1) * 19) : "memory");
    }
    __syncthreads();  // the chunk's data is in LDS for every wave, and every wave is done with the buffer of chunk - 1
    request(chunk + kAhead);
    if (!exists) continue;
    const double* __restrict__ t = tiles[chunk % kCp32Ring];
    const double* __restrict__ tl = t + cp_slot(16 * bl + mi);                  // this lane's left mode, plane 0
    const double* __restrict__ tr = t + 4 * kCp32Plane + cp_slot(16 * br + mi);  // this lane's right mode, plane 4
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int c = g * 4 + kpt;
      const bool valid = rc0 + c < np;
      const double* __restrict__ u = t + kCp32Fields + c * kCouplingWeights;
      // a point past the grid contributes nothing: every left operand is linear in the left fields
      const double lxr = valid ? tl[0 * kCp32Plane + c] : 0.0, lxh = valid ? tl[1 * kCp32Plane + c] : 0.0;
      const double lpsi = valid ? tl[2 * kCp32Plane + c] : 0.0, lp = valid ? tl[3 * kCp32Plane + c] : 0.0;
      const double rxr = tr[0 * kCp32Plane + c];
      const double rho = u[CW_RHO], tr2 = u[CW_TR2];
      if (null_walk) {
        // l_zero += trap r^2 (psi_l rho'_r tea - (rho psi_l + p_l) xi_r adtea / x)
        acc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(tr2 * (lpsi * u[CW_TEA]), tr[4 * kCp32Plane + c], acc[7], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(-(tr2 * ((u[CW_RX] * u[CW_ADTEA]) * (rho * lpsi + lp))), rxr, acc[7], 0, 0, 0);
        continue;
      }
      const double rxh = tr[1 * kCp32Plane + c], rtp = tr[2 * kCp32Plane + c], rtn = tr[3 * kCp32Plane + c], rrho = tr[4 * kCp32Plane + c];
      const double eb = u[CW_EB], mrot = u[CW_MROT], c1 = u[CW_C1];
      {
        const double vol = u[CW_VOL];
        const double c_rr = ipr * u[CW_U1] + q2x2 * u[CW_U3], c_hh = iph * u[CW_U2] + q2hx2 * eb;
        const double adr = vol * (lxr * c_rr + lxh * (eb * q2_hld)), adh = vol * (lxh * c_hh + lxr * (eb * q2_hrd));
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(adr, rxr, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(adh, rxh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(c1 * adr, rxr, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(c1 * adh, rxh, acc[1], 0, 0, 0);
        const double c2 = u[CW_C2];
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(c2 * adr, rxr, acc[2], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(c2 * adh, rxh, acc[2], 0, 0, 0);
        // l_zero, the terms with F and F^2 of (sig rho) (sig (ipr rr + iph hh) + c1 ipr (s + hh)), sig = F - mrot
        const double tw = tr2 * rho, ci = c1 * ipr, m2 = -2.0 * mrot;
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(tw * (lxr * (m2 * ipr) + lxh * ci), rxr, acc[3], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(tw * (lxh * (m2 * iph + ci) + lxr * ci), rxh, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(tw * (lxr * ipr), rxr, acc[4], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(tw * (lxh * iph), rxh, acc[4], 0, 0, 0);
      }
      {
        const double vol2 = u[CW_VOL2], vrot = u[CW_VOL2ROT];
        const double c_s = ipr * u[CW_U4] + q2 * u[CW_U5], c_h2 = ipr * u[CW_U2] + k6x2 * eb;
        const double bs = mrot * (vol2 * c_s), bh2 = mrot * (vol2 * c_h2);
        const double arr = lxh * bs, arh = lxr * bs + lxh * bh2;
        const double atp = vrot * (lxr * q_rt_p + lxh * q_ht_p), atn = vrot * (lxr * q_rt_n + lxh * q_ht_n);
        acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(arr, rxr, acc[5], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(arh, rxh, acc[5], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(atp, rtp, acc[5], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(atn, rtn, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(mrot * arr, rxr, acc[6], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(mrot * arh, rxh, acc[6], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(mrot * atp, rtp, acc[6], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(mrot * atn, rtn, acc[6], 0, 0, 0);
        // l_zero, the terms without F
        const double ca = u[CW_ADTEA] + q2 * u[CW_ADTEB], cb = u[CW_TEA] + q2 * u[CW_TEB], rx = u[CW_RX];
        const double pp = rho * lpsi, mc = mrot * (c1 * ipr), mm = mrot * mrot;
        const double a0r = tr2 * (rho * (lxr * (mm * ipr) - lxh * mc) - (rx * ca) * (pp + lp));
        const double a0h = tr2 * (rho * (lxh * (mm * iph - mc) - lxr * mc) + (rx * (u[CW_TEB] * q2_hrd)) * (pp - lp));
        acc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0r, rxr, acc[7], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0h, rxh, acc[7], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(tr2 * (lpsi * cb), rrho, acc[7], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the requests past the grid's end land in this workgroup's LDS
  if (!exists) return;
  const size_t mm2 = M * M;
  double* __restrict__ out = a.n_split > 1 ? a.partial + size_t(piece) * a.partial_stride : a.out;
  const int rj = a.order[r0 + (size_t)(lane & 15)];        // the caller's indices; -1: padding, nothing to store
  const double F = a.freq[rj >= 0 ? rj : 0];
  bool ok = true;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int li = a.order[l0 + (size_t)((lane >> 4) + 4 * v)];
    if (rj < 0 || li < 0) continue;
    const size_t at = size_t(li) + size_t(rj) * M;
    if (null_walk) {
      out[0 * mm2 + at] = 0.0; out[1 * mm2 + at] = 0.0; out[2 * mm2 + at] = 0.0;
      out[3 * mm2 + at] = 0.0; out[4 * mm2 + at] = 0.0;
      out[5 * mm2 + at] = acc[7][v];
      ok = ok && isfinite(acc[7][v]);
    } else {
      const double lz = acc[7][v] + F * (acc[3][v] + F * acc[4][v]);
      out[0 * mm2 + at] = acc[0][v]; out[1 * mm2 + at] = acc[1][v]; out[2 * mm2 + at] = acc[2][v];
      out[3 * mm2 + at] = acc[5][v]; out[4 * mm2 + at] = acc[6][v];
      out[5 * mm2 + at] = lz;
      ok = ok && isfinite(acc[0][v]) && isfinite(acc[5][v]) && isfinite(lz);
    }
  }
  if (!ok) atomicOr(a.flags, 1);
}
#endif

// One workgroup = one kCpTile x kCpTile tile of (left, right) pairs; its kCpWaves waves share the staged chunks (each
// field value of the tile is fetched once per workgroup) and every wave walks its own 8 x 8 block.
template <int TILE>
PDEV void coupling_body(const CouplingArgs& a, unsigned tile_l, unsigned tile_r) {
  constexpr int kCpTile = TILE, kCpSide = CpGeom<TILE>::kSide, kCpPlane = CpGeom<TILE>::kPlane;
  __shared__ __attribute__((aligned(16))) double tiles[2][CpGeom<TILE>::kFieldTile];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const int tl = (wave / kCpSide) * 8 + (lane >> 3), tr = (wave % kCpSide) * 8 + (lane & 7);
  const size_t M = a.n_modes, np = a.n_pts;
  const size_t l0 = (size_t)tile_l * kCpTile, r0 = (size_t)tile_r * kCpTile;
  size_t lrow, rrow;
  {
    const int mode_in = (lane >> 1) & (kCpTile - 1);
    size_t lmode = l0 + mode_in, rmode = r0 + mode_in;
    lmode = lmode < M ? lmode : M - 1;
    rmode = rmode < M ? rmode : M - 1;
    lrow = a.rows ? (size_t)a.rows[lmode] : lmode;
    rrow = a.rows ? (size_t)a.rows[rmode] : rmode;
  }
  const bool live = l0 + tl < M && r0 + tr < M;
  const size_t lm = l0 + tl < M ? l0 + tl : M - 1, rm = r0 + tr < M ? r0 + tr : M - 1;
  const double* __restrict__ q = a.consts + ((size_t)a.ell_index[lm] * a.n_ell + a.ell_index[rm]) * kCouplingConsts;
  const double q2 = q[CQ_Q2], q2_hrd = q[CQ_Q2_HRD], q2_hld = q[CQ_Q2_HLD], q2_h = q[CQ_Q2_H];
  const double q_rt_p = q[CQ_RT_P], q_rt_n = q[CQ_RT_N], q_ht_p = q[CQ_HT_P], q_ht_n = q[CQ_HT_N];
  const double ipr = q[CQ_IPR], iph = q[CQ_IPH];
  const double k6 = ipr + 6.0 * q2;
  const bool no_q2 = __all(q2 == 0.0 && q2_hrd == 0.0);
  const double rfreq = a.freq[rm];
  const cp_const_ptr wts = (cp_const_ptr)a.weights;
  double d_squared = 0.0, d_linear = 0.0, d_zero = 0.0, r_linear = 0.0, r_zero = 0.0, l_zero = 0.0;
  bool finite = true;
  // this lane's nine values of point c of a tile: planes 0-3 at mode tl, 4-8 at mode tr
  auto fetch = [&](const double* __restrict__ t, int c, double (&f)[9]) {
#pragma unroll
    for (int pl = 0; pl < 9; ++pl) f[pl] = t[pl * kCpPlane + c * kCpTile + (pl < 4 ? tl : tr)];
  };
  // Degrees more than 2 apart: all ten factors are zero.  With finite operands of moderate size every product that
  // contains one of them is +-0, five of the six sums stay +0.0 and what is left of l_zero's term is
  // trap r^2 ((psi_l rho'_r tea - rho psi_l xi_r adtea / r) - p_l xi_r adtea / r): adding or subtracting a zero never
  // changes a non-zero partial sum, and an accumulator that starts at +0.0 never becomes -0.0, so the signs of the
  // dropped zeros cannot reach the result.  "Moderate" (the `tame` flags: |v| < 2^200, so no dropped product overflows
  // into inf * 0 = NaN) is checked per mode by coupling_tame_kernel and for the weights on the host; anything else takes
  // the full expression below.
  const bool all_zero = q2 == 0.0 && q2_hrd == 0.0 && q2_hld == 0.0 && q2_h == 0.0 && q_rt_p == 0.0 && q_rt_n == 0.0 &&
                        q_ht_p == 0.0 && q_ht_n == 0.0 && ipr == 0.0 && iph == 0.0;
  // Different degrees (src/perturbed.rs:134-146): the inner products and the four toroidal factors are zero.  By the
  // same argument a block of such pairs skips the products that contain them (about a quarter of the instructions).
  const bool off_diag = ipr == 0.0 && iph == 0.0 && q_rt_p == 0.0 && q_rt_n == 0.0 && q_ht_p == 0.0 && q_ht_n == 0.0;
  const bool tame_block = a.tame != nullptr && a.weights_tame && __all(a.tame[lm] != 0 && a.tame[rm] != 0);
  // The walk of this WAVE's 8 x 8 block (wave-uniform): 0 = all factors zero, 1 = different degrees, 2 = the reference's
  // expression term by term, 3 = block outside the matrix (stages its share, no arithmetic).
  enum { kNull = 0, kOffDiag = 1, kFull = 2, kDead = 3 };
  const int variant = !__any(live) ? kDead : (tame_block && __all(all_zero)) ? kNull : (tame_block && __all(off_diag)) ? kOffDiag : kFull;
  // the workgroup stages the four planes of the shortcut only if none of its waves needs more
  const bool null_stage = __syncthreads_and(variant == kNull || variant == kDead) != 0;
  // One walk over the grid per variant, each with the same staging and the same number of barriers: the waves of a
  // workgroup may take different ones (the hardware barrier counts arrivals, not program addresses).
  auto walk = [&](auto null_stage_tag, auto points_of_chunk) {
    auto stage = [&](size_t rc0, double* tile) {
      coupling_stage<TILE, decltype(null_stage_tag)::value>(a, lrow, rrow, rc0, tile, lane, wave);
    };
    stage(1, tiles[0]);
    int buf = 0;
    for (size_t rc0 = 1; rc0 < np; rc0 += kCpChunk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (rc0 + kCpChunk < np) stage(rc0 + kCpChunk, tiles[buf ^ 1]);
      const int cnt = np - rc0 < (size_t)kCpChunk ? (int)(np - rc0) : kCpChunk;
      points_of_chunk(tiles[buf], rc0, cnt);
      buf ^= 1;
    }
  };
#ifdef STORM_FAST_BUILD
  // FMA build: the same integrands re-associated so that every per-pair Gaunt factor and per-point weight enters once
  // (src/perturbed.rs:157-252 term by term is the #else branch): with P = rho psi_l the six quotients of l_zero collapse to
  //   -(xi_r,r / x) (P + p_l) (adtea + q2 adteb)  +  (xi_h,r teb / x) q2_hrd (P - p_l),
  // the division by x is a multiplication by the reciprocal of the prepass, and the products xi xi of d_* / r_* are
  // weighted by per-(pair, point) coefficients formed once.  ~70 instead of ~150 FP64 instructions per (pair, point);
  // results differ from the reference's association by rounding only (tests/test_hip_coupling.py: 1e-13 of the largest
  // magnitude of each matrix).  Non-finite terms are absorbing under +=, so they are looked for once, at the end.
  const double q2x2 = 2.0 * q2, q2hx2 = 2.0 * q2_h, k6x2 = 2.0 * k6;
  (void)no_q2;
  auto null_points = [&](const double* __restrict__ t, size_t rc0, int cnt) {
    for (int c = 0; c < cnt; ++c) {
      const cp_const_ptr u = wts + (rc0 + c) * kCouplingWeights;
      const double lpsi = t[2 * kCpPlane + c * kCpTile + tl], lp = t[3 * kCpPlane + c * kCpTile + tl];
      const double rxr = t[4 * kCpPlane + c * kCpTile + tr], rrho = t[8 * kCpPlane + c * kCpTile + tr];
      const double pl = u[CW_RHO] * lpsi + lp;
      l_zero += u[CW_TR2] * (lpsi * rrho * u[CW_TEA] - (rxr * u[CW_RX]) * (pl * u[CW_ADTEA]));
    }
  };
  auto points = [&](auto off_diag_tag, const double* __restrict__ t, size_t rc0, int cnt) {
    constexpr bool OFF_DIAG = decltype(off_diag_tag)::value;
    double f[9];
    for (int c = 0; c < cnt; ++c) {
      fetch(t, c, f);
      const cp_const_ptr u = wts + (rc0 + c) * kCouplingWeights;  // wave-uniform
      const double lxr = f[0], lxh = f[1], lpsi = f[2], lp = f[3];
      const double rxr = f[4], rxh = f[5], rtpv = f[6], rtnv = f[7], rrho = f[8];
      const double rr = lxr * rxr, rh = lxr * rxh, hr = lxh * rxr, hh = lxh * rxh;
      const double eb = u[CW_EB], rho = u[CW_RHO], mrot = u[CW_MROT];
      const double s = rh + hr;
      // d_squared, d_linear, d_zero
      double c_rr = q2x2 * u[CW_U3], c_hh = q2hx2 * eb;
      if constexpr (!OFF_DIAG) { c_rr = ipr * u[CW_U1] + c_rr; c_hh = iph * u[CW_U2] + c_hh; }
      const double val = u[CW_VOL] * (rr * c_rr + hh * c_hh + eb * (rh * q2_hrd + hr * q2_hld));
      d_squared += val;
      d_linear += u[CW_C1] * val;
      d_zero += u[CW_C2] * val;
      // r_linear, r_zero
      double c_s = q2 * u[CW_U5], c_h2 = k6x2 * eb;
      if constexpr (!OFF_DIAG) { c_s = ipr * u[CW_U4] + c_s; c_h2 = ipr * u[CW_U2] + c_h2; }
      double w = mrot * (u[CW_VOL2] * (s * c_s + hh * c_h2));
      if constexpr (!OFF_DIAG)
        w = u[CW_VOL2ROT] * (lxr * (q_rt_p * rtpv + q_rt_n * rtnv) + lxh * (q_ht_p * rtpv + q_ht_n * rtnv)) + w;
      r_linear += w;
      r_zero += mrot * w;
      // l_zero
      const double pp = rho * lpsi;
      const double ca = u[CW_ADTEA] + q2 * u[CW_ADTEB], cb = u[CW_TEA] + q2 * u[CW_TEB];
      const double rx = u[CW_RX];
      double acc = lpsi * rrho * cb - (rxr * rx) * ((pp + lp) * ca) + (rxh * rx) * ((u[CW_TEB] * q2_hrd) * (pp - lp));
      if constexpr (!OFF_DIAG) {
        const double sig = rfreq - mrot;
        acc = (sig * rho) * (sig * (ipr * rr + iph * hh) + (u[CW_C1] * ipr) * (s + hh)) + acc;
      }
      l_zero += u[CW_TR2] * acc;
    }
  };
#else
  auto null_points = [&](const double* __restrict__ t, size_t rc0, int cnt) {
    for (int c = 0; c < cnt; ++c) {
      const cp_const_ptr u = wts + (rc0 + c) * kCouplingWeights;
      const double lpsi = t[2 * kCpPlane + c * kCpTile + tl], lp = t[3 * kCpPlane + c * kCpTile + tl];
      const double rxr = t[4 * kCpPlane + c * kCpTile + tr], rrho = t[8 * kCpPlane + c * kCpTile + tr];
      const double adtea = u[CW_ADTEA];
      double t0 = u[CW_RHO] * lpsi * rxr * adtea, t2 = lp * rxr * adtea;
      shared_div2(t0, t2, SharedDenominator{u[CW_X], u[CW_RX], shared_div_moderate(u[CW_X])});
      l_zero += u[CW_TR2] * (lpsi * rrho * u[CW_TEA] - t0 - t2);
    }
  };
  auto points = [&](auto off_diag_tag, const double* __restrict__ t, size_t rc0, int cnt) {  // src/perturbed.rs:148-261
    constexpr bool OFF_DIAG = decltype(off_diag_tag)::value;
    double f[9];
    for (int c = 0; c < cnt; ++c) {
      fetch(t, c, f);
      const cp_const_ptr u = wts + (rc0 + c) * kCouplingWeights;  // wave-uniform
      const double lxr = f[0], lxh = f[1], lpsi = f[2], lp = f[3];
      const double rxr = f[4], rxh = f[5], rtpv = f[6], rtnv = f[7], rrho = f[8];
      const double rr = lxr * rxr, rh = lxr * rxh, hr = lxh * rxr, hh = lxh * rxh;
      const double eb = u[CW_EB], rho = u[CW_RHO], x = u[CW_X];
      double val;
      if constexpr (OFF_DIAG)
        val = u[CW_VOL] * (rr * 2.0 * q2 * u[CW_U3] + (rh * q2_hrd + hr * q2_hld + hh * 2.0 * q2_h) * eb);
      else
        val = u[CW_VOL] * (rr * ipr * u[CW_U1] + hh * iph * u[CW_U2] + rr * 2.0 * q2 * u[CW_U3] +
                           (rh * q2_hrd + hr * q2_hld + hh * 2.0 * q2_h) * eb);
      finite = finite && isfinite(val);
      d_squared += val;
      d_linear += u[CW_C1] * val;
      d_zero += u[CW_C2] * val;
      const double mrot = u[CW_MROT];
      if constexpr (OFF_DIAG) {
        val = u[CW_VOL2] * ((rh + hr) * q2 * u[CW_U5] + hh * k6 * 2.0 * eb);
        finite = finite && isfinite(val);
        r_linear += mrot * val;
        r_zero += mrot * (mrot * val);
      } else {
        const double rtp = lxr * rtpv, rtn = lxr * rtnv, htp = lxh * rtpv, htn = lxh * rtnv;
        val = u[CW_VOL2] * (ipr * ((hr + rh) * u[CW_U4] + hh * u[CW_U2]) + (rh + hr) * q2 * u[CW_U5] + hh * k6 * 2.0 * eb);
        const double val_t = u[CW_VOL2ROT] * (q_rt_p * rtp + q_rt_n * rtn + q_ht_n * htn + q_ht_p * htp);
        finite = finite && isfinite(val) && isfinite(val_t);
        r_linear += mrot * val + val_t;
        r_zero += mrot * (mrot * val + val_t);
      }
      const double tea = u[CW_TEA], adtea = u[CW_ADTEA], teb = u[CW_TEB], adteb = u[CW_ADTEB];
      const SharedDenominator den{x, u[CW_RX], shared_div_moderate(x)};  // reciprocal: coupling_recip_kernel
      double t0, t1, t2, t3, t4, t5;
      if (no_q2) {
        // the four terms with a factor q2 or q2_hrd are +-0 (or NaN) for every lane of this wave and dividing them by
        // the radius returns them unchanged: two quotients are left
        t0 = rho * lpsi * rxr * adtea;
        t2 = lp * rxr * adtea;
        shared_div2(t0, t2, den);
        t1 = rho * lpsi * q2 * rxr * adteb;
        t3 = lp * q2 * rxr * adteb;
        t4 = lp * q2_hrd * rxh * teb;
        t5 = q2_hrd * rho * lpsi * rxh * teb;
      } else {
        double dv[6] = {rho * lpsi * rxr * adtea, rho * lpsi * q2 * rxr * adteb, lp * rxr * adtea,
                        lp * q2 * rxr * adteb,    lp * q2_hrd * rxh * teb,       q2_hrd * rho * lpsi * rxh * teb};
        shared_div6(dv, den);
        t0 = dv[0]; t1 = dv[1]; t2 = dv[2]; t3 = dv[3]; t4 = dv[4]; t5 = dv[5];
      }
      if constexpr (OFF_DIAG) {
        l_zero += u[CW_TR2] * (lpsi * rrho * tea - t0 + lpsi * q2 * rrho * teb - t1 - t2 - t3 - t4 + t5);
      } else {
        const double sig = rfreq - mrot;
        l_zero += u[CW_TR2] * ((sig * sig) * rho * (ipr * rr + iph * hh) + u[CW_C1] * sig * rho * ipr * (rh + hr + hh) +
                               lpsi * rrho * tea - t0 + lpsi * q2 * rrho * teb - t1 - t2 - t3 - t4 + t5);
      }
      finite = finite && isfinite(l_zero);
    }
  };
#endif
  auto no_points = [](const double*, size_t, int) {};
  if (variant == kOffDiag) walk(std::false_type{}, [&](const double* t, size_t rc0, int cnt) { points(std::true_type{}, t, rc0, cnt); });
  else if (variant == kFull) walk(std::false_type{}, [&](const double* t, size_t rc0, int cnt) { points(std::false_type{}, t, rc0, cnt); });
  else if (variant == kNull && null_stage) walk(std::true_type{}, null_points);
  else if (variant == kNull) walk(std::false_type{}, null_points);
  else if (null_stage) walk(std::true_type{}, no_points);
  else walk(std::false_type{}, no_points);
  if (variant == kNull) finite = isfinite(l_zero);  // non-finite is absorbing under +=: the last value tells
#ifdef STORM_FAST_BUILD
  else finite = isfinite(d_squared) && isfinite(r_linear) && isfinite(l_zero);
#endif
  if (live) {
    const size_t at = lm + rm * M, mm = M * M;
    a.out[0 * mm + at] = d_squared; a.out[1 * mm + at] = d_linear; a.out[2 * mm + at] = d_zero;
    a.out[3 * mm + at] = r_linear; a.out[4 * mm + at] = r_zero; a.out[5 * mm + at] = l_zero;
    if (!finite) atomicOr(a.flags, 1);
  }
}


// Column CW_RX of the weights: the refined reciprocal of the radius every division of a point shares (shared_div.hpp),
// computed once per point here instead of once per (tile, point) in the walk -- the same instructions on the same
// input, so the same bits.
PDEV void coupling_recip_body(const CouplingArgs& a) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < a.n_pts) a.weights[i * kCouplingWeights + CW_RX] = shared_div_prepare(a.weights[i * kCouplingWeights + CW_X]).r;
}

__global__ __launch_bounds__(256) void coupling_recip_kernel(CouplingArgs a) { coupling_recip_body(a); }

__global__ __launch_bounds__(256) void coupling_recip_batch_kernel(const CouplingArgs* __restrict__ table) {
  const CouplingArgs a = table[blockIdx.y];
  if (a.n_modes) coupling_recip_body(a);
}

// tame[k] = 1 if the frequency of mode k and every value of the seven field planes the integrals read are finite and
// smaller than 2^200 in magnitude (exponent test on the high words).  One workgroup per mode.
constexpr unsigned kTameLimit = unsigned(1023 + 200) << 20;
PDEV void coupling_tame_body(const CouplingArgs& a, size_t k) {
  __shared__ unsigned part[4];
  const size_t row = a.rows ? (size_t)a.rows[k] : k;
  constexpr int kPlanes[7] = {XI_R, XI_H, PSI, PRS, XI_TP, XI_TN, RHO};
  unsigned hi = unsigned(__double2hiint(a.freq[k])) & 0x7fffffffu;
  for (int pl = 0; pl < 7; ++pl) {
    const double* __restrict__ v = a.fields + (row * kPostFields + kPlanes[pl]) * a.n_pts;
    for (size_t i = threadIdx.x; i < a.n_pts; i += 256) {
      const unsigned w = unsigned(__double2hiint(v[i])) & 0x7fffffffu;
      hi = w > hi ? w : hi;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)hi, off);
    hi = o > hi ? o : hi;
  }
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = hi;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) hi = part[w] > hi ? part[w] : hi;
    a.tame[k] = hi < kTameLimit ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void coupling_tame_kernel(CouplingArgs a) { coupling_tame_body(a, blockIdx.x); }

__global__ __launch_bounds__(256) void coupling_tame_batch_kernel(const CouplingArgs* __restrict__ table) {
  const CouplingArgs a = table[blockIdx.y];
  if (a.tame == nullptr || blockIdx.x >= a.n_modes) return;
  coupling_tame_body(a, blockIdx.x);
}

#ifndef STORM_CP_MINWAVES16
#define STORM_CP_MINWAVES16 4
#endif
template <int TILE>
__global__ __launch_bounds__(64 * CpGeom<TILE>::kWaves, TILE == 16 ? STORM_CP_MINWAVES16 : 4) void coupling_kernel(CouplingArgs a) {
  coupling_body<TILE>(a, blockIdx.x, blockIdx.y);
}

// blockIdx.z = problem; the grid covers the largest one, tiles past a problem's own size leave at once
template <int TILE>
__global__ __launch_bounds__(64 * CpGeom<TILE>::kWaves, TILE == 16 ? STORM_CP_MINWAVES16 : 4) void coupling_batch_kernel(const CouplingArgs* __restrict__ table) {
  const CouplingArgs a = table[blockIdx.z];
  const size_t tiles = (a.n_modes + TILE - 1) / TILE;
  if (blockIdx.x >= tiles || blockIdx.y >= tiles) return;
  coupling_body<TILE>(a, blockIdx.x, blockIdx.y);
}
#ifdef STORM_FAST_BUILD
// The matrix-pipe walk over the ordered mode list: ceil(n_pad / TILE)^2 tiles per problem.
// Which tile a workgroup takes.  Workgroups go to the chip's 8 XCDs round-robin in launch order (workgroup L runs on XCD
// L % 8) and every XCD has an L2 of its own, so the tiles one XCD works on at a time should share their modes: the tiles
// are numbered super-tile by super-tile (8 x 8 tiles: the 64 workgroups of one read 8 left and 8 right sides between them
// instead of 64 of each) and every XCD takes a contiguous eighth of that order.  With blockIdx.x = left tile, as before,
// the 8 workgroups that share a right side sat on 8 different XCDs (M = 1024, 32 x 32 tiles: 4.3 GB fetched; 1.6 GB with
// this order, profiles/r03_pmc_coupling_matrix.txt).
constexpr unsigned kCpXcds = 8, kCpSuper = 8;
// Tile number q of the T x T grid in super-tile-major order (rows of super-tiles top to bottom, the super-tiles of a row
// left to right, the tiles of a super-tile column by column); the grid's last row and column of super-tiles may be narrower.
PDEV void cp_tile_at(unsigned q, unsigned T, unsigned& tl, unsigned& tr) {
  const unsigned S = (T + kCpSuper - 1) / kCpSuper;
  const unsigned si = min(q / (kCpSuper * T), S - 1), h = min(kCpSuper, T - kCpSuper * si);
  const unsigned r = q - si * kCpSuper * T;
  const unsigned sj = min(r / (h * kCpSuper), S - 1), w = min(kCpSuper, T - kCpSuper * sj);
  const unsigned r2 = r - sj * h * kCpSuper;
  tl = kCpSuper * si + r2 % h;
  tr = kCpSuper * sj + r2 / h;
  (void)w;
}
// XCD `xcd` of n_xcd takes a contiguous, equal share of that order (so its 64 resident workgroups work on one or two
// super-tiles at a time, and every XCD gets the same number of tiles to within one); k = the workgroup's number on its XCD.
PDEV bool cp_tile_of(unsigned xcd, unsigned k, unsigned n_xcd, unsigned T, unsigned& tl, unsigned& tr) {
  const unsigned n = T * T, share = (n + n_xcd - 1) / n_xcd;
  const unsigned q = xcd * share + k;
  if (k >= share || q >= n) return false;
  cp_tile_at(q, T, tl, tr);
  return true;
}
inline unsigned cp_grid_single(unsigned T) {  // workgroups of a one-problem launch
  return (T * T + kCpXcds - 1) / kCpXcds * kCpXcds;
}
inline unsigned cp_per_problem(unsigned T) { return T * T; }  // workgroups per problem of a batch launch (a problem stays on one XCD)
template <int TILE>
PDEV void coupling_matrix_dispatch(const CouplingArgs& a, unsigned tl, unsigned tr, unsigned piece) {
  if constexpr (TILE == 16) coupling_matrix_body(a, tl, tr, piece);
  else coupling_matrix32_body(a, tl, tr, piece);
}
// One problem: `per` = cp_grid_single(T) workgroups per piece of the grid, the pieces one after the other in launch order
// (per is a multiple of 8: a tile's pieces stay on the tile's XCD).
template <int TILE>
__global__ __launch_bounds__(256, 2) void coupling_matrix_kernel(CouplingArgs a, unsigned per) {
  const unsigned piece = blockIdx.x / per, b = blockIdx.x % per;
  unsigned tl, tr;
  if (!cp_tile_of(b % kCpXcds, b / kCpXcds, kCpXcds, (unsigned)((a.n_pad + TILE - 1) / TILE), tl, tr)) return;
  coupling_matrix_dispatch<TILE>(a, tl, tr, piece);
}
// out[i] = partial[0][i] + partial[1][i] + ... (ascending: the same sum on every run), i < n; a non-finite sum raises the flag.
__global__ __launch_bounds__(256) void coupling_pieces_kernel(const double* __restrict__ partial, unsigned n_split, size_t n,
                                                              double* __restrict__ out, int* flags) {
  const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  bool ok = true;
  if (i < n) {
    double s = partial[i];
    for (unsigned k = 1; k < n_split; ++k) s += partial[size_t(k) * n + i];
    out[i] = s;
    ok = isfinite(s);
  }
  if (__syncthreads_or(!ok) && threadIdx.x == 0) atomicOr(flags, 1);
}
// several problems: problem p runs on XCD p % 8 (its modes are read by no other problem), per_problem workgroups each
// (`base` workgroups per piece of the grid, the pieces one after the other: base is a multiple of 8)
template <int TILE>
__global__ __launch_bounds__(256, 2) void coupling_matrix_batch_kernel(const CouplingArgs* __restrict__ table, unsigned n_problems,
                                                                       unsigned per_problem, unsigned base) {
  const unsigned piece = blockIdx.x / base, b = blockIdx.x % base;
  const unsigned xcd = b % kCpXcds, k = b / kCpXcds;
  const unsigned p = (k / per_problem) * kCpXcds + xcd;
  if (p >= n_problems) return;
  const CouplingArgs a = table[p];
  unsigned tl, tr;
  if (!cp_tile_of(0, k % per_problem, 1, (unsigned)((a.n_pad + TILE - 1) / TILE), tl, tr)) return;
  coupling_matrix_dispatch<TILE>(a, tl, tr, piece);
}
// 32 x 32 tiles from this many workgroups on (one per CU and a few more): below, the 16 x 16 walk's four times as many
// smaller workgroups fill the chip better.  0 < `force` = 16 or 32 overrides (option "coupling_matrix_tile").
inline bool cp_use_tiles32(size_t tiles32_total, int force) {
  if (force == 16) return false;
  if (force == 32) return true;
  return tiles32_total >= 384;
}
#endif
}  // namespace

#ifdef STORM_FAST_BUILD
// Workgroups of the matrix-pipe walks that the chip holds at once: 32 x 32 tiles 2 per CU (208 registers per lane), 16 x 16
// tiles 3 per CU (168 registers; 33 KB of LDS each).
constexpr unsigned kCpSlots32 = 2 * 256, kCpSlots16 = 3 * 256;
// Pieces of the grid per tile: the fewest that make (tiles x pieces) at least twelve rounds over the `slots` resident
// workgroups -- tiles differ in cost (a tile of degrees more than 2 apart takes 2 of the 21 instructions) and a launch ends
// with the chip waiting for its last workgroups, so what evens it out is many more, shorter workgroups than slots, not a
// whole number of rounds -- capped at 32 and at pieces of 24 chunks (192 points): the ring's fill costs about five chunks'
// worth per piece.  Measured, tile 32, one problem (profiles/r03_coupling_split_sweep.txt): M = 80 2.15 -> 0.21 ms (32
// pieces), 240: 2.16 -> 0.70 (32), 512: 3.39 -> 2.04 (6) | 2.08 (16), 1024: 8.34 -> 6.99 (3 or 6), 2048: 26.4 -> 25.6 (3) |
// 26.3 (2).
inline unsigned cp_pieces(size_t tiles, unsigned slots, size_t n_chunks) {
  const size_t most = std::max<size_t>(1, std::min<size_t>(32, n_chunks / 24));
  const size_t want = (12 * size_t(slots) + tiles - 1) / std::max<size_t>(tiles, 1);
  return (unsigned)std::min(most, std::max<size_t>(want, 1));
}
}  // namespace STORM_NS
namespace storm {
CouplingPlan plan_coupling_matrix(size_t n_pad, size_t n_pts, int matrix_tile, int split) {
  using namespace storm_fast;
  const size_t t16 = n_pad / 16, t32 = (n_pad + 31) / 32;
  const size_t n_chunks = n_pts > 1 ? (n_pts - 1 + kCpChunk - 1) / kCpChunk : 1;
  CouplingPlan plan;
  // 32 x 32 tiles: half the staged bytes and a third fewer vector instructions per matrix instruction; with the grid cut
  // into pieces the 16 x 16 walk's larger number of workgroups no longer counts for small problems
  plan.tile = (matrix_tile == 16 || matrix_tile == 32) ? matrix_tile : 32;
  plan.n_split = plan.tile == 32 ? cp_pieces(t32 * t32, kCpSlots32, n_chunks) : cp_pieces(t16 * t16, kCpSlots16, n_chunks);
  if (split > 0) plan.n_split = (unsigned)std::min<size_t>(size_t(split), n_chunks);
  return plan;
}
// ... and for n_problems problems of at most max_pad positions in one launch (tile: as launch_coupling_batch chose before
// there were pieces; pieces: the same twelve-rounds rule over all problems' tiles)
CouplingPlan plan_coupling_batch(size_t max_pad, size_t n_problems, size_t n_pts, int matrix_tile, int split) {
  using namespace storm_fast;
  const size_t t16 = max_pad / 16, t32 = (max_pad + 31) / 32;
  const size_t n_chunks = n_pts > 1 ? (n_pts - 1 + kCpChunk - 1) / kCpChunk : 1;
  CouplingPlan plan;
  plan.tile = cp_use_tiles32(t32 * t32 * n_problems, matrix_tile) ? 32 : 16;
  plan.n_split = plan.tile == 32 ? cp_pieces(t32 * t32 * n_problems, kCpSlots32, n_chunks) : cp_pieces(t16 * t16 * n_problems, kCpSlots16, n_chunks);
  if (split > 0) plan.n_split = (unsigned)std::min<size_t>(size_t(split), n_chunks);
  return plan;
}
}  // namespace storm
namespace STORM_NS {
#endif

hipError_t launch_coupling_batch(const CouplingArgs* table, size_t n_problems, size_t max_modes, size_t n_pts,
                                 hipStream_t stream, bool with_tame, int tile, size_t max_pad, int matrix_tile,
                                 unsigned n_split, const double* partial, double* out_all, size_t out_total, int* flags) {
  if (n_problems == 0 || max_modes == 0) return hipSuccess;
  if (tile != 8 && tile != 16) return hipErrorInvalidValue;
  const unsigned tiles = (unsigned)((max_modes + tile - 1) / tile);
  if (tiles > 65535 || n_problems > 65535) return hipErrorInvalidValue;
  if (with_tame) {
    coupling_tame_batch_kernel<<<dim3((unsigned)max_modes, (unsigned)n_problems, 1), 256, 0, stream>>>(table);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  {
    coupling_recip_batch_kernel<<<dim3((unsigned)((n_pts + 255) / 256), (unsigned)n_problems, 1), 256, 0, stream>>>(table);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
#ifdef STORM_FAST_BUILD
  if (max_pad) {  // every entry carries an ordered mode list: the matrix-pipe walk takes all tiles
    const unsigned ptiles = (unsigned)(max_pad / 16), ptiles32 = (unsigned)((max_pad + 31) / 32);
    if (ptiles > 65535) return hipErrorInvalidValue;
    const unsigned rounds = (unsigned)((n_problems + kCpXcds - 1) / kCpXcds);   // problems per XCD
    const size_t n_chunks = (n_pts - 1 + kCpChunk - 1) / kCpChunk;
    if (n_split == 0 || n_split > n_chunks || (n_split > 1 && (!partial || !out_all || !flags))) return hipErrorInvalidValue;
    const bool t32 = cp_use_tiles32(size_t(ptiles32) * ptiles32 * n_problems, matrix_tile);
    const unsigned per = cp_per_problem(t32 ? ptiles32 : ptiles);
    if (size_t(rounds) * per * kCpXcds * n_split > 0x7fffffffu) return hipErrorInvalidValue;
    const unsigned base = rounds * per * kCpXcds;
    if (t32) coupling_matrix_batch_kernel<32><<<dim3(base * n_split, 1, 1), 256, 0, stream>>>(table, (unsigned)n_problems, per, base);
    else coupling_matrix_batch_kernel<16><<<dim3(base * n_split, 1, 1), 256, 0, stream>>>(table, (unsigned)n_problems, per, base);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess || n_split == 1) return e;
    coupling_pieces_kernel<<<(unsigned)((out_total + 255) / 256), 256, 0, stream>>>(partial, n_split, out_total, out_all, flags);
    return hipGetLastError();
  }
#else
  (void)max_pad; (void)matrix_tile; (void)n_split; (void)partial; (void)out_all; (void)out_total; (void)flags;
#endif
  if (tile == 8) coupling_batch_kernel<8><<<dim3(tiles, tiles, (unsigned)n_problems), 64, 0, stream>>>(table);
  else coupling_batch_kernel<16><<<dim3(tiles, tiles, (unsigned)n_problems), 256, 0, stream>>>(table);
  return hipGetLastError();
}

hipError_t launch_coupling(const CouplingArgs& a, int tile, int matrix_tile) {
  if (a.n_modes == 0) return hipSuccess;
  if (a.n_pts < 2 || (tile != 8 && tile != 16)) return hipErrorInvalidValue;
  const unsigned tiles = (unsigned)((a.n_modes + tile - 1) / tile);
  if (tiles > 65535) return hipErrorInvalidValue;
  if (a.tame) {
    coupling_tame_kernel<<<(unsigned)a.n_modes, 256, 0, a.stream>>>(a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  coupling_recip_kernel<<<(unsigned)((a.n_pts + 255) / 256), 256, 0, a.stream>>>(a);
  {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
#ifdef STORM_FAST_BUILD
  if (a.order) {  // the matrix-pipe walk over the ordered mode list takes all tiles
    const unsigned ptiles = (unsigned)(a.n_pad / 16), ptiles32 = (unsigned)((a.n_pad + 31) / 32);
    if (a.n_pad % 16 || ptiles == 0 || ptiles > 65535) return hipErrorInvalidValue;
    const size_t n_chunks = (a.n_pts - 1 + kCpChunk - 1) / kCpChunk;
    if (a.n_split == 0 || a.n_split > n_chunks || (a.n_split > 1 && !a.partial) || (matrix_tile != 16 && matrix_tile != 32))
      return hipErrorInvalidValue;   // (the caller passes plan_coupling_matrix's tile and pieces)
    const unsigned per = cp_grid_single(matrix_tile == 32 ? ptiles32 : ptiles);
    if (size_t(per) * a.n_split > 0x7fffffffu) return hipErrorInvalidValue;
    if (matrix_tile == 32) coupling_matrix_kernel<32><<<dim3(per * a.n_split, 1, 1), 256, 0, a.stream>>>(a, per);
    else coupling_matrix_kernel<16><<<dim3(per * a.n_split, 1, 1), 256, 0, a.stream>>>(a, per);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.n_split == 1) return e;
    const size_t n_out = size_t(kCouplingMatrices) * a.n_modes * a.n_modes;
    coupling_pieces_kernel<<<(unsigned)((n_out + 255) / 256), 256, 0, a.stream>>>(a.partial, a.n_split, n_out, a.out, a.flags);
    return hipGetLastError();
  }
#else
  (void)matrix_tile;
#endif
  if (tile == 8) coupling_kernel<8><<<dim3(tiles, tiles, 1), 64, 0, a.stream>>>(a);
  else coupling_kernel<16><<<dim3(tiles, tiles, 1), 256, 0, a.stream>>>(a);
  return hipGetLastError();
}

}  // namespace STORM_NS
