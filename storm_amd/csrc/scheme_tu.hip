// scheme_tu.hip -- instantiates the step-matrix kernels for ONE difference scheme.
// Compiled 8 times: hipcc -DSTORM_SCHEME=<0..7> ...
#include "scheme_kernels.hpp"

#ifndef STORM_SCHEME
#error "compile with -DSTORM_SCHEME=<0..7>"
#endif

#include <algorithm>
#include <atomic>

namespace STORM_NS {
using namespace storm;

#define STORM_CAT2(a, b) a##b
#define STORM_CAT(a, b) STORM_CAT2(a, b)

template <bool ELL0, bool UNIFORM>
static hipError_t launch_fused_v(const LaunchArgs& a) {
  const unsigned blocks = (unsigned)((a.n + 63) / 64);
  det_fused_kernel<STORM_SCHEME, ELL0, UNIFORM><<<blocks, 64, 0, a.stream>>>(a.sys, a.freq, a.n, a.out, a.flags);
  return hipGetLastError();
}
template <bool ELL0, bool UNIFORM>
static hipError_t launch_quad_v(const LaunchArgs& a) {
  const unsigned blocks = (unsigned)((a.n + 15) / 16);
  det_quad_kernel<STORM_SCHEME, ELL0, UNIFORM><<<blocks, 64, 0, a.stream>>>(a.sys, a.freq, a.n, a.out, a.flags);
  return hipGetLastError();
}
// Waves of det_hex all run for the same time, so a launch takes ceil(waves / slots) "wave rounds".
// LPF lanes per frequency gives n*LPF/64 waves of n_int/LPF rounds, each costing the step matrix
// (F_scheme instructions) plus LPF ordered-product steps (7 flops + DPP moves + LDS reads, weighed as 20
// instructions: calibrated on the box, scripts/kernel_bench.py --lanes): pick the cheaper packing.
template <bool ELL0, bool UNIFORM, int LPF>
static size_t hex_wave_slots() {
  // per device: the current device's CU count x resident workgroups per CU (cached; 0 = not asked yet)
  constexpr int kMaxDev = 64;
  static std::atomic<size_t> cache[kMaxDev];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return size_t(1024);
  const bool cached = dev >= 0 && dev < kMaxDev;
  if (cached) { const size_t v = cache[dev].load(std::memory_order_relaxed); if (v) return v; }
  int cus = 0, per_cu = 0;
  size_t slots = 1024;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, det_hex_kernel<STORM_SCHEME, ELL0, UNIFORM, LPF>, 64, 0) == hipSuccess)
    slots = size_t(cus > 0 ? cus : 256) * size_t(per_cu > 0 ? per_cu : 4);
  if (cached) cache[dev].store(slots, std::memory_order_relaxed);
  return slots;
}
template <bool ELL0, bool UNIFORM, int LPF>
static double hex_cost(size_t n, size_t n_int) {
  constexpr double kStepFlops[8] = {610., 1396., 2694., 5048., 1486., 1848., 2562., 3900.};
  const size_t waves = (n * LPF + 63) / 64, slots = hex_wave_slots<ELL0, UNIFORM, LPF>();
  // (an MFMA step of the ordered product, FMA build: one 16-cycle instruction per wave and an 8-byte LDS read per lane)
  const double step_cost = hex_uses_mfma(STORM_SCHEME, LPF) ? 8.0 : 20.0;
  return double((waves + slots - 1) / slots) * double((n_int + LPF - 1) / LPF) * (kStepFlops[STORM_SCHEME] + step_cost * LPF);
}
// A batch of more waves than the GPU holds goes out in pieces of whole wave rounds.  All waves of a piece end together
// (equal work), and between two pieces the dispatcher serves the other streams: with ONE launch of several rounds it
// keeps refilling every SIMD from that launch, and the small kernels of other handles (refinement rounds,
// eigenvectors: a few waves each) wait until the whole batch has drained -- tens of milliseconds with several scans in
// flight (kernel trace of bench.py).  Costs one launch per round, changes no result.
template <bool ELL0, bool UNIFORM, int LPF>
static hipError_t launch_hex_lpf(const LaunchArgs& a) {
  constexpr size_t kFreqs = 64 / LPF;
  const size_t waves = (a.n + kFreqs - 1) / kFreqs;
  size_t per_launch = waves;
  // only for kernels that run one wave per SIMD (Colloc8, Magnus8: a wave owns the SIMD's register file, nothing else
  // fits beside it, and all waves of a round end together); with two or more waves per SIMD the pieces' ragged ends
  // cost ~10 % and the other streams' kernels find room anyway
  if (a.hex_rounds_per_launch > 0 && hex_min_waves(STORM_SCHEME) == 1 && hex_wave_slots<ELL0, UNIFORM, LPF>() <= 1024)
    per_launch = hex_wave_slots<ELL0, UNIFORM, LPF>() * size_t(a.hex_rounds_per_launch);
  for (size_t w0 = 0; w0 < waves; w0 += per_launch) {
    const size_t off = w0 * kFreqs;
    const size_t n = std::min(a.n - off, per_launch * kFreqs);
    const unsigned blocks = (unsigned)((n + kFreqs - 1) / kFreqs);
    // a single solver's system has the interval-major copy of its planes, a sweep's systems have not
    if (hex_reads_aos(STORM_SCHEME) && a.table == nullptr)
      det_hex_kernel<STORM_SCHEME, ELL0, UNIFORM, LPF, true><<<blocks, 64, 0, a.stream>>>(
          a.sys, a.freq + off, n, a.out + off, a.flags, nullptr, nullptr);
    else
      det_hex_kernel<STORM_SCHEME, ELL0, UNIFORM, LPF><<<blocks, 64, 0, a.stream>>>(
          a.sys, a.freq + off, n, a.out + off, a.flags, a.table, a.sys_index ? a.sys_index + off : nullptr);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
template <bool ELL0, bool UNIFORM>
static size_t col_wave_slots() {
  constexpr int kMaxDev = 64;
  static std::atomic<size_t> cache[kMaxDev];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return size_t(1024);
  const bool cached = dev >= 0 && dev < kMaxDev;
  if (cached) { const size_t v = cache[dev].load(std::memory_order_relaxed); if (v) return v; }
  int cus = 0, per_cu = 0;
  size_t slots = 1024;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, det_mix_kernel<STORM_SCHEME, ELL0, UNIFORM>, 64, 0) == hipSuccess)
    slots = size_t(cus > 0 ? cus : 256) * size_t(per_cu > 0 ? per_cu : 4);
  if (cached) cache[dev].store(slots, std::memory_order_relaxed);
  return slots;
}
// Does det_mix take this batch?  It pays where the column-owner body keeps det_hex's occupancy: measured at 10^5
// frequencies on tams (scripts/time_scheme.py, det_mix vs det_hex alone, parity | FMA build, ms): Colloc2 10.6 vs 11.4 |
// 5.9 vs 6.9, Colloc4 23.5 vs 24.3 | 12.6 vs 13.3, Magnus2/4/6 2-3 % | 5-7 % faster, Magnus8 72.9 vs 78.1 | 48.3 vs 52.8;
// NOT for Colloc6 (58.1 vs 43.8 | 33.8 vs 25.1: the mixed kernel needs more than 256 registers and loses the second
// wave per SIMD) and Colloc8 (119.9 vs 108.2 | 64.3 vs 64.4).
// Column-owner workgroups det_mix gives this batch (0: det_mix does not take it).
template <bool ELL0, bool UNIFORM>
static size_t mix_col_blocks(const LaunchArgs& a) {
  if (STORM_SCHEME == kColloc6 || STORM_SCHEME == kColloc8) return 0;
  if (a.table != nullptr || a.sys.n_int < 4 || a.n < 32) return 0;
  if (a.hex_lanes < 0) return a.n * size_t(-a.hex_lanes) / 100 / 16;  // tuning knob: -p = p % column owners, whatever n
  if (a.hex_lanes != 0) return 0;
  // worth it once the column-owner part alone fills the chip and leaves a det_hex part behind it.  (Measured at 10^5
  // frequencies, FMA build, Colloc4: 80 % column-owner 19.75 ms, whole wave rounds of them (49 %) 20.67 ms, none 20.92 ms.)
  // From 30 wave slots' worth of frequencies (~92 000): 80 % of them.  From 18 (~55 000): as many column-owner waves as
  // are resident at once, less 5 % -- a few waves past one resident set start when the first finish and double the launch
  // (61 500 frequencies, FMA build Colloc4: 3075 waves 12.2 ms, 2690 waves 11.2 ms; det_hex alone 12.2 ms).  Measured
  // with scripts/mix_split_timing.py (share of column owners against time): 56 000 frequencies 11.2 -> 9.6 ms, 68 000
  // 12.9 -> 11.7, 80 000 15.1 -> 13.8, 90 000 16.4 -> 15.6.  Below that the long waves outlast the det_hex part.
  // The 18-slot rule only for kernels with three or more waves per SIMD: with two (parity Colloc4, Magnus2/4/6) it gains
  // 1-3 % at 50-70 000 frequencies and loses 4-7 % at 40 000 (scripts/mix_auto_check.py).  Known soft spot of the 80 %
  // rule: 86 000 frequencies of a two-wave kernel are 2.1 resident sets of column-owner waves and come out 2-4 % behind
  // det_hex (10^5: 2-3 % ahead); cutting the count back to whole sets did not change that, so it is left alone.
  const size_t slots = col_wave_slots<ELL0, UNIFORM>();
  const size_t want = std::min(a.n * 4 / 5 / 16, (a.n - 1) / 16);
  if (a.n >= 30 * slots) return want;
  if (slots >= 3 * 1024 && a.n >= 18 * slots) return std::min(want, slots - slots / 20);
  return 0;
}
template <bool ELL0, bool UNIFORM>
static bool mix_takes(const LaunchArgs& a) { return mix_col_blocks<ELL0, UNIFORM>(a) > 0; }
template <bool ELL0, bool UNIFORM>
static hipError_t launch_hex_v(const LaunchArgs& a) {
  int lpf = a.hex_lanes;
  // Large batches: the first ~80 % of the frequencies go to column-owner workgroups (4 lanes per frequency: a third of
  // det_hex's instructions for the ordered product), the rest to 16-lane det_hex workgroups of the SAME launch, which
  // start as the long column-owner waves finish and give the launch a short-wave end (det_mix_kernel).  Not for sweeps (a
  // wave of 16 frequencies may straddle systems) and not when the caller pinned the lane count to 16 or 32; hex_lanes = 4
  // forces the column-owner kernel alone.
  if ((lpf <= 0 || lpf == 4) && a.table == nullptr && a.sys.n_int >= 4) {
    if (lpf == 4) {
      det_col_kernel<STORM_SCHEME, ELL0, UNIFORM><<<(unsigned)((a.n + 15) / 16), 64, 0, a.stream>>>(a.sys, a.freq, a.n, a.out, a.flags);
      return hipGetLastError();
    }
    const size_t col_blocks = mix_col_blocks<ELL0, UNIFORM>(a);
    if (col_blocks > 0) {
      const size_t hex_blocks = (a.n - col_blocks * 16 + 3) / 4;
      det_mix_kernel<STORM_SCHEME, ELL0, UNIFORM><<<(unsigned)(col_blocks + hex_blocks), 64, 0, a.stream>>>(
          a.sys, a.freq, a.n, a.out, a.flags, (unsigned)col_blocks);
      return hipGetLastError();
    }
  }
  if (lpf != 16 && lpf != 32) {
    lpf = hex_cost<ELL0, UNIFORM, 32>(a.n, a.sys.n_int) < hex_cost<ELL0, UNIFORM, 16>(a.n, a.sys.n_int) ? 32 : 16;
    // The whole-rounds model holds where waves run in lockstep: one wave per SIMD at any size, more waves per SIMD only
    // while the 16-lane waves are less than 1.5 resident sets.  Beyond that the dispatcher refills SIMDs continuously
    // and 32 lanes only cost their longer ordered product (2*10^4 frequencies, tams: Colloc4 parity 5.13 ms with 16
    // lanes, 6.20 with 32; Magnus2/4/6 alike in both builds; FMA Colloc4 at 29 000: 3.90 vs 4.59).
    const size_t slots16 = hex_wave_slots<ELL0, UNIFORM, 16>();
    if (lpf == 32 && slots16 > 1024 && (a.n + 3) / 4 * 2 >= slots16 * 3) lpf = 16;
  }
  // (Tried and measured no gain, round 2: sending the whole wave rounds at 16 lanes and the remainder as a second launch
  // at 32 lanes.  10^5 frequencies are 25000 waves on 2048 slots = 12.2 rounds; the launch takes 12.2 rounds' time, not
  // 13 -- waves drift apart and the dispatcher refills SIMDs continuously -- so the split only added a launch.)
  return lpf == 32 ? launch_hex_lpf<ELL0, UNIFORM, 32>(a) : launch_hex_lpf<ELL0, UNIFORM, 16>(a);
}
template <bool ELL0, bool UNIFORM>
static hipError_t launch_build_v(const LaunchArgs& a) {
  const dim3 grid((unsigned)((a.sys.n_int + kBuildBlock - 1) / kBuildBlock), (unsigned)a.n, 1);
  build_steps_kernel<STORM_SCHEME, ELL0, UNIFORM><<<grid, kBuildBlock, 0, a.stream>>>(a.sys, a.freq, a.n, a.out, a.flags, a.table, a.sys_index);
  return hipGetLastError();
}

template <bool ELL0, bool UNIFORM>
static hipError_t launch_seg_product_v(const LaunchArgs& a) {
  const unsigned blocks = (unsigned)(a.n * (size_t)a.n_seg / 4);
  seg_product_kernel<STORM_SCHEME, ELL0, UNIFORM><<<blocks, 64, 0, a.stream>>>(a.sys, a.freq, a.n_seg, a.seg_len, a.out, a.growth,
                                                                             a.flags, a.table, a.sys_index);
  return hipGetLastError();
}
template <bool ELL0, bool UNIFORM>
static hipError_t launch_seg_fill_v(const LaunchArgs& a) {
  const unsigned blocks = (unsigned)(a.n * (size_t)a.n_seg / 4);
  seg_fill_kernel<STORM_SCHEME, ELL0, UNIFORM><<<blocks, 64, 0, a.stream>>>(a.sys, a.freq, a.n_seg, a.seg_len, a.nodes, a.vec,
                                                                          a.table, a.sys_index);
  return hipGetLastError();
}
static bool seg_args_ok(const LaunchArgs& a) {
  return a.sys.n_int > 0 && a.n_seg > 0 && a.n_seg % 4 == 0 && a.seg_len > 0 && a.seg_len % 16 == 0 &&
         (size_t)(a.n_seg - 1) * (size_t)a.seg_len + 1 >= a.sys.n_int && a.n * (size_t)a.n_seg / 4 < (size_t(1) << 31);
}
hipError_t STORM_CAT(launch_seg_product_scheme_, STORM_SCHEME)(const LaunchArgs& a) {
  if (a.n == 0) return hipSuccess;
  if (!seg_args_ok(a)) return hipErrorInvalidValue;
  if (a.ell0) return launch_seg_product_v<true, true>(a);
  return a.uniform ? launch_seg_product_v<false, true>(a) : launch_seg_product_v<false, false>(a);
}
hipError_t STORM_CAT(launch_seg_fill_scheme_, STORM_SCHEME)(const LaunchArgs& a) {
  if (a.n == 0) return hipSuccess;
  if (!seg_args_ok(a) || !a.nodes || !a.vec) return hipErrorInvalidValue;
  if (a.ell0) return launch_seg_fill_v<true, true>(a);
  return a.uniform ? launch_seg_fill_v<false, true>(a) : launch_seg_fill_v<false, false>(a);
}

// Frequencies that fill the device exactly once with 16-lane det_hex waves (wave slots x 4), and whether the launcher
// would send this batch through det_mix; the host side cuts FMA-build batches into whole wave rounds with these.
hipError_t STORM_CAT(launch_hexinfo_scheme_, STORM_SCHEME)(const LaunchArgs& a) {
  size_t* out = reinterpret_cast<size_t*>(a.out);  // [0]: frequencies per wave round, [1]: 1 if det_mix takes the batch
  size_t slots16;
  bool mix;
  if (a.ell0) { slots16 = hex_wave_slots<true, true, 16>(); mix = mix_takes<true, true>(a); }
  else if (a.uniform) { slots16 = hex_wave_slots<false, true, 16>(); mix = mix_takes<false, true>(a); }
  else { slots16 = hex_wave_slots<false, false, 16>(); mix = mix_takes<false, false>(a); }
  out[0] = slots16 * 4;
  out[1] = mix ? 1 : 0;
  return hipSuccess;
}

hipError_t STORM_CAT(launch_fused_scheme_, STORM_SCHEME)(const LaunchArgs& a) {
  if (a.n == 0 ) return hipSuccess;
  if (a.ell0) return launch_fused_v<true, true>(a);
  return a.uniform ? launch_fused_v<false, true>(a) : launch_fused_v<false, false>(a);
}
hipError_t STORM_CAT(launch_quad_scheme_, STORM_SCHEME)(const LaunchArgs& a) {
  if (a.n == 0) return hipSuccess;
  if (a.ell0) return launch_quad_v<true, true>(a);
  return a.uniform ? launch_quad_v<false, true>(a) : launch_quad_v<false, false>(a);
}
hipError_t STORM_CAT(launch_hex_scheme_, STORM_SCHEME)(const LaunchArgs& a) {
  if (a.n == 0) return hipSuccess;
  if (a.ell0) return launch_hex_v<true, true>(a);
  return a.uniform ? launch_hex_v<false, true>(a) : launch_hex_v<false, false>(a);
}
hipError_t STORM_CAT(launch_build_scheme_, STORM_SCHEME)(const LaunchArgs& a) {
  if (a.n == 0 || a.sys.n_int == 0) return hipSuccess;
  if (a.ell0) return launch_build_v<true, true>(a);
  return a.uniform ? launch_build_v<false, true>(a) : launch_build_v<false, false>(a);
}

}  // namespace STORM_NS
