// scheme_tu.hip -- instantiates the step-matrix kernels for ONE difference scheme.
// Compiled 8 times: hipcc -DSTORM_SCHEME=<0..7> ...
#include "scheme_kernels.hpp"

#ifndef STORM_SCHEME
#error "compile with -DSTORM_SCHEME=<0..7>"
#endif

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
  static const size_t slots = [] {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, det_hex_kernel<STORM_SCHEME, ELL0, UNIFORM, LPF>, 64, 0) != hipSuccess)
      return size_t(1024);
    return size_t(cus > 0 ? cus : 256) * size_t(per_cu > 0 ? per_cu : 4);
  }();
  return slots;
}
template <bool ELL0, bool UNIFORM, int LPF>
static double hex_cost(size_t n, size_t n_int) {
  constexpr double kStepFlops[8] = {610., 1396., 2694., 5048., 1486., 1848., 2562., 3900.};
  const size_t waves = (n * LPF + 63) / 64, slots = hex_wave_slots<ELL0, UNIFORM, LPF>();
  return double((waves + slots - 1) / slots) * double((n_int + LPF - 1) / LPF) * (kStepFlops[STORM_SCHEME] + 20.0 * LPF);
}
template <bool ELL0, bool UNIFORM, int LPF>
static hipError_t launch_hex_lpf(const LaunchArgs& a) {
  const unsigned blocks = (unsigned)((a.n * LPF + 63) / 64);
  det_hex_kernel<STORM_SCHEME, ELL0, UNIFORM, LPF><<<blocks, 64, 0, a.stream>>>(a.sys, a.freq, a.n, a.out, a.flags, a.table, a.sys_index);
  return hipGetLastError();
}
template <bool ELL0, bool UNIFORM>
static hipError_t launch_hex_v(const LaunchArgs& a) {
  int lpf = a.hex_lanes;
  if (lpf != 16 && lpf != 32)
    lpf = hex_cost<ELL0, UNIFORM, 32>(a.n, a.sys.n_int) < hex_cost<ELL0, UNIFORM, 16>(a.n, a.sys.n_int) ? 32 : 16;
  return lpf == 32 ? launch_hex_lpf<ELL0, UNIFORM, 32>(a) : launch_hex_lpf<ELL0, UNIFORM, 16>(a);
}
template <bool ELL0, bool UNIFORM>
static hipError_t launch_build_v(const LaunchArgs& a) {
  const dim3 grid((unsigned)((a.sys.n_int + 255) / 256), (unsigned)a.n, 1);
  build_steps_kernel<STORM_SCHEME, ELL0, UNIFORM><<<grid, 256, 0, a.stream>>>(a.sys, a.freq, a.n, a.out, a.flags, a.table, a.sys_index);
  return hipGetLastError();
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
