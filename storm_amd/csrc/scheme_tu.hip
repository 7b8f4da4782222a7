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
template <bool ELL0, bool UNIFORM>
static hipError_t launch_hex_v(const LaunchArgs& a) {
  const unsigned blocks = (unsigned)((a.n + 3) / 4);
  det_hex_kernel<STORM_SCHEME, ELL0, UNIFORM><<<blocks, 64, 0, a.stream>>>(a.sys, a.freq, a.n, a.out, a.flags);
  return hipGetLastError();
}
template <bool ELL0, bool UNIFORM>
static hipError_t launch_build_v(const LaunchArgs& a) {
  const dim3 grid((unsigned)((a.sys.n_int + 255) / 256), (unsigned)a.n, 1);
  build_steps_kernel<STORM_SCHEME, ELL0, UNIFORM><<<grid, 256, 0, a.stream>>>(a.sys, a.freq, a.n, a.out, a.flags);
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
