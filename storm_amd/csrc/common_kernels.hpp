#pragma once
#include "types.hpp"

// Scheme-independent launchers, one set per arithmetic build (see types.hpp).
#define STORM_DECL_COMMON(ns)                                                                                     \
  namespace ns {                                                                                                  \
  hipError_t launch_chain(const storm::SystemDev& s, bool ell0, const double* freq, size_t n, const double* steps, \
                          double* det_out, hipStream_t stream);                                                   \
  bool eigen_uses_lanes(size_t n, int layout);                                                                    \
  hipError_t launch_eigen(const storm::SystemDev& s, int ell_class, const double* freq, size_t n, const double* steps, \
                          double* upper, int* swaps, double* det_out, double* vec, hipStream_t stream,            \
                          const storm::SystemDev* table, const int* sys_index, int layout);                       \
  }
STORM_DECL_COMMON(storm)
STORM_DECL_COMMON(storm_fast)
