#pragma once
#include "device_system.hpp"

namespace storm {
hipError_t launch_chain(const SystemDev& s, bool ell0, const double* freq, size_t n, const double* steps,
                        double* det_out, hipStream_t stream);
hipError_t launch_eigen(const SystemDev& s, bool ell0, const double* freq, size_t n, const double* steps,
                        double* upper, int* swaps, double* det_out, double* vec, hipStream_t stream);
}  // namespace storm
