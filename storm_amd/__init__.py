"""storm_amd -- MI355X-native frequency-scan hot path of StORM behind a C ABI.

The product is the shared library storm_amd/lib/libstorm_hip.so (hand-written HIP for
gfx950, see storm_amd/csrc and include/storm_hip.h).  This package is only the thin
ctypes binding used by the tests and the benchmark; there is no CPU fallback.
"""
from .ffi import (HipSolver, HipMulti, HipPostproc, HipSweep, StormError, Bracket, SCHEMES, POST_FIELDS, MODEL_FIELDS,  # noqa: F401
                  METRIC_FIELDS, COUPLING_MATRICES,
                  lib_path, load_library, relative_to_ulp)
