"""ctypes binding of include/storm_hip.h (libstorm_hip.so)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SCHEMES = ["colloc2", "colloc4", "colloc6", "colloc8", "magnus2", "magnus4", "magnus6", "magnus8"]
_dp = C.POINTER(C.c_double)


def lib_path():
    """libstorm_hip.so in-tree; STORM_HIP_LIBRARY names another BUILD of the same library (the variant builds of
    tests/attribution_config3.py) -- never anything but a HIP build: there is no CPU fallback."""
    return os.environ.get("STORM_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libstorm_hip.so")


class StormError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"storm_hip error {code}: {msg}")
        self.code = code


class Bracket(C.Structure):
    _fields_ = [("lo_x", C.c_double), ("lo_f", C.c_double), ("hi_x", C.c_double), ("hi_f", C.c_double),
                ("root", C.c_double), ("evals", C.c_uint64)]


BRACKET_DTYPE = np.dtype([("lo_x", "<f8"), ("lo_f", "<f8"), ("hi_x", "<f8"), ("hi_f", "<f8"), ("root", "<f8"), ("evals", "<u8")])
assert BRACKET_DTYPE.itemsize == C.sizeof(Bracket)


class SystemDesc(C.Structure):
    _fields_ = [("scheme", C.c_int32), ("ell", C.c_uint64), ("m", C.c_int64), ("n_int", C.c_size_t),
                ("p", C.c_size_t), ("u", _dp), ("v_gamma", _dp), ("a_star", _dp), ("c1", _dp), ("rot", _dp),
                ("delta_over_x", _dp), ("inner_c1", C.c_double), ("inner_rot", C.c_double),
                ("outer_u", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("det_evals", C.c_uint64), ("refine_evals", C.c_uint64), ("refine_rounds", C.c_uint64),
                ("eigenvectors", C.c_uint64), ("launches", C.c_uint64), ("segment_fallbacks", C.c_uint64)]


class ModelDesc(C.Structure):
    _fields_ = [("n_pts", C.c_size_t)] + [(k, _dp) for k in ("r_coord", "m_coord", "rho", "p", "v", "u", "gamma1",
                                                               "a_star", "c1", "rot")]


class ModeId(C.Structure):
    _fields_ = [("cross_clockwise", C.c_uint64), ("cross_counter_clockwise", C.c_uint64),
                ("radial_order", C.c_int64), ("n_nodes", C.c_uint64)]


class MetricDesc(C.Structure):
    _fields_ = [(k, _dp) for k in ("alpha", "dalpha", "ddalpha", "beta", "dbeta", "ddbeta")]


METRIC_FIELDS = ["alpha", "dalpha", "ddalpha", "beta", "dbeta", "ddbeta"]
COUPLING_MATRICES = ["d_squared", "d_linear", "d_zero", "r_linear", "r_zero", "l_zero"]
MODEL_FIELDS = ["r_coord", "m_coord", "rho", "p", "v", "u", "gamma1", "a_star", "c1", "rot"]
POST_FIELDS = ["y1", "y2", "y3", "y4", "xi_r", "xi_h", "xi_tn", "xi_tp", "psi", "dpsi", "rho", "p", "chi"]

# every symbol include/storm_hip.h declares
EXPORTS = ["storm_solver_create", "storm_solver_destroy", "storm_det_batch", "storm_det_batch_device",
           "storm_scan_and_optimize", "storm_eigenvector_batch", "storm_precision_relative_to_ulp",
           "storm_solver_stats", "storm_solver_kernel_times", "storm_solver_set_option", "storm_solver_n_int", "storm_last_error",
           "storm_version", "storm_postproc_create", "storm_postproc_destroy", "storm_postproc_set_option", "storm_postprocess_batch",
           "storm_modes_batch", "storm_coupling_matrices", "storm_postproc_last_kernel_ms",
           "storm_memory_cache_trim", "storm_memory_cache_limit", "storm_memory_cache_stats",
           "storm_sweep_create", "storm_sweep_destroy", "storm_sweep_size", "storm_sweep_n_int", "storm_sweep_det",
           "storm_sweep_scan_and_optimize", "storm_sweep_eigenvectors", "storm_sweep_modes", "storm_sweep_stats",
           "storm_sweep_set_option", "storm_sweep_modes_resident", "storm_sweep_coupling_matrices",
           "storm_sweep_coupling_matrices_batch", "storm_scan_brackets", "storm_optimize_brackets", "storm_det_batch_device_noalloc", "storm_det_scaled_batch", "storm_multi_create", "storm_multi_destroy",
           "storm_multi_devices", "storm_multi_set_option", "storm_multi_stats", "storm_multi_scan_and_optimize",
           "storm_multi_eigenvector_batch", "storm_shard_bounds"]

_lib = None


def load_library():
    """Load libstorm_hip.so; raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `make -C storm_amd/csrc` "
                          "(or python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    L = C.CDLL(path)
    L.storm_last_error.restype = C.c_char_p
    L.storm_version.restype = C.c_char_p
    L.storm_solver_create.argtypes = [C.POINTER(SystemDesc), C.POINTER(C.c_void_p)]
    L.storm_solver_destroy.argtypes = [C.c_void_p]
    L.storm_solver_destroy.restype = None
    L.storm_det_batch.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp]
    L.storm_det_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.storm_det_scaled_batch.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, C.POINTER(C.c_int64)]
    L.storm_det_batch_device_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.storm_scan_and_optimize.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_uint64, C.POINTER(Bracket), C.c_size_t,
                                          C.POINTER(C.c_size_t), C.POINTER(C.c_int), _dp, C.POINTER(C.c_uint64)]
    L.storm_eigenvector_batch.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, _dp]
    L.storm_scan_brackets.argtypes = [C.c_void_p, _dp, C.c_size_t, C.POINTER(Bracket), C.c_size_t, C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_int), _dp, C.POINTER(C.c_uint64)]
    L.storm_optimize_brackets.argtypes = [C.c_void_p, C.POINTER(Bracket), C.c_size_t, C.c_uint64]
    L.storm_precision_relative_to_ulp.restype = C.c_uint64
    L.storm_precision_relative_to_ulp.argtypes = [C.c_double]
    L.storm_solver_stats.argtypes = [C.c_void_p, C.POINTER(Stats), C.c_int]
    L.storm_solver_kernel_times.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_uint64), C.c_int]
    L.storm_solver_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.storm_solver_n_int.restype = C.c_size_t
    L.storm_solver_n_int.argtypes = [C.c_void_p]
    L.storm_postproc_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]
    L.storm_postproc_destroy.argtypes = [C.c_void_p]
    L.storm_postproc_destroy.restype = None
    L.storm_postproc_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.storm_postprocess_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, _dp, C.c_size_t, _dp, _dp,
                                          C.POINTER(ModeId), _dp, C.c_size_t]
    L.storm_modes_batch.argtypes = [C.c_void_p, C.c_void_p, _dp, C.c_size_t, _dp, _dp, _dp, C.POINTER(ModeId), _dp,
                                    C.c_size_t]
    L.storm_coupling_matrices.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), _dp, _dp, C.c_int64,
                                          C.POINTER(MetricDesc), _dp]
    L.storm_memory_cache_limit.argtypes = [C.c_size_t, C.c_size_t]
    L.storm_memory_cache_stats.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.storm_memory_cache_stats.restype = None
    L.storm_postproc_last_kernel_ms.restype = C.c_double
    L.storm_postproc_last_kernel_ms.argtypes = [C.c_void_p]
    L.storm_multi_create.argtypes = [C.POINTER(SystemDesc), C.POINTER(C.c_int), C.c_size_t, C.POINTER(C.c_void_p)]
    L.storm_multi_destroy.argtypes = [C.c_void_p]
    L.storm_multi_destroy.restype = None
    L.storm_multi_devices.restype = C.c_size_t
    L.storm_multi_devices.argtypes = [C.c_void_p]
    L.storm_multi_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.storm_multi_stats.argtypes = [C.c_void_p, C.POINTER(Stats), C.c_int]
    L.storm_multi_scan_and_optimize.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_uint64, C.POINTER(Bracket), C.c_size_t,
                                                C.POINTER(C.c_size_t), C.POINTER(C.c_int), _dp, C.POINTER(C.c_uint64)]
    L.storm_multi_eigenvector_batch.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, _dp]
    L.storm_shard_bounds.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.storm_shard_bounds.restype = None
    _sp = C.POINTER(C.c_size_t)
    L.storm_sweep_create.argtypes = [C.POINTER(SystemDesc), C.c_size_t, C.POINTER(C.c_void_p)]
    L.storm_sweep_destroy.argtypes = [C.c_void_p]
    L.storm_sweep_destroy.restype = None
    L.storm_sweep_size.restype = C.c_size_t
    L.storm_sweep_size.argtypes = [C.c_void_p]
    L.storm_sweep_n_int.restype = C.c_size_t
    L.storm_sweep_n_int.argtypes = [C.c_void_p]
    L.storm_sweep_det.argtypes = [C.c_void_p, _sp, _dp, _dp]
    L.storm_sweep_scan_and_optimize.argtypes = [C.c_void_p, _sp, _dp, C.c_uint64, C.POINTER(Bracket), C.c_size_t, _sp,
                                                C.POINTER(C.c_int), _dp, C.POINTER(C.c_uint64)]
    L.storm_sweep_eigenvectors.argtypes = [C.c_void_p, _sp, _dp, _dp, _dp]
    L.storm_sweep_modes.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), _sp, _dp, _dp, _dp, _dp, C.POINTER(ModeId), _dp,
                                    C.c_size_t]
    L.storm_sweep_modes_resident.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), _sp, _dp, C.POINTER(ModeId), _dp, C.c_size_t]
    L.storm_sweep_coupling_matrices.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, _sp, C.POINTER(C.c_uint64), _dp, C.c_int64,
                                                C.POINTER(MetricDesc), _dp]
    L.storm_sweep_stats.argtypes = [C.c_void_p, C.POINTER(Stats), C.c_int]
    L.storm_sweep_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    _lib = L
    return L


KERNEL_KINDS = ["det_fused", "build_steps", "chain", "eigen", "det_quad", "det_hex", "postprocess", "segments"]


def kernel_times(handle, reset=False):
    """{kind: (total_ms, launches)} from HIP events (option "profile" must be on)."""
    ms = (C.c_double * len(KERNEL_KINDS))()
    cnt = (C.c_uint64 * len(KERNEL_KINDS))()
    _check(load_library().storm_solver_kernel_times(C.c_void_p(handle), ms, cnt, 1 if reset else 0))
    return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(KERNEL_KINDS)}


STORM_ERR_CAPACITY = -5   # include/storm_hip.h: output buffer too small; *n_out holds the required count


def _check(rc):
    if rc != 0:
        raise StormError(rc, load_library().storm_last_error().decode())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def memory_cache_stats():
    """(device bytes, pinned host bytes) currently held by the library's buffer cache."""
    d, h = C.c_size_t(0), C.c_size_t(0)
    load_library().storm_memory_cache_stats(C.byref(d), C.byref(h))
    return int(d.value), int(h.value)


def memory_cache_trim():
    _check(load_library().storm_memory_cache_trim())


def memory_cache_limit(device_bytes, pinned_host_bytes):
    _check(load_library().storm_memory_cache_limit(device_bytes, pinned_host_bytes))


def relative_to_ulp(rel):
    return int(load_library().storm_precision_relative_to_ulp(rel))


class HipSolver:
    """One storm_solver handle.  `system` is the dict of SoA arrays + boundary scalars described by
    storm_system_desc (what DiscretizedSystemImpl::new knows after interpolating the model)."""

    def __init__(self, scheme, ell, m, system, n_int, p):
        L = load_library()
        if isinstance(scheme, str):
            scheme = SCHEMES.index(scheme)
        arrs = {k: _f64(system[k]) for k in ("u", "v_gamma", "a_star", "c1", "rot", "delta_over_x")}
        for k, a in arrs.items():
            if a.size != n_int * p:
                raise ValueError(f"{k}: expected {n_int * p} entries, got {a.size}")
        d = SystemDesc(scheme, ell, m, n_int, p, *[_p(arrs[k]) for k in ("u", "v_gamma", "a_star", "c1", "rot",
                                                                          "delta_over_x")],
                       float(system["inner_c1"]), float(system["inner_rot"]), float(system["outer_u"]))
        h = C.c_void_p()
        _check(L.storm_solver_create(C.byref(d), C.byref(h)))
        self._h = h
        self.n_int, self.p, self.scheme = n_int, p, scheme

    @classmethod
    def borrow(cls, raw_handle, owner=None):
        """A view of a storm_solver* that someone else owns (e.g. host.ErasedSolver.raw_handle()); `owner` is kept alive."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(raw_handle)
        self._borrowed, self._owner = True, owner
        self.n_int = int(load_library().storm_solver_n_int(self._h))
        self.p = self.scheme = None
        return self

    def close(self):
        if getattr(self, "_h", None) and not getattr(self, "_borrowed", False):
            load_library().storm_solver_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        _check(load_library().storm_solver_set_option(self._h, key.encode(), int(value)))

    def stats(self, reset=False):
        s = Stats()
        _check(load_library().storm_solver_stats(self._h, C.byref(s), 1 if reset else 0))
        return {k: int(getattr(s, k)) for k, _ in Stats._fields_}

    def kernel_times(self, reset=False):
        return kernel_times(self._h.value, reset)

    def det(self, freqs):
        f = _f64(np.atleast_1d(freqs))
        out = np.empty(f.size)
        _check(load_library().storm_det_batch(self._h, _p(f), f.size, _p(out)))
        return out

    def det_device(self, d_freq_ptr, n, d_out_ptr, stream=None):
        _check(load_library().storm_det_batch_device(self._h, C.c_void_p(d_freq_ptr), n, C.c_void_p(d_out_ptr),
                                                    C.c_void_p(stream) if stream else None))

    def det_scaled(self, freqs):
        """storm_det_scaled_batch -> (mantissa in [0.5, 1), exponent): det = mantissa * 2**exponent without overflow."""
        f = _f64(np.atleast_1d(freqs))
        m = np.empty(f.size)
        e = np.zeros(f.size, dtype=np.int64)
        _check(load_library().storm_det_scaled_batch(self._h, _p(f), f.size, _p(m), e.ctypes.data_as(C.POINTER(C.c_int64))))
        return m, e

    def det_device_noalloc(self, d_freq_ptr, n, d_out_ptr, stream=None):
        _check(load_library().storm_det_batch_device_noalloc(self._h, C.c_void_p(d_freq_ptr), n, C.c_void_p(d_out_ptr),
                                                            C.c_void_p(stream) if stream else None))

    def scan_and_optimize(self, freqs, ulp, cap=None, want_dets=True):
        """storm_scan_and_optimize -> (brackets, dets, invalid); want_dets=False: the reference's return value only (the
        discriminants of the grid stay on the device: dets is None)."""
        f = _f64(freqs)
        cap = max(1, f.size) if cap is None else cap
        out = (Bracket * max(cap, 1))()
        idx = (C.c_uint64 * max(cap, 1))()
        dets = np.empty(f.size) if want_dets else None
        n_out = C.c_size_t(0)
        inv = C.c_int(0)
        _check(load_library().storm_scan_and_optimize(self._h, _p(f), f.size, ulp, out, cap, C.byref(n_out),
                                                      C.byref(inv), _p(dets) if want_dets else None, idx))
        res = [dict(lo_x=b.lo_x, lo_f=b.lo_f, hi_x=b.hi_x, hi_f=b.hi_f, root=b.root, evals=int(b.evals),
                    index=int(idx[i])) for i, b in enumerate(out[:n_out.value])]
        return res, dets, bool(inv.value)

    def scan_brackets(self, freqs, cap=None, want_dets=True):
        """storm_scan_brackets: the unrefined sign-swap brackets (root = NaN, evals = 0) -> (brackets, dets, invalid)."""
        f = _f64(freqs)
        cap = max(1, f.size) if cap is None else cap
        out = (Bracket * max(cap, 1))()
        idx = (C.c_uint64 * max(cap, 1))()
        dets = np.empty(f.size) if want_dets else None
        n_out, inv = C.c_size_t(0), C.c_int(0)
        _check(load_library().storm_scan_brackets(self._h, _p(f), f.size, out, cap, C.byref(n_out), C.byref(inv),
                                                  _p(dets) if want_dets else None, idx))
        res = [dict(lo_x=b.lo_x, lo_f=b.lo_f, hi_x=b.hi_x, hi_f=b.hi_f, root=b.root, evals=int(b.evals),
                    index=int(idx[i])) for i, b in enumerate(out[:n_out.value])]
        return res, dets, bool(inv.value)

    # ---- the same two calls on record arrays (storm_amd.distributed: rows of RECORD float64 columns lo_x, lo_f, hi_x, hi_f,
    # root, evals, global index), packed and unpacked through numpy views of the C struct array: no per-bracket Python
    def scan_and_optimize_records(self, freqs, ulp, offset=0, cap=None):
        """storm_scan_and_optimize -> (refined records [n, 7], invalid); `offset` is added to the bracket indices (a slice's
        position in the whole grid).  The discriminants of the grid stay on the device."""
        f = _f64(freqs)
        cap = min(max(1, f.size), 8192) if cap is None else cap
        while True:
            out = (Bracket * max(cap, 1))()
            idx = (C.c_uint64 * max(cap, 1))()
            n_out, inv = C.c_size_t(0), C.c_int(0)
            rc = load_library().storm_scan_and_optimize(self._h, _p(f), f.size, ulp, out, cap, C.byref(n_out), C.byref(inv), None, idx)
            if rc == STORM_ERR_CAPACITY and cap < max(n_out.value, 1):   # (more modes than room: once more, with the count it reported)
                cap = n_out.value
                continue
            _check(rc)
            break
        n = n_out.value
        rec = np.empty((n, 7))
        b = np.frombuffer(out, dtype=BRACKET_DTYPE, count=n)
        for k, name in enumerate(("lo_x", "lo_f", "hi_x", "hi_f", "root", "evals")):
            rec[:, k] = b[name]
        rec[:, 6] = np.frombuffer(idx, dtype=np.uint64, count=n).astype(np.float64) + float(offset)
        return rec, bool(inv.value)

    def scan_brackets_records(self, freqs, offset=0, cap=None):
        """storm_scan_brackets -> (records [n, 7], invalid); `offset` is added to the bracket indices (a slice's position in
        the whole grid).  The discriminants of the grid stay on the device."""
        f = _f64(freqs)
        cap = max(1, f.size) if cap is None else cap
        out = (Bracket * max(cap, 1))()
        idx = (C.c_uint64 * max(cap, 1))()
        n_out, inv = C.c_size_t(0), C.c_int(0)
        _check(load_library().storm_scan_brackets(self._h, _p(f), f.size, out, cap, C.byref(n_out), C.byref(inv), None, idx))
        n = n_out.value
        rec = np.empty((n, 7))
        b = np.frombuffer(out, dtype=BRACKET_DTYPE, count=n)
        for k, name in enumerate(("lo_x", "lo_f", "hi_x", "hi_f", "root", "evals")):
            rec[:, k] = b[name]
        rec[:, 6] = np.frombuffer(idx, dtype=np.uint64, count=n) + np.uint64(offset)
        return rec, bool(inv.value)

    def optimize_records(self, rec, ulp):
        """storm_optimize_brackets on records [n, 7] (columns 0..3 in; 0..5 out, column 6 kept) -> new array"""
        rec = np.ascontiguousarray(rec, dtype=np.float64)
        n = rec.shape[0]
        arr = (Bracket * max(n, 1))()
        b = np.frombuffer(arr, dtype=BRACKET_DTYPE, count=n)
        for k, name in enumerate(("lo_x", "lo_f", "hi_x", "hi_f")):
            b[name] = rec[:, k]
        b["root"] = np.nan
        b["evals"] = 0
        _check(load_library().storm_optimize_brackets(self._h, arr, n, ulp))
        out = rec.copy()
        for k, name in enumerate(("lo_x", "lo_f", "hi_x", "hi_f", "root", "evals")):
            out[:, k] = b[name]
        return out

    def optimize_brackets(self, brackets, ulp):
        """storm_optimize_brackets on a list of dicts with lo_x, lo_f, hi_x, hi_f (other keys are kept)."""
        n = len(brackets)
        arr = (Bracket * max(n, 1))()
        for k, b in enumerate(brackets):
            arr[k] = Bracket(b["lo_x"], b["lo_f"], b["hi_x"], b["hi_f"], float("nan"), 0)
        _check(load_library().storm_optimize_brackets(self._h, arr, n, ulp))
        out = []
        for k, b in enumerate(brackets):
            d = dict(b)
            d.update(lo_x=arr[k].lo_x, lo_f=arr[k].lo_f, hi_x=arr[k].hi_x, hi_f=arr[k].hi_f, root=arr[k].root,
                     evals=int(arr[k].evals))
            out.append(d)
        return out

    def eigenvector(self, freqs, out=None):
        """storm_eigenvector_batch -> (dets, vec[n][(n_int + 1) * 4]); out: a float64 array of at least that many elements
        to write the eigenvectors into (a caller that scans repeatedly keeps one instead of touching fresh pages)."""
        f = _f64(np.atleast_1d(freqs))
        dets = np.empty(f.size)
        length = (self.n_int + 1) * 4
        if out is None:
            vec = np.empty((f.size, length))
        else:
            if out.dtype != np.float64 or not out.flags.c_contiguous or out.size < f.size * length:
                raise ValueError("out must be a C-contiguous float64 array of at least n * (n_int + 1) * 4 elements")
            vec = out.reshape(-1)[:f.size * length].reshape(f.size, length)
        _check(load_library().storm_eigenvector_batch(self._h, _p(f), f.size, _p(dets), _p(vec)))
        return dets, vec

    def modes(self, post, freqs, want_vec=False, want_fields=True, node_cap=None):
        """storm_modes_batch: eigenvectors + post-processing with the eigenvectors staying on the device."""
        f = _f64(np.atleast_1d(freqs))
        npts = post.n_pts
        cap = npts if node_cap is None else node_cap
        dets = np.empty(f.size)
        vec = np.empty((f.size, npts * 4)) if want_vec else None
        fields = np.empty((f.size, len(POST_FIELDS), npts)) if want_fields else None
        ids = (ModeId * max(f.size, 1))()
        nodes = np.full((f.size, max(cap, 1)), np.nan)
        _check(load_library().storm_modes_batch(self._h, post._h, _p(f), f.size, _p(dets), _p(vec) if want_vec else None,
                                                _p(fields) if want_fields else None, ids, _p(nodes) if cap else None, cap))
        out = _post_result(f.size, fields, ids, nodes, cap)
        out["det"] = dets
        out["vec"] = vec
        return out


class HipMulti:
    """One storm_multi handle: a scan over several devices of the node from this process (devices may repeat)."""

    def __init__(self, scheme, ell, m, system, n_int, p, devices):
        L = load_library()
        if isinstance(scheme, str):
            scheme = SCHEMES.index(scheme)
        arrs = {k: _f64(system[k]) for k in ("u", "v_gamma", "a_star", "c1", "rot", "delta_over_x")}
        d = SystemDesc(scheme, ell, m, n_int, p, *[_p(arrs[k]) for k in ("u", "v_gamma", "a_star", "c1", "rot", "delta_over_x")],
                       float(system["inner_c1"]), float(system["inner_rot"]), float(system["outer_u"]))
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        _check(L.storm_multi_create(C.byref(d), devs, len(devices), C.byref(h)))
        self._h = h
        self.n_int, self.p, self.scheme, self.n_dev = n_int, p, scheme, len(devices)
        self._owner = None

    @classmethod
    def borrow(cls, raw_handle, n_int, p, scheme, n_dev, owner=None):
        """A view of a storm_multi owned by someone else (storm_amd.host.MultiSolver)."""
        load_library()
        self = cls.__new__(cls)
        self._h = C.c_void_p(raw_handle)
        self.n_int, self.p, self.scheme, self.n_dev = n_int, p, scheme, n_dev
        self._owner = owner if owner is not None else True
        return self

    def close(self):
        if getattr(self, "_h", None) and getattr(self, "_owner", None) is None:
            load_library().storm_multi_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        _check(load_library().storm_multi_set_option(self._h, key.encode(), int(value)))

    def stats(self, reset=False):
        s = Stats()
        _check(load_library().storm_multi_stats(self._h, C.byref(s), 1 if reset else 0))
        return {k: int(getattr(s, k)) for k, _ in Stats._fields_}

    def scan_and_optimize(self, freqs, ulp, cap=None, want_dets=True):
        f = _f64(freqs)
        cap = max(1, f.size) if cap is None else cap
        out = (Bracket * max(cap, 1))()
        idx = (C.c_uint64 * max(cap, 1))()
        dets = np.empty(f.size) if want_dets else None
        n_out, inv = C.c_size_t(0), C.c_int(0)
        _check(load_library().storm_multi_scan_and_optimize(self._h, _p(f), f.size, ulp, out, cap, C.byref(n_out),
                                                            C.byref(inv), _p(dets) if want_dets else None, idx))
        res = [dict(lo_x=b.lo_x, lo_f=b.lo_f, hi_x=b.hi_x, hi_f=b.hi_f, root=b.root, evals=int(b.evals),
                    index=int(idx[i])) for i, b in enumerate(out[:n_out.value])]
        return res, dets, bool(inv.value)

    def eigenvector(self, freqs, out=None):
        f = _f64(np.atleast_1d(freqs))
        dets = np.empty(f.size)
        length = (self.n_int + 1) * 4
        vec = np.empty((f.size, length)) if out is None else out.reshape(-1)[:f.size * length].reshape(f.size, length)
        _check(load_library().storm_multi_eigenvector_batch(self._h, _p(f), f.size, _p(dets), _p(vec)))
        return dets, vec


def shard_bounds(n, n_parts, k):
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    load_library().storm_shard_bounds(n, n_parts, k, C.byref(lo), C.byref(hi))
    return int(lo.value), int(hi.value)


def _offsets(lists):
    off = np.zeros(len(lists) + 1, dtype=np.uintp)
    np.cumsum([len(x) for x in lists], out=off[1:])
    return off


def _sp(a):
    return a.ctypes.data_as(C.POINTER(C.c_size_t))


class HipSweep:
    """One storm_sweep handle: S systems (same scheme and grid; ell, m and coefficients free) evaluated in lock-step.
    `systems` is a list of (ell, m, system dict) with the dicts HipSolver takes.  Per-system inputs and outputs are
    Python lists with one entry per system."""

    def __init__(self, scheme, systems, n_int, p):
        L = load_library()
        if isinstance(scheme, str):
            scheme = SCHEMES.index(scheme)
        keys = ("u", "v_gamma", "a_star", "c1", "rot", "delta_over_x")
        descs = (SystemDesc * max(len(systems), 1))()
        keep = []
        for k, (ell, m, system) in enumerate(systems):
            arrs = {q: _f64(system[q]) for q in keys}
            for q, a in arrs.items():
                if a.size != n_int * p:
                    raise ValueError(f"system {k}, {q}: expected {n_int * p} entries, got {a.size}")
            keep.append(arrs)
            descs[k] = SystemDesc(scheme, ell, m, n_int, p, *[_p(arrs[q]) for q in keys], float(system["inner_c1"]),
                                  float(system["inner_rot"]), float(system["outer_u"]))
        h = C.c_void_p()
        _check(L.storm_sweep_create(descs, len(systems), C.byref(h)))
        self._h = h
        self.n_sys, self.n_int, self.p, self.scheme = len(systems), n_int, p, scheme

    def close(self):
        if getattr(self, "_h", None):
            load_library().storm_sweep_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        _check(load_library().storm_sweep_set_option(self._h, key.encode(), int(value)))

    def stats(self, reset=False):
        s = Stats()
        _check(load_library().storm_sweep_stats(self._h, C.byref(s), 1 if reset else 0))
        return {k: int(getattr(s, k)) for k, _ in Stats._fields_}

    def _flat(self, lists):
        if len(lists) != self.n_sys:
            raise ValueError(f"expected {self.n_sys} per-system lists, got {len(lists)}")
        lists = [_f64(np.atleast_1d(x)).ravel() for x in lists]
        off = _offsets(lists)
        flat = np.concatenate(lists) if lists else np.empty(0)
        return off, _f64(flat)

    def det(self, freq_lists):
        off, f = self._flat(freq_lists)
        out = np.empty(f.size)
        _check(load_library().storm_sweep_det(self._h, _sp(off), _p(f), _p(out)))
        return [out[off[u]:off[u + 1]] for u in range(self.n_sys)]

    def scan_and_optimize(self, freq_lists, ulp):
        """-> per system (brackets, dets, had_invalid), exactly what HipSolver.scan_and_optimize returns."""
        L = load_library()
        off, f = self._flat(freq_lists)
        cap = max(1, f.size)
        out = (Bracket * cap)()
        idx = (C.c_uint64 * cap)()
        dets = np.empty(f.size)
        ooff = np.zeros(self.n_sys + 1, dtype=np.uintp)
        inv = (C.c_int * max(self.n_sys, 1))()
        _check(L.storm_sweep_scan_and_optimize(self._h, _sp(off), _p(f), ulp, out, cap, _sp(ooff), inv, _p(dets), idx))
        res = []
        for u in range(self.n_sys):
            br = [dict(lo_x=b.lo_x, lo_f=b.lo_f, hi_x=b.hi_x, hi_f=b.hi_f, root=b.root, evals=int(b.evals),
                       index=int(idx[i])) for i, b in zip(range(int(ooff[u]), int(ooff[u + 1])), out[int(ooff[u]):int(ooff[u + 1])])]
            res.append((br, dets[off[u]:off[u + 1]], bool(inv[u])))
        return res

    def eigenvector(self, freq_lists):
        off, f = self._flat(freq_lists)
        dets = np.empty(f.size)
        vec = np.empty((f.size, (self.n_int + 1) * 4))
        _check(load_library().storm_sweep_eigenvectors(self._h, _sp(off), _p(f), _p(dets), _p(vec)))
        return [(dets[off[u]:off[u + 1]], vec[off[u]:off[u + 1]]) for u in range(self.n_sys)]

    def modes(self, posts, freq_lists, want_vec=False, want_fields=True, node_cap=None):
        """storm_sweep_modes; posts: one HipPostproc per system (may repeat).  -> per system the dict HipSolver.modes returns."""
        off, f = self._flat(freq_lists)
        npts = self.n_int + 1
        cap = npts if node_cap is None else node_cap
        n = f.size
        dets = np.empty(n)
        vec = np.empty((n, npts * 4)) if want_vec else None
        fields = np.empty((n, len(POST_FIELDS), npts)) if want_fields else None
        ids = (ModeId * max(n, 1))()
        nodes = np.full((n, max(cap, 1)), np.nan)
        hs = (C.c_void_p * max(self.n_sys, 1))(*[q._h for q in posts])
        _check(load_library().storm_sweep_modes(self._h, hs, _sp(off), _p(f), _p(dets), _p(vec) if want_vec else None,
                                                _p(fields) if want_fields else None, ids, _p(nodes) if cap else None, cap))
        allr = _post_result(n, fields, ids, nodes, cap)
        res = []
        for u in range(self.n_sys):
            a, b = int(off[u]), int(off[u + 1])
            res.append(dict(fields=fields[a:b] if want_fields else None, ids=allr["ids"][a:b], n_nodes=allr["n_nodes"][a:b],
                            nodes=allr["nodes"][a:b], det=dets[a:b], vec=vec[a:b] if want_vec else None))
        return res


def _sweep_resident(self, posts, freq_lists, node_cap=None):
    """storm_sweep_modes_resident: ids and nodes per system; the 13 fields of every mode stay in HBM for
    coupling_matrices_resident (mode index = position in the concatenation of freq_lists)."""
    off, f = self._flat(freq_lists)
    npts = self.n_int + 1
    cap = npts if node_cap is None else node_cap
    n = f.size
    ids = (ModeId * max(n, 1))()
    nodes = np.full((n, max(cap, 1)), np.nan)
    hs = (C.c_void_p * max(self.n_sys, 1))(*[q._h if q is not None else None for q in posts])
    _check(load_library().storm_sweep_modes_resident(self._h, hs, _sp(off), _p(f), ids, _p(nodes) if cap else None, cap))
    self._resident_freq = f
    allr = _post_result(n, None, ids, nodes, cap)
    return [dict(ids=allr["ids"][int(off[u]):int(off[u + 1])], n_nodes=allr["n_nodes"][int(off[u]):int(off[u + 1])],
                 nodes=allr["nodes"][int(off[u]):int(off[u + 1])]) for u in range(self.n_sys)]


def _sweep_coupling(self, post, mode_index, ells, m, metric):
    """storm_sweep_coupling_matrices on resident modes; dict name -> [n, n] array indexed [left, right]."""
    idx = np.ascontiguousarray(mode_index, dtype=np.uintp)
    ells = np.ascontiguousarray(ells, dtype=np.uint64)
    f = _f64(self._resident_freq[idx])
    n = idx.size
    arrs = [_f64(metric[k]) for k in METRIC_FIELDS]
    md = MetricDesc(*[_p(a) for a in arrs])
    out = np.zeros((len(COUPLING_MATRICES), n * n))
    _check(load_library().storm_sweep_coupling_matrices(self._h, post._h, n, _sp(idx), ells.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                        _p(f), m, C.byref(md), _p(out)))
    return {k: out[i].reshape(n, n).T.copy() for i, k in enumerate(COUPLING_MATRICES)}


HipSweep.modes_resident = _sweep_resident
HipSweep.coupling_matrices_resident = _sweep_coupling


def _post_result(n, fields, ids, nodes, cap):
    idarr = np.array([[i.cross_clockwise, i.cross_counter_clockwise, i.radial_order] for i in ids[:n]],
                     dtype=np.int64).reshape(n, 3)
    nn = np.array([i.n_nodes for i in ids[:n]], dtype=np.int64)
    return dict(fields=fields, ids=idarr, n_nodes=nn, nodes=[nodes[k, :min(int(nn[k]), cap)].copy() for k in range(n)])


class HipPostproc:
    """One storm_postproc handle: the model (evaluated on the solver grid) resident on the device.
    `model` maps the names of MODEL_FIELDS to arrays of equal length."""

    def __init__(self, model):
        arrs = {k: _f64(model[k]) for k in MODEL_FIELDS}
        n = arrs["r_coord"].size
        for k, a in arrs.items():
            if a.size != n:
                raise ValueError(f"{k}: expected {n} entries, got {a.size}")
        d = ModelDesc(n, *[_p(arrs[k]) for k in MODEL_FIELDS])
        h = C.c_void_p()
        _check(load_library().storm_postproc_create(C.byref(d), C.byref(h)))
        self._h = h
        self.n_pts = n

    def close(self):
        if getattr(self, "_h", None):
            load_library().storm_postproc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        """storm_postproc_set_option ("arithmetic": 0 parity, 1 FMA build of the coupling integrals)."""
        _check(load_library().storm_postproc_set_option(self._h, key.encode(), int(value)))

    def postprocess(self, ell, m, freqs, vecs, want_fields=True, node_cap=None):
        """storm_postprocess_batch on host eigenvectors [n][n_pts*4]."""
        f = _f64(np.atleast_1d(freqs))
        v = _f64(vecs).reshape(f.size, -1) if f.size else np.empty((0, self.n_pts * 4))
        if f.size and v.shape[1] != self.n_pts * 4:
            raise ValueError("eigenvectors do not live on the model grid")
        cap = self.n_pts if node_cap is None else node_cap
        fields = np.empty((f.size, len(POST_FIELDS), self.n_pts)) if want_fields else None
        ids = (ModeId * max(f.size, 1))()
        nodes = np.full((f.size, max(cap, 1)), np.nan)
        _check(load_library().storm_postprocess_batch(self._h, ell, m, _p(f), f.size, _p(v), _p(fields) if want_fields else None,
                                                      ids, _p(nodes) if cap else None, cap))
        return _post_result(f.size, fields, ids, nodes, cap)

    def coupling_matrices(self, ells, freqs, fields, m, metric):
        """storm_coupling_matrices: dict name -> [n, n] array indexed [left, right] (perturb_deformed's six matrices)."""
        ells = np.ascontiguousarray(ells, dtype=np.uint64)
        f = _f64(freqs)
        fl = _f64(fields)
        n = f.size
        if fl.size != n * len(POST_FIELDS) * self.n_pts:
            raise ValueError("fields must be [n, 13, n_pts]")
        arrs = [_f64(metric[k]) for k in METRIC_FIELDS]
        for a in arrs:
            if a.size != self.n_pts:
                raise ValueError("metric arrays must have n_pts entries")
        md = MetricDesc(*[_p(a) for a in arrs])
        out = np.zeros((len(COUPLING_MATRICES), n * n))
        _check(load_library().storm_coupling_matrices(self._h, n, ells.ctypes.data_as(C.POINTER(C.c_uint64)), _p(f), _p(fl), m,
                                                      C.byref(md), _p(out)))
        return {k: out[i].reshape(n, n).T.copy() for i, k in enumerate(COUPLING_MATRICES)}

    def last_kernel_ms(self):
        return float(load_library().storm_postproc_last_kernel_ms(self._h))
