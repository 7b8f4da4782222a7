"""ctypes binding of the C++ host mirror (storm_amd/csrc/host/storm_host.hpp -> libstorm_host.so).

Names follow the reference's Rust API for the scan path (DiscreteModel, Polytrope0,
construct_polytrope, ErasedSolver.{det,eigenvector,scan_and_optimize}, Precision, scan) so the
parity tests read like the reference's own tests (src/dynamic_interface.rs:223-522)."""
import ctypes as C
import os

import numpy as np

from . import ffi
from .ffi import Bracket, SCHEMES

_dp = C.POINTER(C.c_double)
_lib = None


def host_lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libstorm_host.so")


def load_host_library():
    global _lib
    if _lib is not None:
        return _lib
    ffi.load_library()  # fail loudly if the HIP library is missing
    path = host_lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `make -C storm_amd/csrc`")
    L = C.CDLL(path)
    L.storm_host_last_error.restype = C.c_char_p
    L.storm_host_model_from_gsm.restype = C.c_void_p
    L.storm_host_model_from_gsm.argtypes = [C.c_size_t] + [_dp] * 7 + [C.c_double] * 2
    L.storm_host_model_polytrope.restype = C.c_void_p
    L.storm_host_model_polytrope.argtypes = [C.c_double] * 3
    L.storm_host_model_poly0.restype = C.c_void_p
    L.storm_host_model_poly0.argtypes = [C.c_double]
    L.storm_host_model_free.argtypes = [C.c_void_p]
    L.storm_host_model_len.restype = C.c_size_t
    L.storm_host_model_len.argtypes = [C.c_void_p]
    L.storm_host_model_get.argtypes = [C.c_void_p, C.c_int, _dp]
    L.storm_host_model_set.argtypes = [C.c_void_p, C.c_int, _dp]
    L.storm_host_model_set_rotation_constant.argtypes = [C.c_void_p, C.c_double]
    L.storm_host_model_grid.restype = C.c_size_t
    L.storm_host_model_grid.argtypes = [C.c_void_p, C.c_size_t, _dp, C.c_size_t]
    L.storm_host_solver_new.restype = C.c_void_p
    L.storm_host_solver_new.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_int, _dp, C.c_size_t]
    L.storm_host_solver_free.argtypes = [C.c_void_p]
    L.storm_host_solver_len.restype = C.c_size_t
    L.storm_host_solver_len.argtypes = [C.c_void_p]
    L.storm_host_solver_handle.restype = C.c_void_p
    L.storm_host_solver_handle.argtypes = [C.c_void_p]
    L.storm_host_solver_det.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp]
    L.storm_host_multi_new.restype = C.c_void_p
    L.storm_host_multi_new.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_int, _dp, C.c_size_t, C.POINTER(C.c_int), C.c_size_t]
    L.storm_host_multi_free.argtypes = [C.c_void_p]
    L.storm_host_multi_len.restype = C.c_size_t
    L.storm_host_multi_len.argtypes = [C.c_void_p]
    L.storm_host_multi_points.restype = C.c_size_t
    L.storm_host_multi_points.argtypes = [C.c_void_p]
    L.storm_host_multi_handle.restype = C.c_void_p
    L.storm_host_multi_handle.argtypes = [C.c_void_p]
    L.storm_host_solver_scan_and_optimize.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_int, C.c_uint64, C.c_double,
                                                      C.POINTER(Bracket), C.c_size_t, C.POINTER(C.c_size_t),
                                                      C.POINTER(C.c_int)]
    L.storm_host_solver_eigenvectors.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp]
    L.storm_host_scan.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_uint64, C.c_int64, C.c_double, C.c_double,
                                  C.c_size_t, C.c_int, C.c_double, C.c_int, C.POINTER(Bracket), _dp, C.c_size_t,
                                  C.POINTER(C.c_size_t)]
    L.storm_host_polytrope0_exact.restype = C.c_double
    L.storm_host_polytrope0_exact.argtypes = [C.c_double, C.c_uint64, C.c_int64]
    L.storm_host_linspace.argtypes = [C.c_double, C.c_double, C.c_size_t, _dp]
    L.storm_host_model_eval.restype = C.c_void_p
    L.storm_host_model_eval.argtypes = [C.c_void_p, _dp, C.c_size_t]
    L.storm_host_postproc_new.restype = C.c_void_p
    L.storm_host_postproc_new.argtypes = [C.c_void_p]
    L.storm_host_postproc_free.argtypes = [C.c_void_p]
    _i64p, _u64p = C.POINTER(C.c_int64), C.POINTER(C.c_uint64)
    L.storm_host_post_process.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, _dp, _dp, C.c_size_t, _dp, _i64p, _u64p,
                                          _dp, C.c_size_t]
    L.storm_host_perturb_structure.argtypes = [C.c_void_p, C.c_double, C.POINTER(_dp), _dp]
    L.storm_host_coupling_matrices.argtypes = [C.c_void_p, C.c_size_t, _u64p, _dp, _dp, C.c_int64, C.POINTER(_dp), _dp]
    L.storm_host_modes.argtypes = [C.c_void_p, C.c_void_p, _dp, C.c_size_t, C.c_int, _dp, _i64p, _u64p, _dp, C.c_size_t]
    L.storm_host_rev_linspace.argtypes = [C.c_double, C.c_double, C.c_size_t, _dp]
    _sp, _vpp = C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)
    L.storm_host_sweep_new.restype = C.c_void_p
    L.storm_host_sweep_new.argtypes = [_vpp, _u64p, _i64p, C.c_size_t, C.c_int, _dp, C.c_size_t]
    L.storm_host_sweep_free.argtypes = [C.c_void_p]
    L.storm_host_sweep_len.restype = C.c_size_t
    L.storm_host_sweep_len.argtypes = [C.c_void_p]
    L.storm_host_sweep_handle.restype = C.c_void_p
    L.storm_host_sweep_handle.argtypes = [C.c_void_p]
    L.storm_host_sweep_det.argtypes = [C.c_void_p, _sp, _dp, _dp]
    L.storm_host_sweep_scan_and_optimize.argtypes = [C.c_void_p, _sp, _dp, C.c_int, C.c_uint64, C.c_double,
                                                     C.POINTER(Bracket), C.c_size_t, _sp, C.POINTER(C.c_int)]
    L.storm_host_sweep_eigenvectors.argtypes = [C.c_void_p, _sp, _dp, _dp]
    L.storm_host_sweep_modes.argtypes = [C.c_void_p, _vpp, _sp, _dp, C.c_int, _dp, _i64p, _u64p, _dp, C.c_size_t]
    _lib = L
    return L


class HostError(RuntimeError):
    pass


def _err():
    return HostError(load_host_library().storm_host_last_error().decode())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def linspace(lo, hi, n):
    out = np.empty(n)
    load_host_library().storm_host_linspace(lo, hi, n, _p(out))
    return out


def rev_linspace(lo, hi, n):
    out = np.empty(n)
    load_host_library().storm_host_rev_linspace(lo, hi, n, _p(out))
    return out


class Precision:
    def __init__(self, kind, ulp=1, rel=0.0):
        self.kind, self.ulp, self.rel = kind, ulp, rel

    @classmethod
    def ULP(cls, u):
        return cls(0, ulp=int(u))

    @classmethod
    def Relative(cls, f):
        return cls(1, rel=float(f))


class Model:
    """DiscreteModel / ContinuousModel handle on the C++ host side."""
    FIELDS = ["r_coord", "v", "u", "gamma1", "a_star", "c1", "rot", "m_coord", "rho", "p"]

    def __init__(self, handle, poly0=False):
        if not handle:
            raise _err()
        self._h = C.c_void_p(handle)
        self.poly0 = poly0

    @classmethod
    def from_gsm_npz(cls, path):
        d = np.load(path)
        arrs = [_f64(d[k]) for k in ("r", "M_r", "rho", "P", "Gamma_1", "N2", "Omega_rot")]
        return cls(load_host_library().storm_host_model_from_gsm(int(d["n"]), *[_p(a) for a in arrs],
                                                                 float(d["R_star"]), float(d["M_star"])))

    @classmethod
    def construct_polytrope(cls, n, gamma1, step):
        return cls(load_host_library().storm_host_model_polytrope(n, gamma1, step))

    @classmethod
    def Polytrope0(cls, gamma1):
        return cls(load_host_library().storm_host_model_poly0(gamma1), poly0=True)

    def __len__(self):
        return load_host_library().storm_host_model_len(self._h)

    def field(self, name):
        out = np.empty(len(self))
        load_host_library().storm_host_model_get(self._h, self.FIELDS.index(name), _p(out))
        return out

    def set_field(self, name, values):
        v = _f64(values)
        assert v.size == len(self)
        load_host_library().storm_host_model_set(self._h, self.FIELDS.index(name), _p(v))

    def set_rotation_constant(self, value):
        if load_host_library().storm_host_model_set_rotation_constant(self._h, value) != 0:
            raise _err()

    def grid(self, scale=1):
        L = load_host_library()
        n = L.storm_host_model_grid(self._h, scale, None, 0)
        out = np.empty(n)
        L.storm_host_model_grid(self._h, scale, _p(out), n)
        return out

    def perturb_structure(self, rot):
        """perturb_structure (reference src/perturbed.rs:324-460; the `deform` command): dict of the six metric arrays
        + mass_delta.  Host-side, like in the reference."""
        arrs = [np.zeros(len(self)) for _ in ffi.METRIC_FIELDS]
        ptrs = (_dp * 6)(*[_p(a) for a in arrs])
        md = np.zeros(1)
        if load_host_library().storm_host_perturb_structure(self._h, rot, ptrs, _p(md)) != 0:
            raise _err()
        out = dict(zip(ffi.METRIC_FIELDS, arrs))
        out["mass_delta"] = float(md[0])
        return out

    def eval(self, grid):
        """ContinuousModel::eval on `grid` (what post-process evaluates, reference src/main.rs:892)."""
        g = _f64(grid)
        return Model(load_host_library().storm_host_model_eval(self._h, _p(g), g.size))

    def __del__(self):
        try:
            load_host_library().storm_host_model_free(self._h)
        except Exception:
            pass


class ErasedSolver:
    """ErasedSolver::new(model, Rotating1D::new(ell, m), scheme, solver_grid) on the MI355X."""

    def __init__(self, model, ell, m, scheme, solver_grid):
        if isinstance(scheme, str):
            scheme = SCHEMES.index(scheme)
        g = _f64(solver_grid)
        h = load_host_library().storm_host_solver_new(model._h, ell, m, scheme, _p(g), g.size)
        if not h:
            raise _err()
        self._h = C.c_void_p(h)
        self._model = model
        self.n_int = load_host_library().storm_host_solver_len(self._h)
        self.scheme, self.ell, self.m = scheme, ell, m

    def raw_handle(self):
        return load_host_library().storm_host_solver_handle(self._h)

    def set_option(self, key, value):
        L = ffi.load_library()
        rc = L.storm_solver_set_option(C.c_void_p(self.raw_handle()), key.encode(), int(value))
        if rc != 0:
            raise ffi.StormError(rc, L.storm_last_error().decode())

    def stats(self, reset=False):
        L = ffi.load_library()
        s = ffi.Stats()
        rc = L.storm_solver_stats(C.c_void_p(self.raw_handle()), C.byref(s), 1 if reset else 0)
        if rc != 0:
            raise ffi.StormError(rc, L.storm_last_error().decode())
        return {k: int(getattr(s, k)) for k, _ in ffi.Stats._fields_}

    def kernel_times(self, reset=False):
        return ffi.kernel_times(self.raw_handle(), reset)

    def det(self, freqs):
        f = _f64(np.atleast_1d(freqs))
        out = np.empty(f.size)
        if load_host_library().storm_host_solver_det(self._h, _p(f), f.size, _p(out)) != 0:
            raise _err()
        return out

    def scan_and_optimize(self, freq_grid, precision):
        f = _f64(freq_grid)
        cap = max(1, f.size)
        out = (Bracket * cap)()
        n_out, inv = C.c_size_t(0), C.c_int(0)
        rc = load_host_library().storm_host_solver_scan_and_optimize(self._h, _p(f), f.size, precision.kind,
                                                                     precision.ulp, precision.rel, out, cap,
                                                                     C.byref(n_out), C.byref(inv))
        if rc != 0:
            raise _err()
        res = [dict(lo_x=b.lo_x, lo_f=b.lo_f, hi_x=b.hi_x, hi_f=b.hi_f, root=b.root, evals=int(b.evals))
               for b in out[:n_out.value]]
        return res, bool(inv.value)

    def eigenvector(self, freqs):
        f = _f64(np.atleast_1d(freqs))
        out = np.empty((f.size, (self.n_int + 1) * 4))
        if load_host_library().storm_host_solver_eigenvectors(self._h, _p(f), f.size, _p(out)) != 0:
            raise _err()
        return out

    def __del__(self):
        try:
            load_host_library().storm_host_solver_free(self._h)
        except Exception:
            pass


class MultiSolver:
    """storm_host::MultiSolver: ErasedSolver over several devices of the node from this process (storm_multi_*);
    `.hip` is the ffi.HipMulti view of its handle (scan_and_optimize, eigenvector, set_option, stats)."""

    def __init__(self, model, ell, m, scheme, solver_grid, devices):
        if isinstance(scheme, str):
            scheme = SCHEMES.index(scheme)
        g = _f64(solver_grid)
        devs = (C.c_int * len(devices))(*devices)
        L = load_host_library()
        h = L.storm_host_multi_new(model._h, ell, m, scheme, _p(g), g.size, devs, len(devices))
        if not h:
            raise _err()
        self._h = C.c_void_p(h)
        self._model = model
        self.n_int = L.storm_host_multi_len(self._h)
        self.hip = ffi.HipMulti.borrow(L.storm_host_multi_handle(self._h), self.n_int, L.storm_host_multi_points(self._h),
                                       scheme, len(devices), owner=self)

    def __del__(self):
        try:
            load_host_library().storm_host_multi_free(self._h)
        except Exception:
            pass


def scan(model, solver_grid, ell, m, lower, upper, steps, inverse=False, precision=1e-8, scheme="colloc4"):
    """StormState::scan (reference src/main.rs:793-838), dynamical frequency units."""
    if isinstance(scheme, str):
        scheme = SCHEMES.index(scheme)
    g = _f64(solver_grid)
    cap = max(1, steps)
    out = (Bracket * cap)()
    vec = np.empty((cap, g.size * 4))
    n_out = C.c_size_t(0)
    rc = load_host_library().storm_host_scan(model._h, _p(g), g.size, ell, m, lower, upper, steps, 1 if inverse else 0,
                                             precision, scheme, out, _p(vec), cap, C.byref(n_out))
    if rc != 0:
        raise _err()
    n = n_out.value
    res = [dict(lo_x=b.lo_x, lo_f=b.lo_f, hi_x=b.hi_x, hi_f=b.hi_f, root=b.root, evals=int(b.evals))
           for b in out[:n]]
    return res, vec[:n].copy()


POST_FIELDS = ffi.POST_FIELDS


class PostProcessor:
    """StormState::post_process (reference src/main.rs:878-899) on the device: Rotating1DPostprocessing::new
    for every solution.  `model_on_grid` = model.eval(solver_grid)."""

    def __init__(self, model_on_grid):
        h = load_host_library().storm_host_postproc_new(model_on_grid._h)
        if not h:
            raise _err()
        self._h = C.c_void_p(h)
        self._model = model_on_grid
        self.n_pts = len(model_on_grid)

    def _buffers(self, n, want_fields):
        fields = np.empty((n, len(POST_FIELDS), self.n_pts)) if want_fields else None
        ids = np.empty((n, 3), dtype=np.int64)
        nn = np.zeros(n, dtype=np.uint64)
        nodes = np.full((n, self.n_pts), np.nan)
        return fields, ids, nn, nodes

    @staticmethod
    def _result(fields, ids, nn, nodes):
        return dict(fields=fields, ids=ids, n_nodes=nn.astype(np.int64),
                    nodes=[nodes[k, :int(nn[k])].copy() for k in range(len(nn))])

    def post_process(self, ell, m, roots, eigenvectors):
        f = _f64(np.atleast_1d(roots))
        v = _f64(eigenvectors).reshape(f.size, -1)
        fields, ids, nn, nodes = self._buffers(f.size, True)
        rc = load_host_library().storm_host_post_process(
            self._h, ell, m, _p(f), _p(v), f.size, _p(fields), ids.ctypes.data_as(C.POINTER(C.c_int64)),
            nn.ctypes.data_as(C.POINTER(C.c_uint64)), _p(nodes), self.n_pts)
        if rc != 0:
            raise _err()
        return self._result(fields, ids, nn, nodes)

    def modes(self, solver, roots, want_fields=False):
        """Eigenvectors + post-processing in one device pass (eigenvectors never leave HBM)."""
        f = _f64(np.atleast_1d(roots))
        fields, ids, nn, nodes = self._buffers(f.size, want_fields)
        rc = load_host_library().storm_host_modes(
            self._h, solver._h, _p(f), f.size, 1 if want_fields else 0, _p(fields) if want_fields else None,
            ids.ctypes.data_as(C.POINTER(C.c_int64)), nn.ctypes.data_as(C.POINTER(C.c_uint64)), _p(nodes), self.n_pts)
        if rc != 0:
            raise _err()
        return self._result(fields, ids, nn, nodes)

    def coupling_matrices(self, ells, freqs, fields, m, metric):
        """perturb_deformed's six coupling matrices (reference src/perturbed.rs:110-263) for modes given by their
        post-processing fields [n, 13, n_pts]; dict name -> [n, n] array indexed [left, right]."""
        ells = np.ascontiguousarray(ells, dtype=np.uint64)
        f, fl = _f64(freqs), _f64(fields)
        n = f.size
        arrs = [_f64(metric[k]) for k in ffi.METRIC_FIELDS]
        ptrs = (_dp * 6)(*[_p(a) for a in arrs])
        out = np.zeros((len(ffi.COUPLING_MATRICES), n * n))
        rc = load_host_library().storm_host_coupling_matrices(self._h, n, ells.ctypes.data_as(C.POINTER(C.c_uint64)), _p(f),
                                                              _p(fl), m, ptrs, _p(out))
        if rc != 0:
            raise _err()
        return {k: out[i].reshape(n, n).T.copy() for i, k in enumerate(ffi.COUPLING_MATRICES)}

    def __del__(self):
        try:
            load_host_library().storm_host_postproc_free(self._h)
        except Exception:
            pass


class SolverSweep:
    """Many (model, ell, m) systems on one solver grid evaluated in lock-step on the device (storm_sweep): one
    ErasedSolver per `scan` line of a rotation-sweep batch input (reference src/main.rs:815-831,
    scripts/generate-batch-input-deformed.py:19-35) as one object.  `systems`: list of (model, ell, m); models may
    repeat.  Per-system inputs/outputs are lists indexed by system; results are bit-identical to one ErasedSolver each."""

    def __init__(self, systems, scheme, solver_grid):
        if isinstance(scheme, str):
            scheme = SCHEMES.index(scheme)
        g = _f64(solver_grid)
        n = len(systems)
        models = (C.c_void_p * max(n, 1))(*[mdl._h for mdl, _, _ in systems])
        ells = np.array([e for _, e, _ in systems], dtype=np.uint64)
        ms = np.array([m for _, _, m in systems], dtype=np.int64)
        h = load_host_library().storm_host_sweep_new(models, ells.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                     ms.ctypes.data_as(C.POINTER(C.c_int64)), n, scheme, _p(g), g.size)
        if not h:
            raise _err()
        self._h = C.c_void_p(h)
        self._models = [mdl for mdl, _, _ in systems]
        self.n_sys, self.scheme = n, scheme
        self.n_int = load_host_library().storm_host_sweep_len(self._h)

    def raw_handle(self):
        return load_host_library().storm_host_sweep_handle(self._h)

    def set_option(self, key, value):
        L = ffi.load_library()
        rc = L.storm_sweep_set_option(C.c_void_p(self.raw_handle()), key.encode(), int(value))
        if rc != 0:
            raise ffi.StormError(rc, L.storm_last_error().decode())

    def stats(self, reset=False):
        L = ffi.load_library()
        s = ffi.Stats()
        rc = L.storm_sweep_stats(C.c_void_p(self.raw_handle()), C.byref(s), 1 if reset else 0)
        if rc != 0:
            raise ffi.StormError(rc, L.storm_last_error().decode())
        return {k: int(getattr(s, k)) for k, _ in ffi.Stats._fields_}

    def _flat(self, lists):
        if len(lists) != self.n_sys:
            raise ValueError(f"expected {self.n_sys} per-system lists, got {len(lists)}")
        lists = [_f64(np.atleast_1d(x)).ravel() for x in lists]
        off = np.zeros(self.n_sys + 1, dtype=np.uintp)
        np.cumsum([x.size for x in lists], out=off[1:])
        return off, _f64(np.concatenate(lists))

    @staticmethod
    def _sp(a):
        return a.ctypes.data_as(C.POINTER(C.c_size_t))

    def det(self, freq_lists):
        off, f = self._flat(freq_lists)
        out = np.empty(f.size)
        if load_host_library().storm_host_sweep_det(self._h, self._sp(off), _p(f), _p(out)) != 0:
            raise _err()
        return [out[off[u]:off[u + 1]] for u in range(self.n_sys)]

    def scan_and_optimize(self, freq_grids, precision):
        """-> per system (brackets, had_invalid), as ErasedSolver.scan_and_optimize returns them."""
        off, f = self._flat(freq_grids)
        cap = max(1, f.size)
        out = (Bracket * cap)()
        ooff = np.zeros(self.n_sys + 1, dtype=np.uintp)
        inv = (C.c_int * max(self.n_sys, 1))()
        rc = load_host_library().storm_host_sweep_scan_and_optimize(self._h, self._sp(off), _p(f), precision.kind,
                                                                    precision.ulp, precision.rel, out, cap,
                                                                    self._sp(ooff), inv)
        if rc != 0:
            raise _err()
        res = []
        for u in range(self.n_sys):
            br = [dict(lo_x=b.lo_x, lo_f=b.lo_f, hi_x=b.hi_x, hi_f=b.hi_f, root=b.root, evals=int(b.evals))
                  for b in out[int(ooff[u]):int(ooff[u + 1])]]
            res.append((br, bool(inv[u])))
        return res

    def eigenvector(self, root_lists):
        off, f = self._flat(root_lists)
        out = np.empty((f.size, (self.n_int + 1) * 4))
        if load_host_library().storm_host_sweep_eigenvectors(self._h, self._sp(off), _p(f), _p(out)) != 0:
            raise _err()
        return [out[off[u]:off[u + 1]] for u in range(self.n_sys)]

    def modes(self, posts, root_lists, want_fields=False):
        """Eigenvectors + post-processing of every system's modes in one device pass; posts: one PostProcessor per
        system (may repeat).  -> per system the dict PostProcessor.modes returns."""
        off, f = self._flat(root_lists)
        n, npts = f.size, self.n_int + 1
        fields = np.empty((n, len(POST_FIELDS), npts)) if want_fields else None
        ids = np.empty((n, 3), dtype=np.int64)
        nn = np.zeros(n, dtype=np.uint64)
        nodes = np.full((n, npts), np.nan)
        hs = (C.c_void_p * max(self.n_sys, 1))(*[q._h if q is not None else None for q in posts])
        rc = load_host_library().storm_host_sweep_modes(
            self._h, hs, self._sp(off), _p(f), 1 if want_fields else 0, _p(fields) if want_fields else None,
            ids.ctypes.data_as(C.POINTER(C.c_int64)), nn.ctypes.data_as(C.POINTER(C.c_uint64)), _p(nodes), npts)
        if rc != 0:
            raise _err()
        res = []
        for u in range(self.n_sys):
            a, b = int(off[u]), int(off[u + 1])
            res.append(PostProcessor._result(fields[a:b] if want_fields else None, ids[a:b], nn[a:b], nodes[a:b]))
        return res

    def __del__(self):
        try:
            load_host_library().storm_host_sweep_free(self._h)
        except Exception:
            pass


def polytrope0_exact(gamma1, ell, radial_order):
    return load_host_library().storm_host_polytrope0_exact(gamma1, ell, radial_order)
