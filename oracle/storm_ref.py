"""ctypes wrapper over the CPU ORACLE (oracle/_build/libstorm_ref.so).

Test infrastructure only -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package storm_amd never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libstorm_ref.so")

POST_FIELDS = ["y1", "y2", "y3", "y4", "xi_r", "xi_h", "xi_tn", "xi_tp", "psi", "dpsi", "rho", "p", "chi"]
METRIC_FIELDS = ["alpha", "dalpha", "ddalpha", "beta", "dbeta", "ddbeta"]
COUPLING_MATRICES = ["d_squared", "d_linear", "d_zero", "r_linear", "r_zero", "l_zero"]
SCHEMES = ["colloc2", "colloc4", "colloc6", "colloc8", "magnus2", "magnus4", "magnus6", "magnus8"]


def build(force=False):
    """Compile the oracle with g++ (-O2 -ffp-contract=off)."""
    src_newer = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("storm_ref.hpp", "storm_ref_capi.cpp", "Makefile"))
    if force or src_newer:
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _LIB_PATH


class Bracket(C.Structure):
    _fields_ = [("lo_x", C.c_double), ("lo_f", C.c_double), ("hi_x", C.c_double), ("hi_f", C.c_double),
                ("root", C.c_double), ("evals", C.c_uint64)]


_lib = None
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.sref_last_error.restype = C.c_char_p
        L.sref_model_from_gsm.restype = C.c_void_p
        L.sref_model_from_gsm.argtypes = [C.c_size_t] + [_dp] * 7 + [C.c_double, C.c_double]
        L.sref_model_polytrope.restype = C.c_void_p
        L.sref_model_polytrope.argtypes = [C.c_double] * 3
        L.sref_model_poly0.restype = C.c_void_p
        L.sref_model_poly0.argtypes = [C.c_double]
        L.sref_model_free.argtypes = [C.c_void_p]
        L.sref_model_len.restype = C.c_size_t
        L.sref_model_len.argtypes = [C.c_void_p]
        L.sref_model_get.argtypes = [C.c_void_p, C.c_int, _dp]
        L.sref_model_set.argtypes = [C.c_void_p, C.c_int, _dp]
        L.sref_model_set_rot_const.argtypes = [C.c_void_p, C.c_double]
        L.sref_model_freq_scale.restype = C.c_double
        L.sref_model_freq_scale.argtypes = [C.c_void_p]
        L.sref_model_grid.restype = C.c_size_t
        L.sref_model_grid.argtypes = [C.c_void_p, C.c_size_t, _dp, C.c_size_t]
        L.sref_solver_new.restype = C.c_void_p
        L.sref_solver_new.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.c_int, _dp, C.c_size_t]
        L.sref_solver_free.argtypes = [C.c_void_p]
        L.sref_solver_len.restype = C.c_size_t
        L.sref_solver_len.argtypes = [C.c_void_p]
        L.sref_solver_points.argtypes = [C.c_void_p]
        L.sref_solver_system.argtypes = [C.c_void_p] + [_dp] * 7
        L.sref_det_batch.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, C.c_int]
        L.sref_step_matrix.argtypes = [C.c_void_p, C.c_size_t, C.c_double, _dp]
        L.sref_scan_and_optimize.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_uint64, C.POINTER(Bracket),
                                             C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.c_int,
                                             _dp, C.POINTER(C.c_uint64)]
        L.sref_precision_relative_to_ulp.restype = C.c_uint64
        L.sref_precision_relative_to_ulp.argtypes = [C.c_double]
        L.sref_optimize.argtypes = [C.c_void_p] + [C.c_double] * 4 + [C.c_uint64, C.POINTER(Bracket)]
        L.sref_eigenvector_batch.argtypes = [C.c_void_p, _dp, C.c_size_t, _dp, _dp, C.c_int]
        L.sref_exp.argtypes = [C.c_int, _dp]
        L.sref_inv4.argtypes = [_dp, _dp]
        L.sref_matmul4.argtypes = [_dp, _dp, _dp]
        L.sref_count_flops.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_uint64)]
        L.sref_polytrope0_exact.restype = C.c_double
        L.sref_polytrope0_exact.argtypes = [C.c_double, C.c_uint64, C.c_int64]
        L.sref_linspace.argtypes = [C.c_double, C.c_double, C.c_size_t, _dp]
        L.sref_model_eval.restype = C.c_void_p
        L.sref_model_eval.argtypes = [C.c_void_p, _dp, C.c_size_t]
        L.sref_postprocess_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, _dp, C.c_size_t, _dp, _dp,
                                             C.POINTER(C.c_int64), _dp, C.c_size_t, C.POINTER(C.c_uint64), C.c_int]
        L.sref_gaunt.argtypes = [C.c_uint64, C.c_int64, _dp]
        L.sref_coupling_matrices.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), _dp, _dp, C.c_int64,
                                             C.POINTER(_dp), _dp, C.POINTER(C.c_uint64), C.c_int]
        L.sref_perturb_structure.argtypes = [C.c_void_p, C.c_double, C.POINTER(_dp), _dp]
        L.sref_powi.restype = C.c_double
        L.sref_powi.argtypes = [C.c_double, C.c_int]
        L.sref_rev_linspace.argtypes = [C.c_double, C.c_double, C.c_size_t, _dp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _check(rc):
    if rc != 0:
        raise RuntimeError("oracle: " + lib().sref_last_error().decode())


def linspace(lo, hi, n):
    out = np.empty(n)
    lib().sref_linspace(lo, hi, n, _p(out))
    return out


def rev_linspace(lo, hi, n):
    out = np.empty(n)
    lib().sref_rev_linspace(lo, hi, n, _p(out))
    return out


def relative_to_ulp(rel):
    return int(lib().sref_precision_relative_to_ulp(rel))


class Model:
    FIELDS = ["r_coord", "v", "u", "gamma1", "a_star", "c1", "rot", "m_coord", "rho", "p"]

    def __init__(self, handle, poly0=False):
        if not handle:
            raise RuntimeError("oracle: " + lib().sref_last_error().decode())
        self._h = C.c_void_p(handle)
        self.poly0 = poly0

    @classmethod
    def from_gsm_npz(cls, path):
        d = np.load(path)
        arrs = [_f64(d[k]) for k in ("r", "M_r", "rho", "P", "Gamma_1", "N2", "Omega_rot")]
        n = int(d["n"])
        return cls(lib().sref_model_from_gsm(n, *[_p(a) for a in arrs], float(d["R_star"]), float(d["M_star"])))

    @classmethod
    def polytrope(cls, n, gamma1, step):
        return cls(lib().sref_model_polytrope(n, gamma1, step))

    @classmethod
    def poly0(cls, gamma1):
        return cls(lib().sref_model_poly0(gamma1), poly0=True)

    def __len__(self):
        return lib().sref_model_len(self._h)

    def field(self, name):
        out = np.empty(len(self))
        lib().sref_model_get(self._h, self.FIELDS.index(name), _p(out))
        return out

    def set_field(self, name, values):
        v = _f64(values)
        assert v.size == len(self)
        lib().sref_model_set(self._h, self.FIELDS.index(name), _p(v))

    def set_rotation_constant(self, value):
        lib().sref_model_set_rot_const(self._h, value)

    def freq_scale(self):
        return lib().sref_model_freq_scale(self._h)

    def grid(self, scale=1):
        n = lib().sref_model_grid(self._h, scale, None, 0)
        out = np.empty(n)
        lib().sref_model_grid(self._h, scale, _p(out), n)
        return out

    def eval(self, grid):
        """ContinuousModel::eval: this model on `grid` (linear interpolation, or Polytrope0's formulas)."""
        g = _f64(grid)
        return Model(lib().sref_model_eval(self._h, _p(g), g.size))

    def postprocess(self, ell, m, freqs, vecs, threads=1, node_cap=None):
        """Rotating1DPostprocessing::new (src/postprocessing.rs:58-301) for each (freq, eigenvector) on this
        (already evaluated) model.  Returns dict(fields [n,13,n_pts], ids [n,3] = cw, ccw, radial order,
        nodes = list of arrays)."""
        f = _f64(np.atleast_1d(freqs))
        v = _f64(vecs).reshape(f.size, -1)
        npts = len(self)
        assert v.shape[1] == npts * 4, "eigenvector does not live on the model grid"
        cap = npts if node_cap is None else node_cap
        fields = np.empty((f.size, len(POST_FIELDS), npts))
        ids = np.empty((f.size, 3), dtype=np.int64)
        nodes = np.full((f.size, max(cap, 1)), np.nan)
        nn = np.zeros(f.size, dtype=np.uint64)
        _check(lib().sref_postprocess_batch(self._h, ell, m, _p(f), f.size, _p(v), _p(fields),
                                            ids.ctypes.data_as(C.POINTER(C.c_int64)), _p(nodes), cap,
                                            nn.ctypes.data_as(C.POINTER(C.c_uint64)), threads))
        return dict(fields=fields, ids=ids, n_nodes=nn.astype(np.int64),
                    nodes=[nodes[k, :min(int(nn[k]), cap)].copy() for k in range(f.size)])

    def perturb_structure(self, rot):
        """perturb_structure (src/perturbed.rs:324-460): dict of alpha, dalpha, ddalpha, beta, dbeta, ddbeta, mass_delta."""
        arrs = [np.zeros(len(self)) for _ in METRIC_FIELDS]
        ptrs = (_dp * 6)(*[_p(a) for a in arrs])
        md = np.zeros(1)
        _check(lib().sref_perturb_structure(self._h, rot, ptrs, _p(md)))
        out = dict(zip(METRIC_FIELDS, arrs))
        out["mass_delta"] = float(md[0])
        return out

    def coupling_matrices(self, ells, freqs, fields, m, metric, threads=1):
        """The six M x M matrices of perturb_deformed (src/perturbed.rs:110-263) for modes given by their
        post-processing fields [M,13,n_pts].  Returns (dict name -> [M,M] array indexed [left,right], flops per term)."""
        ells = np.ascontiguousarray(ells, dtype=np.uint64)
        f = _f64(freqs)
        fl = _f64(fields)
        M = f.size
        assert fl.shape == (M, len(POST_FIELDS), len(self))
        arrs = [_f64(metric[k]) for k in METRIC_FIELDS]
        ptrs = (_dp * 6)(*[_p(a) for a in arrs])
        out = np.zeros((len(COUPLING_MATRICES), M * M))
        flops = C.c_uint64(0)
        _check(lib().sref_coupling_matrices(self._h, M, ells.ctypes.data_as(C.POINTER(C.c_uint64)), _p(f), _p(fl), m, ptrs,
                                            _p(out), C.byref(flops), threads))
        return {k: out[i].reshape(M, M).T.copy() for i, k in enumerate(COUPLING_MATRICES)}, int(flops.value)

    def __del__(self):
        try:
            lib().sref_model_free(self._h)
        except Exception:
            pass


class Solver:
    """Mirror of ErasedSolver (reference src/dynamic_interface.rs:60-160) on the CPU oracle."""

    def __init__(self, model, ell, m, scheme, grid):
        if isinstance(scheme, str):
            scheme = SCHEMES.index(scheme)
        grid = _f64(grid)
        self.scheme = scheme
        self._model = model
        h = lib().sref_solver_new(model._h, ell, m, scheme, _p(grid), grid.size)
        if not h:
            raise RuntimeError("oracle: " + lib().sref_last_error().decode())
        self._h = C.c_void_p(h)
        self.n_int = lib().sref_solver_len(self._h)
        self.p = lib().sref_solver_points(self._h)
        self.ell, self.m = ell, m

    def system(self):
        """SoA coefficient arrays + boundary scalars handed to the HIP C-ABI."""
        k = self.n_int * self.p
        arrs = [np.empty(k) for _ in range(6)]
        bnd = np.empty(3)
        lib().sref_solver_system(self._h, *[_p(a) for a in arrs], _p(bnd))
        names = ["u", "v_gamma", "a_star", "c1", "rot", "delta_over_x"]
        out = dict(zip(names, arrs))
        out.update(inner_c1=bnd[0], inner_rot=bnd[1], outer_u=bnd[2])
        return out

    def det(self, freqs, threads=1):
        f = _f64(np.atleast_1d(freqs))
        out = np.empty(f.size)
        _check(lib().sref_det_batch(self._h, _p(f), f.size, _p(out), threads))
        return out

    def step_matrix(self, step, f):
        out = np.empty(16)
        _check(lib().sref_step_matrix(self._h, step, f, _p(out)))
        return out

    def scan_and_optimize(self, freqs, ulp, threads=1):
        f = _f64(freqs)
        cap = max(1, f.size)
        out = (Bracket * cap)()
        idx = (C.c_uint64 * cap)()
        dets = np.empty(f.size)
        n_out = C.c_size_t(0)
        inv = C.c_int(0)
        _check(lib().sref_scan_and_optimize(self._h, _p(f), f.size, ulp, out, cap, C.byref(n_out), C.byref(inv),
                                            threads, _p(dets), idx))
        n = n_out.value
        res = [dict(lo_x=b.lo_x, lo_f=b.lo_f, hi_x=b.hi_x, hi_f=b.hi_f, root=b.root, evals=b.evals, index=int(idx[i]))
               for i, b in enumerate(out[:n])]
        return res, dets, bool(inv.value)

    def optimize(self, lo_x, lo_f, hi_x, hi_f, ulp):
        b = Bracket()
        _check(lib().sref_optimize(self._h, lo_x, lo_f, hi_x, hi_f, ulp, C.byref(b)))
        return dict(lo_x=b.lo_x, lo_f=b.lo_f, hi_x=b.hi_x, hi_f=b.hi_f, root=b.root, evals=b.evals)

    def eigenvector(self, freqs, threads=1):
        f = _f64(np.atleast_1d(freqs))
        dets = np.empty(f.size)
        vec = np.empty((f.size, (self.n_int + 1) * 4))
        _check(lib().sref_eigenvector_batch(self._h, _p(f), f.size, _p(dets), _p(vec), threads))
        return dets, vec

    def count_flops(self, f):
        out = (C.c_uint64 * 3)()
        _check(lib().sref_count_flops(self._h, f, out))
        return dict(add=out[0], mul=out[1], div=out[2], total=out[0] + out[1] + out[2])

    def __del__(self):
        try:
            lib().sref_solver_free(self._h)
        except Exception:
            pass


def mat_exp(a):
    a = np.array(a, dtype=np.float64, order="F")
    n = a.shape[0]
    buf = np.ascontiguousarray(a.reshape(-1, order="F"))
    _check(lib().sref_exp(n, _p(buf)))
    return buf.reshape((n, n), order="F")


def inv4(a):
    buf = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, order="F"))
    out = np.empty(16)
    if lib().sref_inv4(_p(buf), _p(out)) != 0:
        raise ZeroDivisionError("singular")
    return out.reshape((4, 4), order="F")


def matmul4(a, b):
    x = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1, order="F"))
    y = np.ascontiguousarray(np.asarray(b, dtype=np.float64).reshape(-1, order="F"))
    out = np.empty(16)
    lib().sref_matmul4(_p(x), _p(y), _p(out))
    return out.reshape((4, 4), order="F")


def gaunt(ell, m):
    """{q_hd_n, q_hd_p, q_h_n, q_h_p} of src/postprocessing.rs:88-91."""
    out = np.empty(4)
    lib().sref_gaunt(ell, m, _p(out))
    return out


def powi(a, b):
    return lib().sref_powi(a, b)


def polytrope0_exact(gamma1, ell, radial_order):
    return lib().sref_polytrope0_exact(gamma1, ell, radial_order)
