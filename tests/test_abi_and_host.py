"""CPU-only checks: the C-ABI library loads and exports every symbol include/storm_hip.h
declares, argument validation works without a GPU, and the C++ host mirror's model layer
agrees bit-for-bit with the oracle's.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "storm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(storm_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from storm_amd import ffi
    L = ffi.load_library()
    declared = _declared_functions()
    assert declared, "no declarations parsed"
    assert sorted(ffi.EXPORTS) == declared
    for name in declared:
        assert hasattr(L, name), f"libstorm_hip.so does not export {name}"
    assert b"gfx950" in L.storm_version()


def test_argument_validation_without_gpu():
    from storm_amd import ffi
    L = ffi.load_library()
    h = C.c_void_p()
    assert L.storm_solver_create(None, C.byref(h)) == -1
    d = ffi.SystemDesc()
    d.scheme = 99
    assert L.storm_solver_create(C.byref(d), C.byref(h)) == -1
    assert b"scheme" in L.storm_last_error()
    d.scheme = 1
    d.p = 3  # colloc4 has 2 points
    assert L.storm_solver_create(C.byref(d), C.byref(h)) == -1
    L.storm_solver_destroy(None)  # must be a no-op
    assert L.storm_det_batch(None, None, 0, None) == -1


def test_postproc_argument_validation_without_gpu():
    from storm_amd import ffi
    L = ffi.load_library()
    h = C.c_void_p()
    assert L.storm_postproc_create(None, C.byref(h)) == -1
    d = ffi.ModelDesc()
    d.n_pts = 4
    assert L.storm_postproc_create(C.byref(d), C.byref(h)) == -1     # null arrays
    a = np.array([0.1, 0.2, 0.3, 0.4])
    p = a.ctypes.data_as(C.POINTER(C.c_double))
    d = ffi.ModelDesc(4, *([p] * 10))
    assert L.storm_postproc_create(C.byref(d), C.byref(h)) == -1     # r_coord[0] != 0
    assert b"r_coord[0]" in L.storm_last_error()
    L.storm_postproc_destroy(None)
    assert L.storm_modes_batch(None, None, None, 0, None, None, None, None, None, 0) == -1


def test_no_cpu_fallback_when_no_gpu():
    """Without a HIP device creation must fail with STORM_ERR_HIP, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from storm_amd import ffi
    L = ffi.load_library()
    one = np.ones(1)
    p = one.ctypes.data_as(C.POINTER(C.c_double))
    d = ffi.SystemDesc(0, 0, 0, 1, 1, p, p, p, p, p, p, 1.0, 0.0, 3.0)
    h = C.c_void_p()
    assert L.storm_solver_create(C.byref(d), C.byref(h)) == -2
    assert not h.value
    r = np.array([0.0, 0.5, 1.0])
    md = ffi.ModelDesc(3, *([r.ctypes.data_as(C.POINTER(C.c_double))] * 10))
    assert L.storm_postproc_create(C.byref(md), C.byref(h)) == -2
    assert not h.value


def test_sweep_argument_validation_without_gpu():
    from storm_amd import ffi
    L = ffi.load_library()
    h = C.c_void_p()
    assert L.storm_sweep_create(None, 1, C.byref(h)) == -1
    d = (ffi.SystemDesc * 2)()
    assert L.storm_sweep_create(d, 0, C.byref(h)) == -1
    assert b"at least one" in L.storm_last_error()
    one = np.ones(2)
    p = one.ctypes.data_as(C.POINTER(C.c_double))
    d[0] = ffi.SystemDesc(1, 0, 0, 1, 2, p, p, p, p, p, p, 1.0, 0.0, 3.0)
    d[1] = ffi.SystemDesc(0, 0, 0, 1, 1, p, p, p, p, p, p, 1.0, 0.0, 3.0)   # another scheme
    assert L.storm_sweep_create(d, 2, C.byref(h)) == -1
    assert b"share the scheme" in L.storm_last_error()
    d[1] = ffi.SystemDesc(1, 2, 0, 2, 2, p, p, p, p, p, p, 1.0, 0.0, 3.0)   # another grid
    assert L.storm_sweep_create(d, 2, C.byref(h)) == -1
    L.storm_sweep_destroy(None)  # no-op
    assert L.storm_sweep_size(None) == 0
    assert L.storm_sweep_det(None, None, None, None) == -1
    assert L.storm_sweep_scan_and_optimize(None, None, None, 1, None, 0, None, None, None, None) == -1
    assert L.storm_sweep_modes(None, None, None, None, None, None, None, None, None, 0) == -1
    import torch
    if not torch.cuda.is_available():   # no CPU fallback
        d[1] = ffi.SystemDesc(1, 2, 0, 1, 2, p, p, p, p, p, p, 1.0, 0.0, 3.0)
        assert L.storm_sweep_create(d, 2, C.byref(h)) == -2
        assert not h.value


def test_precision_conversion():
    from storm_amd import ffi
    assert ffi.relative_to_ulp(1e-8) == 45035996   # CLI default, reference src/bracket.rs:30-40
    assert ffi.relative_to_ulp(0.0) == 1
    assert ffi.relative_to_ulp(1e-300) == 1


@pytest.mark.parametrize("name", ["zams", "tams"])
def test_host_model_matches_oracle(oracle, oracle_models, golden_dir, name):
    from storm_amd import host
    hm = host.Model.from_gsm_npz(os.path.join(golden_dir, f"gsm_{name}.npz"))
    om = oracle_models(name)
    for f in host.Model.FIELDS:
        a, b = hm.field(f), om.field(f)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), f
    for scale in (1, 3):
        assert np.array_equal(hm.grid(scale), om.grid(scale))
    g = hm.grid(2)   # the model on a resampled solver grid, as post-process evaluates it (src/main.rs:892)
    he, oe = hm.eval(g), om.eval(g)
    for f in host.Model.FIELDS:
        assert np.array_equal(he.field(f).view(np.uint64), oe.field(f).view(np.uint64)), f


def test_host_polytrope_matches_oracle(oracle):
    from storm_amd import host
    hm = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 0.01)
    om = oracle.Model.polytrope(3.0, 5.0 / 3.0, 0.01)
    assert len(hm) == len(om) and len(hm) > 600
    for f in host.Model.FIELDS:
        assert np.array_equal(hm.field(f).view(np.uint64), om.field(f).view(np.uint64)), f
    assert host.polytrope0_exact(5.0 / 3.0, 0, 1) == oracle.polytrope0_exact(5.0 / 3.0, 0, 1)
    assert np.array_equal(host.linspace(1.0, 25.0, 1000), oracle.linspace(1.0, 25.0, 1000))
    assert np.array_equal(host.rev_linspace(0.5, 1.0, 1000), oracle.rev_linspace(0.5, 1.0, 1000))
    p0 = host.Model.Polytrope0(5.0 / 3.0)
    assert np.array_equal(p0.grid(100), oracle.Model.poly0(5.0 / 3.0).grid(100))


@pytest.mark.parametrize("name", ["zams", "tams"])
def test_host_perturb_structure_matches_oracle(oracle, oracle_models, golden_dir, name):
    # `deform` (src/main.rs:840-876 -> perturb_structure, src/perturbed.rs:324-460) stays on the host; the C++ mirror
    # and the oracle restate it independently and must agree bit for bit (parity unpinned: no reference test)
    from storm_amd import host
    hm = host.Model.from_gsm_npz(os.path.join(golden_dir, f"gsm_{name}.npz"))
    om = oracle_models(name)
    for rot in (0.01, 0.1):
        a, b = hm.perturb_structure(rot), om.perturb_structure(rot)
        for k in ("alpha", "dalpha", "ddalpha", "beta", "dbeta", "ddbeta"):
            assert np.array_equal(a[k].view(np.uint64), b[k].view(np.uint64)), k
            assert np.all(np.isfinite(a[k])) and a[k][0] == 0.0
        assert a["mass_delta"] == b["mass_delta"]
        assert np.all(a["alpha"][1:] > 0) and np.all(a["beta"][1:] < 0)     # P0 expands, P2 flattens the poles
    a1, a2 = hm.perturb_structure(0.01), hm.perturb_structure(0.02)
    np.testing.assert_allclose(a2["alpha"], 4.0 * a1["alpha"], rtol=1e-13)   # quadratic in the rotation rate


def test_shard_bounds_of_the_library_cover_every_pair_once():
    """storm_shard_bounds (the split storm_multi uses; pure index arithmetic, runs without a GPU) equals the split of
    storm_amd/distributed.py, and with one halo point every adjacent pair of grid points belongs to exactly one part."""
    from storm_amd import ffi, distributed as D
    for n in (0, 1, 2, 7, 64, 1000, 100001):
        for parts in (1, 2, 3, 8, 13):
            pairs = []
            for k in range(parts):
                lo, hi = ffi.shard_bounds(n, parts, k)
                assert (lo, hi) == D.shard_bounds(n, parts, k)
                assert 0 <= lo <= hi <= n
                if n <= 1000:
                    pairs += [(i, i + 1) for i in range(lo, min(hi + 1, n) - 1)]
            if n <= 1000:
                assert pairs == [(i, i + 1) for i in range(max(0, n - 1))]
    # the deal of the brackets: cyclic by global index, every bracket exactly once
    for n_br in (0, 1, 5, 22, 163):
        for parts in (1, 2, 3, 8):
            dealt = sorted(sum((D.deal(n_br, parts, k) for k in range(parts)), []))
            assert dealt == list(range(n_br))
