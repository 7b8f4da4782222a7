"""GPU tests of the opt-in FMA build ("arithmetic" = 1).  Not bit-exact by construction; the
contract is BASELINE.json's north_star: discriminants within a stated relative tolerance, bracket
indices / sign changes identical, converged frequencies within 1e-10 relative.

Tolerances (stated here, used below):
  DET_TOL  = 1e-9   |det_fast - det_ref| <= DET_TOL * max(|det_ref[i-1]|, |det_ref[i]|, |det_ref[i+1]|)
                    (a pointwise relative tolerance is meaningless next to a root, where det -> 0)
  ROOT_TOL = 1e-10  relative, on converged roots
  VEC_TOL  = 1e-7   on eigenvectors, relative to the vector's max-norm
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DET_TOL, ROOT_TOL, VEC_TOL = 1e-9, 1e-10, 1e-7


def hip_from_oracle(o, arithmetic):
    import storm_amd
    h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
    h.set_option("arithmetic", arithmetic)
    return h


def local_scale(d):
    a = np.abs(d)
    return np.maximum.reduce([np.roll(a, 1), a, np.roll(a, -1)])


@pytest.mark.parametrize("scheme", ["colloc2", "colloc4", "colloc6", "colloc8", "magnus2", "magnus4", "magnus6", "magnus8"])
@pytest.mark.parametrize("ell", [0, 2])
def test_fast_det_within_tolerance_all_paths(oracle, oracle_models, scheme, ell):
    model = oracle_models("zams")
    o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
    h = hip_from_oracle(o, 1)
    f = oracle.linspace(1.0, 25.0, 400)
    ref = o.det(f, threads=16)
    scale = local_scale(ref)
    for path in (1, 2, 3, 4):
        h.set_option("path", path)
        got = h.det(f)
        assert np.all(np.abs(got - ref) <= DET_TOL * scale), (path, np.max(np.abs(got - ref) / scale))
        assert np.array_equal(np.signbit(got), np.signbit(ref)), path


@pytest.mark.parametrize("scheme,ell", [("magnus6", 1), ("colloc8", 3), ("magnus6", 2), ("colloc8", 1)])
def test_fast_scan_config2_brackets_identical_roots_close(oracle, golden_dir, scheme, ell):
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
    exact = hip_from_oracle(o, 0)   # parity build = the reference's values (bit-exact, see test_hip_parity)
    fast = hip_from_oracle(o, 1)
    f = oracle.linspace(1.0, 25.0, 10000)
    ulp = oracle.relative_to_ulp(1e-8)
    eres, edets, einv = exact.scan_and_optimize(f, ulp)
    fres, fdets, finv = fast.scan_and_optimize(f, ulp)
    assert einv == finv
    assert np.array_equal(np.signbit(edets), np.signbit(fdets))                 # sign changes bit-exact
    assert [b["index"] for b in eres] == [b["index"] for b in fres]             # bracket indices bit-exact
    assert np.all(np.abs(fdets - edets) <= DET_TOL * local_scale(edets))
    er, fr = np.array([b["root"] for b in eres]), np.array([b["root"] for b in fres])
    assert np.all(np.abs(fr - er) <= ROOT_TOL * np.abs(er)), np.max(np.abs(fr - er) / np.abs(er))
    _, evec = exact.eigenvector(er)
    _, fvec = fast.eigenvector(er)
    err = np.max(np.abs(fvec - evec), axis=1) / np.max(np.abs(evec), axis=1)
    assert np.all(err <= VEC_TOL), err.max()


def test_fast_rotation_and_inverse_grid(oracle, golden_dir):
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    model.set_rotation_constant(0.1)
    for ell, m, f in ((2, 1, oracle.linspace(1.0, 25.0, 500)), (2, 0, oracle.rev_linspace(0.5, 1.0, 500))):
        o = oracle.Solver(model, ell, m, "colloc4", model.field("r_coord"))
        fast = hip_from_oracle(o, 1)
        ref = o.det(f, threads=16)
        got = fast.det(f)
        assert np.all(np.abs(got - ref) <= DET_TOL * local_scale(ref))
        assert np.array_equal(np.signbit(got), np.signbit(ref))


def test_default_arithmetic_is_parity(oracle, oracle_models):
    import storm_amd
    model = oracle_models("zams")
    o = oracle.Solver(model, 1, 0, "colloc4", model.field("r_coord"))
    h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
    f = oracle.linspace(1.0, 9.0, 17)
    assert np.array_equal(h.det(f).view(np.uint64), o.det(f).view(np.uint64))
    with pytest.raises(storm_amd.StormError):
        h.set_option("arithmetic", 2)
