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


def hip_from_oracle(o, arithmetic=0):
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


@pytest.mark.parametrize("scheme", ["colloc4", "colloc6", "magnus4"])
def test_fast_zero_frequency_edge(oracle, golden_dir, scheme):
    """Grid points where add_frequency's omega_r^2 c1 is exactly zero -- omega = m Omega (co-rotating frequency 0) and, without
    rotation, omega = 0: the reference divides by it (src/system/adiabatic.rs:141-146) and gets inf, the FMA build's
    reciprocal must not turn that into something the sign-swap filter treats differently (a NaN always "differs" and raises
    the invalid flag, an infinity carries a sign): same finite pattern, same brackets, same flag as the parity build."""
    import storm_amd
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    for rot, ell, m, edge in ((0.1, 2, 1, 0.1), (0.0, 2, 0, 0.0), (0.25, 3, 2, 0.5)):
        model.set_rotation_constant(rot)
        o = oracle.Solver(model, ell, m, scheme, model.field("r_coord"))
        f = np.sort(np.concatenate([oracle.linspace(1.0, 8.0, 37), [edge, edge + 0.25, edge + 0.5]]))
        out = {}
        for arith in (0, 1):
            h = hip_from_oracle(o, arithmetic=arith)
            try:
                res, dets, inv = h.scan_and_optimize(f, oracle.relative_to_ulp(1e-8))
            except storm_amd.StormError as e:      # (a non-finite value inside a refinement: the reference asserts)
                res, dets, inv = e.code, h.det(f), None
            out[arith] = (res, dets, inv)
        (pres, pdets, pinv), (fres, fdets, finv) = out[0], out[1]
        assert not np.isfinite(pdets[np.searchsorted(f, edge)])                 # the edge is what it is meant to be
        assert np.array_equal(np.isfinite(pdets), np.isfinite(fdets)), (rot, ell, m)
        assert np.array_equal(np.isnan(pdets), np.isnan(fdets)), (rot, ell, m)
        fin = np.isfinite(pdets)
        assert np.array_equal(np.signbit(pdets[~np.isnan(pdets)]), np.signbit(fdets[~np.isnan(pdets)]))
        assert np.all(np.abs(fdets[fin] - pdets[fin]) <= DET_TOL * local_scale(np.where(fin, pdets, 0.0))[fin])
        assert pinv == finv
        if isinstance(pres, list):
            assert [b["index"] for b in pres] == [b["index"] for b in fres]
        else:
            assert pres == fres


@pytest.mark.parametrize("scheme", ["colloc2", "colloc4", "colloc6", "colloc8"])
def test_fast_collocation_on_the_largest_fixture(oracle, oracle_models, scheme):
    # joel: the finest grid of the fixtures; the collocation schemes of the FMA build solve their stage systems by LU with
    # the adjugate inverse as the fall-back next to the centre (mat4.hpp: lu4_factor / m4_solve_n), both are walked here
    # On this grid the discriminant of ell = 3 is ill-conditioned beyond the contract: ONE ulp of the frequency moves the
    # reference's own value by up to 1e-2 of the local scale (ell = 1: 5e-11), so no re-association can stay within 1e-9
    # there.  The bar is therefore the larger of DET_TOL and 32 x that one-ulp sensitivity, measured with the oracle
    # (19 204 intervals each round their matrices: measured 1.4e-9 = 27 one-ulp steps at ell = 1, 1.7e-3 at ell = 3).
    model = oracle_models("joel")
    for ell in (1, 3):
        o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
        h = hip_from_oracle(o, 1)
        f = oracle.linspace(1.0, 25.0, 96)
        ref = o.det(f, threads=16)
        scale = local_scale(ref)
        one_ulp = np.max(np.abs(o.det(np.nextafter(f, 100.0), threads=16) - ref) / scale)
        tol = max(DET_TOL, 32.0 * one_ulp)
        for path in (2, 4):
            h.set_option("path", path)
            got = h.det(f)
            assert np.all(np.abs(got - ref) <= tol * scale), (ell, path, np.max(np.abs(got - ref) / scale), one_ulp)
            clear = np.abs(ref) > 2.0 * tol * scale
            assert np.array_equal(np.signbit(got[clear]), np.signbit(ref[clear])), (ell, path)


@pytest.mark.parametrize("scheme", ["colloc4", "colloc6", "magnus4"])
def test_fast_differential_rotation(oracle, golden_dir, scheme):
    # non-uniform rotation: the frequency terms (and, in the FMA build, their reciprocal) are evaluated per point
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_zams.npz"))
    x = model.field("r_coord")
    model.set_field("rot", 0.05 + 0.1 * np.exp(-3.0 * x) - 0.02 * x)
    o = oracle.Solver(model, 2, 1, scheme, x)
    h = hip_from_oracle(o, 1)
    f = oracle.linspace(1.0, 20.0, 200)
    ref = o.det(f, threads=16)
    got = h.det(f)
    assert np.all(np.abs(got - ref) <= DET_TOL * local_scale(ref)), np.max(np.abs(got - ref) / local_scale(ref))
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
