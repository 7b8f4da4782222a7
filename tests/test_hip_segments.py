"""GPU tests of the segment-parallel path of the FMA build (option "segments", include/storm_hip.h): small batches cut the
interval axis into segments that are evaluated in parallel.  It re-associates the chain of 4x4 products, so it exists in
the FMA build only and is held to that build's contract (tests/test_hip_fast_mode.py): discriminants within 1e-9 of the
local scale with identical signs, roots within 1e-10, eigenvectors within 1e-7 of their max-norm -- all against the
ORACLE.  The parity build must ignore the option and stay bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DET_TOL, ROOT_TOL, VEC_TOL = 1e-9, 1e-10, 1e-7
SCHEMES = ["colloc2", "colloc4", "colloc6", "colloc8", "magnus2", "magnus4", "magnus6", "magnus8"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def hip_from_oracle(o, arithmetic=1):
    import storm_amd
    h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
    h.set_option("arithmetic", arithmetic)
    h.set_option("profile", 1)
    return h


def local_scale(d):
    a = np.abs(d)
    return np.maximum.reduce([np.roll(a, 1), a, np.roll(a, -1)])


def used(h, kind):
    return h.kernel_times(reset=True)[kind][1]


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("ell", [0, 2])
def test_segmented_discriminants_within_tolerance(oracle, oracle_models, scheme, ell):
    model = oracle_models("zams")   # 3033 intervals: not a multiple of any segment length
    o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
    h = hip_from_oracle(o)
    f = oracle.linspace(1.0, 25.0, 300)
    ref = o.det(f, threads=16)
    scale = local_scale(ref)
    for seg in (0, 16, 48, 1000, 4096):       # auto, shortest, rounded up to 64, a few long ones, one segment (> n_int)
        h.set_option("segments", seg)
        for n in (300, 37, 1):
            used(h, "segments")
            got = h.det(f[:n])
            assert used(h, "segments") == 1, (seg, n)
            assert np.all(np.abs(got - ref[:n]) <= DET_TOL * scale[:n]), (seg, n, np.max(np.abs(got - ref[:n]) / scale[:n]))
            assert np.array_equal(np.signbit(got), np.signbit(ref[:n])), (seg, n)
    h.set_option("segments", -1)              # off: the ordered-product kernels
    got = h.det(f)
    assert used(h, "segments") == 0 and np.all(np.abs(got - ref) <= DET_TOL * scale)


def test_parity_build_ignores_segments(oracle, oracle_models):
    model = oracle_models("zams")
    o = oracle.Solver(model, 2, 0, "colloc4", model.field("r_coord"))
    h = hip_from_oracle(o, arithmetic=0)
    h.set_option("segments", 64)
    f = oracle.linspace(1.0, 25.0, 40)
    assert np.array_equal(bits(h.det(f)), bits(o.det(f, threads=16)))
    assert used(h, "segments") == 0
    rdet, rvec = o.eigenvector(f[:5], threads=16)
    gdet, gvec = h.eigenvector(f[:5])
    assert np.array_equal(bits(rvec), bits(gvec)) and np.array_equal(bits(rdet), bits(gdet))


def test_tiny_grids_and_rotation(oracle, golden_dir):
    # fewer than 64 intervals: the segment path does not apply; differential rotation: per-point frequency terms
    model = oracle.Model.poly0(5.0 / 3.0)
    o = oracle.Solver(model, 2, 0, "colloc4", oracle.linspace(0.0, 1.0, 40))
    h = hip_from_oracle(o)
    f = oracle.linspace(1.0, 6.0, 9)
    ref = o.det(f)
    assert np.all(np.abs(h.det(f) - ref) <= DET_TOL * local_scale(ref)) and used(h, "segments") == 0
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_zams.npz"))
    x = model.field("r_coord")
    model.set_field("rot", 0.05 + 0.1 * np.exp(-3.0 * x) - 0.02 * x)
    for scheme in ("colloc4", "magnus4"):
        o = oracle.Solver(model, 2, 1, scheme, x)
        h = hip_from_oracle(o)
        f = oracle.linspace(1.0, 20.0, 33)
        ref = o.det(f, threads=16)
        got = h.det(f)
        assert used(h, "segments") == 1
        assert np.all(np.abs(got - ref) <= DET_TOL * local_scale(ref)) and np.array_equal(np.signbit(got), np.signbit(ref))
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    model.set_rotation_constant(0.1)
    o = oracle.Solver(model, 3, -1, "colloc6", model.field("r_coord"))
    h = hip_from_oracle(o)
    f = oracle.linspace(1.0, 25.0, 50)
    ref = o.det(f, threads=16)
    got = h.det(f)
    assert np.all(np.abs(got - ref) <= DET_TOL * local_scale(ref)) and np.array_equal(np.signbit(got), np.signbit(ref))


@pytest.mark.parametrize("scheme,ell", [("colloc4", 1), ("colloc4", 2), ("colloc4", 3), ("magnus6", 2), ("colloc8", 3), ("colloc2", 0)])
def test_segmented_refinement_and_eigenvectors_tams(oracle, golden_dir, scheme, ell):
    """A whole scan of the TAMS model in the FMA build with the refinement rounds and the eigenvectors on the segment
    path: brackets identical, roots within 1e-10, every eigenvector within 1e-7 of its max-norm of the ORACLE's."""
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
    h = hip_from_oracle(o)
    f = oracle.linspace(1.0, 25.0, 2000)
    ulp = oracle.relative_to_ulp(1e-8)
    rres, rdets, _ = o.scan_and_optimize(f, ulp, threads=16)
    gres, gdets, _ = h.scan_and_optimize(f, ulp)
    assert used(h, "segments") >= 2 and len(rres) > 10
    assert [r["index"] for r in rres] == [g["index"] for g in gres]
    rr, gr = np.array([r["root"] for r in rres]), np.array([g["root"] for g in gres])
    assert np.all(np.abs(gr - rr) <= ROOT_TOL * np.abs(rr)), np.max(np.abs(gr - rr) / np.abs(rr))
    for g in gres:
        assert np.signbit(g["lo_f"]) != np.signbit(g["hi_f"]) and g["lo_x"] <= g["root"] <= g["hi_x"]
    _, rvec = o.eigenvector(rr, threads=16)
    for seg in (0, 32, 256):
        h.set_option("segments", seg)
        h.stats(reset=True)
        _, gvec = h.eigenvector(rr)
        err = np.max(np.abs(gvec - rvec), axis=1) / np.max(np.abs(rvec), axis=1)
        assert np.all(err <= VEC_TOL), (seg, err.max(), int(np.argmax(err)), h.stats()["segment_fallbacks"])
    # the guard: with the bound at 2^0 every mode goes through the sequential elimination -- the FMA build's own
    h.set_option("seg_growth_log2", 0)
    h.stats(reset=True)
    _, fb = h.eigenvector(rr)
    assert h.stats()["segment_fallbacks"] == len(rr)
    h.set_option("segments", -1)
    _, classic = h.eigenvector(rr)
    assert np.array_equal(bits(fb), bits(classic))


def test_segmented_eigenvectors_config3_polytrope(oracle):
    model = oracle.Model.polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
    grid = model.grid(1)
    o = oracle.Solver(model, 2, 0, "colloc4", grid)
    h = hip_from_oracle(o)
    f = oracle.linspace(1.0, 25.0, 2000)
    ulp = oracle.relative_to_ulp(1e-8)
    rres, _, _ = o.scan_and_optimize(f, ulp, threads=16)
    gres, _, _ = h.scan_and_optimize(f, ulp)
    rr, gr = np.array([r["root"] for r in rres]), np.array([g["root"] for g in gres])
    assert len(rr) == 22 and np.all(np.abs(gr - rr) <= ROOT_TOL * np.abs(rr))
    _, rvec = o.eigenvector(rr, threads=16)
    h.stats(reset=True)
    _, gvec = h.eigenvector(rr)
    err = np.max(np.abs(gvec - rvec), axis=1) / np.max(np.abs(rvec), axis=1)
    # 2e-6 on THIS model, like its discriminants (tests/test_hip_full_size.py): the Colloc4 stage matrices of the outermost
    # intervals (V -> inf) amplify any rounding difference, and the eigenvector is normalised at the surface
    assert np.all(err <= 2e-6), (err.max(), h.stats()["segment_fallbacks"])
    h.set_option("segments", -1)
    _, cvec = h.eigenvector(rr)   # the FMA build's sequential elimination: the same order of magnitude
    cerr = np.max(np.abs(cvec - rvec), axis=1) / np.max(np.abs(rvec), axis=1)
    assert np.all(cerr <= 2e-6) and err.max() <= 20 * max(cerr.max(), 1e-9), (err.max(), cerr.max())


def test_mode_ids_through_segmented_eigenvectors(oracle, golden_dir):
    """eigenvectors (segment path) + device post-processing: the mode ids of the oracle."""
    import storm_amd
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    grid = model.field("r_coord")
    o = oracle.Solver(model, 2, 0, "colloc4", grid)
    h = hip_from_oracle(o)
    ulp = oracle.relative_to_ulp(1e-8)
    rres, _, _ = o.scan_and_optimize(oracle.linspace(1.0, 25.0, 1500), ulp, threads=16)
    rr = np.array([r["root"] for r in rres])
    _, rvec = o.eigenvector(rr, threads=16)
    ref = model.postprocess(2, 0, rr, rvec, threads=16)
    post = storm_amd.HipPostproc({k: model.field(k) for k in storm_amd.MODEL_FIELDS})
    got = h.modes(post, rr, want_fields=True)
    assert used(h, "segments") >= 2
    assert got["ids"].tolist() == ref["ids"].tolist()
    scale = np.max(np.abs(ref["fields"]), axis=2, keepdims=True)
    assert np.all(np.abs(got["fields"] - ref["fields"]) <= 1e-6 * np.maximum(scale, 1e-300))
