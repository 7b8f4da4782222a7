"""GPU tests of the FMA build's ordered product on v_mfma_f64_4x4x4 (storm_amd/csrc/scheme_kernels.hpp: hex_uses_mfma,
col_uses_mfma): the 16-lane and 32-lane bodies of det_hex, the column-owner body and det_mix, each forced through the
`hex_lanes` option, against the oracle (= the reference's arithmetic, src/solver.rs:486-492 for the product) to the fast
build's tolerance contract (tests/test_hip_fast_mode.py: |det_fast - det_ref| <= 1e-9 of the local scale, identical signs).
The grid has 3033 intervals -- not a multiple of 4, 16 or 32, so the last, partial round of every body is walked -- and the
batch sizes are not multiples of the frequencies per wave (4, 2, 16).  The parity build never takes the MFMA (fused sums):
its bodies stay bit-identical to the oracle (tests/test_hip_parity.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DET_TOL = 1e-9
SCHEMES = ["colloc2", "colloc4", "colloc6", "colloc8", "magnus2", "magnus4", "magnus6", "magnus8"]


def local_scale(d):
    a = np.abs(d)
    return np.maximum.reduce([np.roll(a, 1), a, np.roll(a, -1)])


def fast_solver(o):
    import storm_amd
    h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
    h.set_option("arithmetic", 1)
    h.set_option("hex_rounds", 0)
    return h


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("ell", [0, 2])
def test_every_body_within_tolerance_of_the_oracle(oracle, oracle_models, scheme, ell):
    model = oracle_models("zams")
    o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
    assert o.n_int % 4 != 0
    h = fast_solver(o)
    f = oracle.linspace(1.0, 25.0, 777)
    ref = o.det(f, threads=16)
    scale = local_scale(ref)
    # 16 / 32 lanes per frequency (det_hex), 4 = column owners alone, -80 = det_mix with 80 % column owners
    for lanes in (16, 32, 4, -80):
        h.set_option("hex_lanes", lanes)
        got = h.det(f)
        assert np.all(np.abs(got - ref) <= DET_TOL * scale), (lanes, np.max(np.abs(got - ref) / scale))
        assert np.array_equal(np.signbit(got), np.signbit(ref)), lanes


@pytest.mark.parametrize("scheme", ["colloc4", "magnus2", "colloc8"])
def test_a_nan_frequency_stays_in_its_own_mfma_block(oracle, oracle_models, scheme):
    """One instruction multiplies the products of four frequencies (one per MFMA block): a frequency that is NaN must
    come out NaN and leave its wave's other frequencies exactly as they are without it."""
    model = oracle_models("zams")
    o = oracle.Solver(model, 2, 0, scheme, model.field("r_coord"))
    h = fast_solver(o)
    f = oracle.linspace(2.0, 20.0, 613)
    poisoned = f.copy()
    where = np.array([0, 5, 6, 63, 64, 130, 257, 612])
    poisoned[where] = np.nan
    for lanes in (16, 32, 4, -80):
        h.set_option("hex_lanes", lanes)
        clean = h.det(f)
        got = h.det(poisoned)
        assert np.all(np.isnan(got[where])), lanes
        keep = np.ones(f.size, dtype=bool)
        keep[where] = False
        assert np.array_equal(got[keep].view(np.uint64), clean[keep].view(np.uint64)), lanes


def test_scan_roots_and_brackets_through_the_mfma_bodies(oracle, oracle_models):
    """scan_and_optimize with each body forced: bracket indices identical to the oracle's, roots within 1e-10."""
    model = oracle_models("zams")
    o = oracle.Solver(model, 1, 0, "colloc4", model.field("r_coord"))
    f = oracle.linspace(1.0, 25.0, 1500)
    ulp = oracle.relative_to_ulp(1e-8)
    want, _, _ = o.scan_and_optimize(f, ulp, threads=16)
    h = fast_solver(o)
    for lanes in (16, 32, 4, -80, 0):
        h.set_option("hex_lanes", lanes)
        got, _, _ = h.scan_and_optimize(f, ulp)
        assert [b["index"] for b in got] == [b["index"] for b in want], lanes
        gr, wr = np.array([b["root"] for b in got]), np.array([b["root"] for b in want])
        assert np.all(np.abs(gr - wr) <= 1e-10 * np.abs(wr)), (lanes, np.max(np.abs(gr - wr) / np.abs(wr)))


@pytest.mark.parametrize("scheme", ["colloc4", "magnus2", "magnus8"])
def test_tiny_and_ragged_grids(oracle, oracle_models, scheme):
    """Grids of 1 .. 49 intervals (a stretch of the zams grid): fewer intervals than one round holds, exactly one round,
    one more than a round -- every guard of the conditional MFMA chain -- for each body, against the oracle."""
    model = oracle_models("zams")
    full = model.field("r_coord")
    f = oracle.linspace(1.5, 12.0, 531)
    for npts in (2, 4, 5, 6, 16, 17, 18, 33, 34, 50):
        o = oracle.Solver(model, 2, 0, scheme, np.ascontiguousarray(full[800:800 + npts]))
        assert o.n_int == npts - 1
        ref = o.det(f, threads=16)
        assert np.all(np.isfinite(ref)), npts
        scale = local_scale(ref)
        h = fast_solver(o)
        for lanes in (16, 32, 4, -80):
            h.set_option("hex_lanes", lanes)
            got = h.det(f)
            assert np.all(np.abs(got - ref) <= DET_TOL * scale), (npts, lanes, np.max(np.abs(got - ref) / scale))
            assert np.array_equal(np.signbit(got), np.signbit(ref)), (npts, lanes)
