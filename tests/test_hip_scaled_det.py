"""GPU tests of storm_det_scaled_batch: the discriminant as mantissa * 2^exponent from a running product that is renormalised
by exact powers of two (north_star's "running log-determinant"; the reference keeps the raw product, src/solver.rs:504,531,541,
and only detects overflow afterwards, src/dynamic_interface.rs:130-137).  Where the raw value is finite the scaled one must
reproduce it to rounding, signs included; where the raw product overflows it must still be the determinant of the same
boundary-value matrix (checked against numpy's slogdet on a system whose step matrix is known in closed form)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scheme,ell", [("colloc4", 2), ("magnus4", 1), ("colloc2", 0), ("colloc8", 3)])
def test_scaled_equals_raw_where_raw_is_finite(oracle, oracle_models, scheme, ell):
    import storm_amd
    model = oracle_models("tams")
    o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
    f = oracle.linspace(1.0, 25.0, 173)
    ref = o.det(f, threads=16)
    for arith in (0, 1):
        h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
        h.set_option("arithmetic", arith)
        m, e = h.det_scaled(f)
        assert np.all((np.abs(m) >= 0.5) & (np.abs(m) < 1.0))
        got = np.ldexp(m, e.astype(np.int64))
        scale = np.maximum.reduce([np.roll(np.abs(ref), 1), np.abs(ref), np.roll(np.abs(ref), -1)])
        assert np.all(np.abs(got - ref) <= 1e-9 * scale), np.max(np.abs(got - ref) / scale)
        assert np.array_equal(np.signbit(got), np.signbit(ref))
        h.set_option("steps_bytes_max", 5 * o.n_int * 128)   # chunks of 5 frequencies: same values
        m2, e2 = h.det_scaled(f)
        assert np.array_equal(m2.view(np.uint64), m.view(np.uint64)) and np.array_equal(e2, e)


def uniform_colloc2_reference(n_int, dx, u, vg, a_s, c1, f, outer_u):
    """log2 |D| and sign of D for a system whose every interval has the same coefficients (ell = 0, Colloc2), in numpy:
    L = -(J - I)^-1 (J + I), J = A/2; the product L^n_int with power-of-two renormalisation; then slogdet of the 8 x 8
    boundary-value matrix [Bi 0; acc' -I; 0 Bo] (+ 2K for the two outer rows that carry the scale, see common_kernels.hip)."""
    A = np.zeros((4, 4))
    A[0, 0] = dx * (vg - 1.0); A[0, 1] = -dx * vg
    A[1, 0] = -dx * a_s + dx * c1 * f * f; A[1, 1] = dx * (a_s - u + 3.0)
    A[2, 2] = dx * (3.0 - u); A[3, 0] = dx * u * a_s; A[3, 1] = dx * u * vg; A[3, 2] = 0.0; A[3, 3] = dx * (-u + 2.0)
    A[1, 3] = -dx; A[2, 3] = dx
    J = A / 2.0
    L = -np.linalg.solve(J - np.eye(4), J + np.eye(4))
    acc, K = np.eye(4), 0
    for _ in range(n_int):
        acc = L @ acc
        k = int(np.floor(np.log2(np.max(np.abs(acc)))))
        acc, K = np.ldexp(acc, -k), K + k
    M = np.zeros((8, 8))
    M[0, :4] = [c1 * f * f, 0.0, 0.0, 0.0]      # inner boundary, ell = 0 (src/system/adiabatic.rs:154-183)
    M[1, :4] = [0.0, 0.0, 0.0, -1.0]
    M[2:6, :4] = acc; M[2:6, 4:] = -np.eye(4)
    M[6, 4:] = [1.0, -1.0, 0.0, 0.0]            # outer boundary (:185-208)
    M[7, 4:] = [outer_u, 0.0, 1.0, 1.0]
    sign, logabs = np.linalg.slogdet(M)
    return sign, logabs / np.log(2.0) + 2 * K, bool(np.all(np.max(np.abs(acc), axis=0) > 1e-200))


def test_scaled_survives_overflow_of_the_raw_product():
    import storm_amd
    # (synthetic coefficients: every solution grows, by factors 1.37..1.69 per interval; the product overflows after ~1400
    # intervals.  Frequencies at which one column of the product underflows to exactly zero next to the others are skipped:
    # there the reference's elimination divides by a zero pivot as well)
    n_int, dx, u, vg, a_s, c1, outer_u = 3000, 0.02, -18.0, 21.0, 0.0, 1.0, 3.0
    sys_ = dict(u=np.full(n_int, u), v_gamma=np.full(n_int, vg), a_star=np.full(n_int, a_s), c1=np.full(n_int, c1),
                rot=np.zeros(n_int), delta_over_x=np.full(n_int, dx), inner_c1=c1, inner_rot=0.0, outer_u=outer_u)
    h = storm_amd.HipSolver("colloc2", 0, 0, sys_, n_int, 1)
    f = np.array([3.3, 3.7, 4.1, 4.6, 5.2, 6.0])
    raw = h.det(f)
    assert not np.any(np.isfinite(raw))         # ~1.6^3000: far beyond f64
    m, e = h.det_scaled(f)
    checked = 0
    for k in range(f.size):
        sign, log2abs, regular = uniform_colloc2_reference(n_int, dx, u, vg, a_s, c1, f[k], outer_u)
        if not regular:
            continue
        checked += 1
        assert np.isfinite(m[k]) and 0.5 <= abs(m[k]) < 1.0 and e[k] > 2000, (k, m[k], e[k])
        assert np.sign(m[k]) == sign, k
        assert abs((np.log2(abs(m[k])) + e[k]) - log2abs) <= 1e-6 * abs(log2abs), (k, np.log2(abs(m[k])) + e[k], log2abs)
    assert checked >= 3
    # a short version of the same system, where the raw product is finite: both outputs agree
    short = {k: (v[:200] if isinstance(v, np.ndarray) else v) for k, v in sys_.items()}
    hs = storm_amd.HipSolver("colloc2", 0, 0, short, 200, 1)
    raw, (m, e) = hs.det(f), hs.det_scaled(f)
    assert np.all(np.isfinite(raw)) and np.allclose(np.ldexp(m, e), raw, rtol=1e-10, atol=0)


# ---------------------------------------------------------------------------------------------- option "scaled_scan"
@pytest.mark.parametrize("scheme,ell,arith", [("colloc4", 2, 0), ("colloc4", 2, 1), ("magnus4", 1, 0), ("colloc8", 3, 1)])
def test_scaled_scan_finds_the_raw_scans_modes_where_the_raw_product_is_finite(oracle, oracle_models, scheme, ell, arith):
    """Option scaled_scan = 1: storm_scan_and_optimize brackets on mantissa * 2^exponent.  Where the raw product is finite the
    scaled value is the raw one up to the rounding of the closing elimination, and InverseQuadratic's proposals are invariant
    under the exact common scaling: same bracket index list as the raw scan, roots within 1e-10 (the FMA build's own root
    contract; measured below 1e-13), the same again through the two-call form (storm_scan_brackets, storm_optimize_brackets)."""
    import storm_amd
    model = oracle_models("tams")
    o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
    f = oracle.linspace(1.0, 25.0, 1500)
    ulp = storm_amd.relative_to_ulp(1e-10)
    h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
    h.set_option("arithmetic", arith)
    raw, raw_dets, raw_inv = h.scan_and_optimize(f, ulp)
    h.set_option("scaled_scan", 1)
    got, dets, inv = h.scan_and_optimize(f, ulp)
    assert len(raw) > 10 and [b["index"] for b in got] == [b["index"] for b in raw] and inv == raw_inv
    r0, r1 = np.array([b["root"] for b in raw]), np.array([b["root"] for b in got])
    assert np.max(np.abs(r1 - r0) / r0) <= 1e-10, np.max(np.abs(r1 - r0) / r0)
    scale = np.maximum.reduce([np.roll(np.abs(raw_dets), 1), np.abs(raw_dets), np.roll(np.abs(raw_dets), -1)])
    assert np.all(np.abs(dets - raw_dets) <= 1e-9 * scale)              # det_out: the raw value where it exists
    for a, b in zip(raw, got):
        assert abs(b["evals"] - a["evals"]) <= 2 and np.sign(b["lo_f"]) == np.sign(a["lo_f"]) and np.isfinite(b["lo_f"])
    found, _, _ = h.scan_brackets(f)
    assert [b["index"] for b in found] == [b["index"] for b in raw] and all(np.isnan(b["root"]) for b in found)
    two = h.optimize_brackets(found, ulp)
    assert np.array_equal(np.array([b["root"] for b in two]).view(np.uint64), r1.view(np.uint64))
    # a bracket whose signs are not the discriminant's: rejected
    wrong = [dict(found[0], lo_f=-found[0]["lo_f"], hi_f=-found[0]["hi_f"])]
    with pytest.raises(storm_amd.StormError):
        h.optimize_brackets(wrong, ulp)


@pytest.mark.parametrize("arith,segments", [(0, 0), (1, 0), (1, 1488)])
def test_scaled_scan_brackets_where_the_raw_product_overflows(arith, segments):
    """(arith 1: the FMA build takes the segment path -- transfer matrices of 128-interval segments from the fast kernels,
    then the scaled chain over the segments; segments = 1488 makes one segment of nearly the whole grid, whose product
    overflows: every frequency must then come back through the staged path with the same answers.)
    The synthetic system of test_scaled_survives_overflow_of_the_raw_product: every raw discriminant is +-inf, the raw
    scan drops every pair (the reference: `invalid numbers`, src/dynamic_interface.rs:130-137) and finds nothing.  With
    scaled_scan the sign changes of the numpy restatement are found at the same grid indices and the refined roots sit
    where that restatement changes sign."""
    import storm_amd
    # (1500 intervals: |D| ~ 2^1820; between 4.45 and 4.6 the discriminant oscillates with roots 0.0065-0.008 apart and no
    # column of the running product underflows, so the numpy restatement is a fair judge of every sign)
    n_int, dx, u, vg, a_s, c1, outer_u = 1500, 0.02, -18.0, 21.0, 0.0, 1.0, 3.0
    sys_ = dict(u=np.full(n_int, u), v_gamma=np.full(n_int, vg), a_star=np.full(n_int, a_s), c1=np.full(n_int, c1),
                rot=np.zeros(n_int), delta_over_x=np.full(n_int, dx), inner_c1=c1, inner_rot=0.0, outer_u=outer_u)
    h = storm_amd.HipSolver("colloc2", 0, 0, sys_, n_int, 1)
    h.set_option("arithmetic", arith)
    if segments:
        h.set_option("segments", segments)
    f = np.linspace(4.45, 4.6, 61)
    ulp = storm_amd.relative_to_ulp(1e-9)
    raw, raw_dets, raw_inv = h.scan_and_optimize(f, ulp)
    assert not np.any(np.isfinite(raw_dets)) and raw == []
    ref = [uniform_colloc2_reference(n_int, dx, u, vg, a_s, c1, x, outer_u) for x in f]
    assert all(r[2] for r in ref)                                       # (no column of the product underflows on this grid)
    ref_sign = np.array([r[0] for r in ref])
    want = np.nonzero(ref_sign[1:] != ref_sign[:-1])[0].tolist()
    assert len(want) >= 15
    h.set_option("scaled_scan", 1)
    got, dets, inv = h.scan_and_optimize(f, ulp)
    assert [b["index"] for b in got] == want and not inv
    assert np.all(np.isinf(dets)) and np.array_equal(np.sign(dets), ref_sign)
    for b in got:
        assert np.isinf(b["lo_f"]) and np.isinf(b["hi_f"]) and np.sign(b["lo_f"]) != np.sign(b["hi_f"])
        assert f[b["index"]] <= b["lo_x"] < b["root"] < b["hi_x"] <= f[b["index"] + 1] and b["evals"] >= 2
        assert (b["hi_x"] - b["lo_x"]) <= 4e-9 * b["root"]
        below = uniform_colloc2_reference(n_int, dx, u, vg, a_s, c1, b["root"] * (1 - 1e-6), outer_u)[0]
        above = uniform_colloc2_reference(n_int, dx, u, vg, a_s, c1, b["root"] * (1 + 1e-6), outer_u)[0]
        assert below == np.sign(b["lo_f"]) and above == np.sign(b["hi_f"]), (b, below, above)
    # the two-call form takes the saturated end values (their signs) and evaluates the ends again
    found, _, _ = h.scan_brackets(f)
    two = h.optimize_brackets(found, ulp)
    assert [t["root"] for t in two] == [b["root"] for b in got]


def test_scaled_scan_through_storm_multi():
    """The same overflowing system through storm_multi (slices scanned with storm_scan_brackets, brackets dealt and refined
    with storm_optimize_brackets -- the saturated end values travel between the two): same bracket list, same roots as the
    single handle."""
    import storm_amd
    from storm_amd import ffi
    n_int, dx, u, vg, a_s, c1, outer_u = 1500, 0.02, -18.0, 21.0, 0.0, 1.0, 3.0
    sys_ = dict(u=np.full(n_int, u), v_gamma=np.full(n_int, vg), a_star=np.full(n_int, a_s), c1=np.full(n_int, c1),
                rot=np.zeros(n_int), delta_over_x=np.full(n_int, dx), inner_c1=c1, inner_rot=0.0, outer_u=outer_u)
    f = np.linspace(4.45, 4.6, 61)
    ulp = storm_amd.relative_to_ulp(1e-9)
    h = storm_amd.HipSolver("colloc2", 0, 0, sys_, n_int, 1)
    h.set_option("scaled_scan", 1)
    one, _, _ = h.scan_and_optimize(f, ulp)
    assert len(one) >= 15
    m = ffi.HipMulti("colloc2", 0, 0, sys_, n_int, 1, [0, 0, 0])
    m.set_option("scaled_scan", 1)
    got, _, inv = m.scan_and_optimize(f, ulp)
    assert not inv and [b["index"] for b in got] == [b["index"] for b in one]
    assert [b["root"] for b in got] == [b["root"] for b in one]
    m.close()
