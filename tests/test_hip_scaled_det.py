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
