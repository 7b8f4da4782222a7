"""Pin the CPU oracle against the reference's own known answers.

Everything here is exact equality on f64 unless stated: the reference's tests are
assert_eq! on f64, so they pin the operation order.
"""
import numpy as np
import pytest

from helpers import mode_id, ndarray_linspace

SCHEMES = ["colloc2", "colloc4", "colloc6", "colloc8", "magnus2", "magnus4", "magnus6", "magnus8"]


def test_exp_known_answer(oracle, known_answers):
    # reference src/linalg/storage.rs:498-520
    kat = known_answers["exp_kat"]
    a = np.array(kat["input_row_major_3x3"]).reshape(3, 3)
    out = oracle.mat_exp(a).reshape(-1, order="F")
    assert out.tolist() == kat["output_column_major_3x3"]


@pytest.mark.parametrize("scheme", SCHEMES)
def test_golden_roots(oracle, oracle_models, known_answers, scheme):
    # reference src/dynamic_interface.rs:241-503
    ka = known_answers["radial_roots_zams"]
    model = oracle_models("zams")
    solver = oracle.Solver(model, ka["ell"], ka["m"], scheme, model.field("r_coord"))
    lo, hi, n = ka["grid"]
    res, dets, invalid = solver.scan_and_optimize(oracle.linspace(lo, hi, n), ka["ulp"], threads=8)
    assert not invalid
    assert [r["root"] for r in res] == ka["roots"][scheme]


def test_polytrope0(oracle, known_answers):
    # reference src/dynamic_interface.rs:505-521
    ka = known_answers["polytrope0"]
    model = oracle.Model.poly0(ka["gamma1"])
    grid = oracle.linspace(0.0, 1.0, ka["npoints"])
    solver = oracle.Solver(model, 0, 0, ka["scheme"], grid)
    res, _, _ = solver.scan_and_optimize(np.array(ka["scan"]), 1)
    assert len(res) == ka["n_roots"]
    assert res[0]["root"] / oracle.polytrope0_exact(ka["gamma1"], 0, 1) - 1.0 < ka["rel_tol"]


@pytest.mark.parametrize("name", ["radial", "radial_other", "dipole", "quadrupole"])
def test_mode_ids(oracle, oracle_models, known_answers, name):
    # reference src/postprocessing.rs:376-498: pins eigenvector() (banded elimination + back-substitution)
    ka = known_answers["mode_id"][name]
    ell, m, lo, hi, steps, mname = ka["args"]
    model = oracle_models(mname)
    grid = model.field("r_coord")
    solver = oracle.Solver(model, ell, m, "colloc2", grid)
    res, _, _ = solver.scan_and_optimize(ndarray_linspace(lo, hi, steps), 1, threads=8)
    assert len(res) == len(ka["expected"])
    roots = np.array([r["root"] for r in res])
    _, vecs = solver.eigenvector(roots, threads=8)
    u = model.field("u")
    got = [list(mode_id(v, ell, u)) for v in vecs]
    assert got == ka["expected"]
    # the same through the oracle's full Rotating1DPostprocessing::new (src/postprocessing.rs:58-301), which is
    # what the reference test calls: pins the normalisation + count_windings restatement
    post = model.postprocess(ell, m, roots, vecs, threads=8)
    assert post["ids"].tolist() == ka["expected"]
    assert [len(n) for n in post["nodes"]] == [e[0] + e[1] for e in ka["expected"]]


def test_postprocess_properties(oracle, oracle_models):
    # properties the reference's definitions imply (no reference test asserts the field values: parity unpinned)
    model = oracle_models("zams")
    grid = model.field("r_coord")
    r, rho = grid, model.field("rho")
    for ell, m, rot in ((0, 0, 0.0), (1, 0, 0.0), (2, 1, 0.05), (3, -2, 0.05)):
        model.set_rotation_constant(rot)
        solver = oracle.Solver(model, ell, m, "colloc4", grid)
        res, _, _ = solver.scan_and_optimize(oracle.linspace(3.0, 9.0, 40), oracle.relative_to_ulp(1e-10), threads=8)
        assert res
        roots = np.array([b["root"] for b in res])
        _, vecs = solver.eigenvector(roots, threads=8)
        post = model.postprocess(ell, m, roots, vecs, threads=8)
        F = {k: post["fields"][:, i, :] for i, k in enumerate(oracle.POST_FIELDS)}
        lam = ell * (ell + 1)
        trap = np.empty_like(r)
        trap[0], trap[-1] = 0.5 * (r[1] - r[0]), 0.5 * (r[-1] - r[-2])
        trap[1:-1] = 0.5 * (r[2:] - r[:-2])
        inertia = np.sum((rho * r * r * (F["xi_r"] ** 2 + lam * F["xi_h"] ** 2) * trap)[:, 1:], axis=1)
        np.testing.assert_allclose(inertia, 1.0, rtol=1e-12)            # unit mode inertia after normalisation
        for k, v in enumerate(vecs):                                       # y's are the eigenvector, rescaled
            e = v.reshape(-1, 4)
            scale = F["y4"][k, -1] / e[-1, 3]
            np.testing.assert_allclose(F["y1"][k], e[:, 0] * scale, rtol=1e-15, atol=0)
        if ell == 0:
            assert np.all(F["xi_h"] == 0.0) and np.all(F["xi_tn"] == 0.0)
        if rot == 0.0:
            assert np.all(F["xi_tp"] == 0.0)
        for k, nodes in enumerate(post["nodes"]):
            assert len(nodes) == post["ids"][k, 0] + post["ids"][k, 1]
            assert np.all(np.diff(nodes) > 0) and np.all((nodes > 0) & (nodes < 1))
    model.set_rotation_constant(0.0)


def test_powi_and_gaunt(oracle):
    # f64::powi (compiler-builtins __powidf2) and src/gaunt.rs
    assert oracle.powi(3.0, 0) == 1.0 and oracle.powi(3.0, 1) == 3.0 and oracle.powi(3.0, -1) == 1.0 / 3.0
    assert oracle.powi(1.1, 5) == (1.1 * ((1.1 * 1.1) * (1.1 * 1.1)))
    assert oracle.powi(0.7, -2) == 1.0 / (0.7 * 0.7)
    q = oracle.gaunt(2, 1)   # k=2: l=1 and l=3, m=1 -> sign -1
    b2, b3 = np.sqrt((4 - 1) / (16 - 1.0)), np.sqrt((9 - 1) / (36 - 1.0))
    np.testing.assert_allclose(q, [-b2 * ((2 - 6) / 2 + 1), -b3 * ((12 - 6) / 2 + 1), -b2 * ((6 + 2) / 2 - 1),
                                   -b3 * ((6 + 12) / 2 - 1)], rtol=1e-15)
    assert np.all(oracle.gaunt(0, 0)[[0, 2]] == 0.0)


def test_ulp_conversion(oracle):
    # Precision::as_ulp, reference src/bracket.rs:30-40 (CLI default 1e-8)
    assert oracle.relative_to_ulp(1e-8) == 45035996
    assert oracle.relative_to_ulp(1e-20) == 1


def test_linspace_forms(oracle):
    g = oracle.linspace(1.0, 25.0, 25)
    assert g[0] == 1.0 and g[-1] == 25.0 and g[7] == 1.0 + 24.0 * 7.0 / 24.0
    r = oracle.rev_linspace(0.5, 1.0, 5)
    assert r[0] == 1.0 / (1.0 / 0.5) and r[-1] == 1.0


def test_scan_edge_cases(oracle, oracle_models):
    model = oracle_models("zams")
    solver = oracle.Solver(model, 0, 0, "colloc2", model.field("r_coord"))
    res, dets, inv = solver.scan_and_optimize(np.array([]), 1)
    assert res == [] and dets.size == 0 and not inv
    res, dets, inv = solver.scan_and_optimize(np.array([5.0]), 1)
    assert res == [] and dets.size == 1
    # det() agrees with what scan reports and is deterministic
    assert solver.det(5.0)[0] == dets[0]


def test_flop_counts_are_stable(oracle, oracle_models):
    # Frozen algorithmic flop counts per (frequency, interval) evaluation, reference form
    # (add/sub/mul/div = 1).  These feed bench.py's roofline; see DESIGN.md.
    model = oracle_models("zams")
    grid = model.field("r_coord")
    n_int = grid.size - 1
    per_eval = {}
    for ell in (0, 2):
        for s in SCHEMES:
            c = oracle.Solver(model, ell, 0, s, grid).count_flops(3.0)
            per_eval[(ell, s)] = c["total"] / n_int
    # monotone in order within a family, and ell != 0 costs more than ell == 0
    assert per_eval[(2, "colloc2")] < per_eval[(2, "colloc4")] < per_eval[(2, "colloc6")] < per_eval[(2, "colloc8")]
    assert per_eval[(2, "magnus2")] < per_eval[(2, "magnus4")] < per_eval[(2, "magnus6")] < per_eval[(2, "magnus8")]
    for s in SCHEMES:
        assert per_eval[(0, s)] < per_eval[(2, s)]
