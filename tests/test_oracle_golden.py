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
