"""GPU parity of the device post-processing against the CPU oracle's restatement of
Rotating1DPostprocessing::new / count_windings (reference src/postprocessing.rs:58-355), through the C ABI.
Bit-exact: all 13 fields, the crossing counts, the radial order and the node positions.  The reference pins
only the mode ids (src/postprocessing.rs:424-498, reproduced below on the GPU); the field VALUES are asserted
nowhere in the reference ("parity unpinned": GPU == oracle)."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import ndarray_linspace

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same_bits(a, b):
    return np.array_equal(bits(a), bits(b))


def hip_from_oracle(o):
    import storm_amd
    return storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)


def hip_post_from_oracle(model_on_grid):
    import storm_amd
    return storm_amd.HipPostproc({k: model_on_grid.field(k) for k in storm_amd.MODEL_FIELDS})


def assert_same_post(ref, got, label=""):
    assert ref["ids"].tolist() == got["ids"].tolist(), label
    assert ref["n_nodes"].tolist() == got["n_nodes"].tolist(), label
    for k, (a, b) in enumerate(zip(ref["nodes"], got["nodes"])):
        assert same_bits(a, b), (label, "nodes", k)
    if got["fields"] is not None:
        for i, name in enumerate(__import__("storm_amd").POST_FIELDS):
            assert same_bits(ref["fields"][:, i, :], got["fields"][:, i, :]), (label, name)


CASES = [  # model, ell, m, uniform rotation, scheme, scan range
    ("zams", 0, 0, 0.0, "colloc4", (3.0, 20.0, 60)),
    ("zams", 1, 0, 0.0, "magnus2", (1.0, 20.0, 80)),
    ("tams", 2, 0, 0.0, "colloc2", (2.0, 20.0, 80)),
    ("tams", 1, 1, 0.1, "colloc4", (1.0, 12.0, 60)),
    ("tams", 2, -1, 0.1, "magnus4", (1.0, 12.0, 60)),
    ("zams", 3, 3, 0.05, "colloc6", (2.0, 15.0, 50)),     # |m| == ell: xi_tn = 0
    ("zams", 4, -2, 0.05, "colloc4", (2.0, 15.0, 50)),
    ("joel", 0, 0, 0.0, "colloc2", (2.0, 25.0, 25)),
]


@pytest.mark.parametrize("name,ell,m,rot,scheme,rng", CASES)
def test_postprocess_bit_exact(oracle, golden_dir, name, ell, m, rot, scheme, rng):
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, f"gsm_{name}.npz"))
    if rot:
        model.set_rotation_constant(rot)
    grid = model.field("r_coord")
    o = oracle.Solver(model, ell, m, scheme, grid)
    res, _, _ = o.scan_and_optimize(oracle.linspace(*rng), oracle.relative_to_ulp(1e-9), threads=16)
    assert len(res) >= 3
    roots = np.array([b["root"] for b in res])
    _, vecs = o.eigenvector(roots, threads=16)
    ref = model.postprocess(ell, m, roots, vecs, threads=16)
    post = hip_post_from_oracle(model)
    # (1) host eigenvectors in
    assert_same_post(ref, post.postprocess(ell, m, roots, vecs), "postprocess_batch")
    assert_same_post(ref, post.postprocess(ell, m, roots, vecs, want_fields=False), "ids only")
    # (2) fused with the eigenvector kernels: nothing but ids/nodes/fields crosses PCIe
    h = hip_from_oracle(o)
    got = h.modes(post, roots, want_vec=True)
    assert same_bits(got["vec"], vecs)
    assert_same_post(ref, got, "modes_batch")
    assert_same_post(ref, h.modes(post, roots, want_fields=False), "modes_batch ids only")


def test_differential_rotation_and_resampled_grid(oracle, golden_dir):
    # non-uniform rotation + the model interpolated onto a finer solver grid (Model::grid, resample 2)
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_zams.npz"))
    x = model.field("r_coord")
    model.set_field("rot", 0.05 + 0.1 * np.exp(-3.0 * x) - 0.02 * x)
    grid = model.grid(2)
    on_grid = model.eval(grid)
    assert len(on_grid) == grid.size
    o = oracle.Solver(model, 2, 1, "colloc4", grid)
    res, _, _ = o.scan_and_optimize(oracle.linspace(2.0, 12.0, 50), oracle.relative_to_ulp(1e-9), threads=16)
    roots = np.array([b["root"] for b in res])
    _, vecs = o.eigenvector(roots, threads=16)
    ref = on_grid.postprocess(2, 1, roots, vecs, threads=16)
    post = hip_post_from_oracle(on_grid)
    assert_same_post(ref, hip_from_oracle(o).modes(post, roots), "resampled")


@pytest.fixture(scope="module")
def host():
    from storm_amd import host as h
    h.load_host_library()
    return h


@pytest.mark.parametrize("name", ["radial", "radial_other", "dipole", "quadrupole"])
def test_mode_id_golden_on_device(host, golden_dir, known_answers, name):
    # reference src/postprocessing.rs:376-498 (compute_mode_id) with every step on the GPU
    ka = known_answers["mode_id"][name]
    ell, m, lo, hi, steps, mname = ka["args"]
    model = host.Model.from_gsm_npz(os.path.join(golden_dir, f"gsm_{mname}.npz"))
    grid = model.field("r_coord")
    determinant = host.ErasedSolver(model, ell, m, "colloc2", grid)
    res, _ = determinant.scan_and_optimize(ndarray_linspace(lo, hi, steps), host.Precision.ULP(1))
    roots = np.array([r["root"] for r in res])
    pp = host.PostProcessor(model.eval(grid))
    fused = pp.modes(determinant, roots)
    assert fused["ids"].tolist() == ka["expected"]
    split = pp.post_process(ell, m, roots, determinant.eigenvector(roots))
    assert split["ids"].tolist() == ka["expected"]
    assert [len(n) for n in split["nodes"]] == [e[0] + e[1] for e in ka["expected"]]
    # unit mode inertia after normalisation
    F = {k: split["fields"][:, i, :] for i, k in enumerate(host.POST_FIELDS)}
    r, rho = grid, model.field("rho")
    trap = np.empty_like(r)
    trap[0], trap[-1] = 0.5 * (r[1] - r[0]), 0.5 * (r[-1] - r[-2])
    trap[1:-1] = 0.5 * (r[2:] - r[:-2])
    inertia = np.sum((rho * r * r * (F["xi_r"] ** 2 + ell * (ell + 1) * F["xi_h"] ** 2) * trap)[:, 1:], axis=1)
    np.testing.assert_allclose(inertia, 1.0, rtol=1e-12)


@pytest.mark.parametrize("npts", [2, 3, 4, 7, 65, 130])
def test_tiny_and_ragged_grids(oracle, npts):
    # loop remainders of the ordered sum and of the pair loop; Polytrope0 has p = inf (rho'(0) = 0 for ell = 0)
    model = oracle.Model.poly0(5.0 / 3.0)
    grid = oracle.linspace(0.0, 1.0, npts)
    on_grid = model.eval(grid)
    post = hip_post_from_oracle(on_grid)
    for ell in (0, 1, 2):
        o = oracle.Solver(model, ell, 0, "colloc2", grid)
        f = oracle.linspace(1.5, 6.0, 5)
        _, vecs = o.eigenvector(f)
        ref = on_grid.postprocess(ell, 0, f, vecs)
        assert_same_post(ref, post.postprocess(ell, 0, f, vecs), (npts, ell))
        assert_same_post(ref, hip_from_oracle(o).modes(post, f), (npts, ell, "fused"))


def test_full_size_modes_config2(oracle, golden_dir):
    # BASELINE config 2 shape: tams, 10^4-point scan, every mode identified on the device; compared with the
    # oracle mode by mode, plus the ordering property the ids must have on a non-rotating star
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    grid = model.field("r_coord")
    o = oracle.Solver(model, 2, 0, "colloc4", grid)
    h = hip_from_oracle(o)
    res, _, _ = h.scan_and_optimize(oracle.linspace(1.0, 25.0, 10000), oracle.relative_to_ulp(1e-8))
    roots = np.array([b["root"] for b in res])
    assert roots.size >= 25
    post = hip_post_from_oracle(model)
    got = h.modes(post, roots, want_vec=True)
    ref = model.postprocess(2, 0, roots, got["vec"], threads=16)
    assert_same_post(ref, got, "config 2")
    orders = got["ids"][:, 2]
    assert np.all(np.diff(orders) >= 0) and orders[0] < 0 < orders[-1]   # g modes, then p modes, ascending n_pg


def test_node_capacity_and_chunking(oracle, oracle_models):
    model = oracle_models("zams")
    grid = model.field("r_coord")
    o = oracle.Solver(model, 0, 0, "colloc4", grid)
    f = np.array([18.3, 21.1, 24.7])
    _, vecs = o.eigenvector(f)
    ref = model.postprocess(0, 0, f, vecs)
    assert ref["n_nodes"].min() > 3
    post = hip_post_from_oracle(model)
    got = post.postprocess(0, 0, f, vecs, node_cap=3)          # truncated node list, full count reported
    assert got["n_nodes"].tolist() == ref["n_nodes"].tolist()
    for a, b in zip(ref["nodes"], got["nodes"]):
        assert same_bits(a[:3], b)
    got0 = post.postprocess(0, 0, f, vecs, node_cap=0)          # no node output at all
    assert got0["ids"].tolist() == ref["ids"].tolist()
    empty = post.postprocess(0, 0, np.array([]), np.empty((0, grid.size * 4)))
    assert empty["ids"].shape == (0, 3)


def test_argument_errors(oracle, oracle_models):
    import storm_amd
    from storm_amd import ffi
    model = oracle_models("zams")
    grid = model.field("r_coord")
    post = hip_post_from_oracle(model)
    coarse = oracle.Solver(model, 1, 0, "colloc2", grid[::2])
    with pytest.raises(storm_amd.StormError) as e:
        hip_from_oracle(coarse).modes(post, np.array([3.0]))
    assert e.value.code == -1 and "grid" in str(e.value)
    bad = {k: model.field(k) for k in storm_amd.MODEL_FIELDS}
    bad["r_coord"] = bad["r_coord"] + 1e-3                     # assert_eq!(model.r_coord[0], 0.)
    with pytest.raises(storm_amd.StormError) as e:
        storm_amd.HipPostproc(bad)
    assert e.value.code == -1
    L = ffi.load_library()
    assert L.storm_postprocess_batch(None, 0, 0, None, 0, None, None, None, None, 0) == -1
    L.storm_postproc_destroy(None)
