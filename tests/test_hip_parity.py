"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle and the reference's golden values.  Bit-exact unless stated: the library
is built with un-fused arithmetic in the reference's operation order."""
import os

import numpy as np
import pytest

from helpers import mode_id, ndarray_linspace

pytestmark = pytest.mark.gpu

SCHEMES = ["colloc2", "colloc4", "colloc6", "colloc8", "magnus2", "magnus4", "magnus6", "magnus8"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same_bits(a, b):
    return np.array_equal(bits(a), bits(b))


@pytest.fixture(scope="module")
def host():
    from storm_amd import host as h
    h.load_host_library()
    return h


@pytest.fixture(scope="module")
def host_models(host, golden_dir):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = host.Model.from_gsm_npz(os.path.join(golden_dir, f"gsm_{name}.npz"))
        return cache[name]

    return get


def hip_from_oracle(o):
    """HipSolver fed directly (through storm_system_desc) with the oracle's discretised arrays."""
    import storm_amd
    return storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)


# ---------------------------------------------------------------------------------------------
# the reference's own tests, run on the GPU through the C++ host mirror
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scheme", SCHEMES)
def test_frequencies_golden(host, host_models, known_answers, scheme):
    # reference src/dynamic_interface.rs:241-503 (test_frequencies_<scheme>)
    ka = known_answers["radial_roots_zams"]
    model = host_models("zams")
    determinant = host.ErasedSolver(model, 0, 0, scheme, model.field("r_coord"))
    points = host.linspace(1.0, 25.0, 25)
    res, invalid = determinant.scan_and_optimize(points, host.Precision.ULP(1))
    assert not invalid
    assert [r["root"] for r in res] == ka["roots"][scheme]


def test_polytrope(host, known_answers):
    # reference src/dynamic_interface.rs:505-521
    model = host.Model.Polytrope0(5.0 / 3.0)
    solver = host.ErasedSolver(model, 0, 0, "colloc6", host.linspace(0.0, 1.0, 10000))
    res, _ = solver.scan_and_optimize(np.array([3.0, 4.0]), host.Precision.ULP(1))
    assert len(res) == 1
    assert res[0]["root"] / host.polytrope0_exact(5.0 / 3.0, 0, 1) - 1.0 < 1e-13


@pytest.mark.parametrize("name", ["radial", "radial_other", "dipole", "quadrupole"])
def test_mode_id(host, host_models, known_answers, name):
    # reference src/postprocessing.rs:376-498
    ka = known_answers["mode_id"][name]
    ell, m, lo, hi, steps, mname = ka["args"]
    model = host_models(mname)
    determinant = host.ErasedSolver(model, ell, m, "colloc2", model.field("r_coord"))
    res, _ = determinant.scan_and_optimize(ndarray_linspace(lo, hi, steps), host.Precision.ULP(1))
    assert len(res) == len(ka["expected"])
    vecs = determinant.eigenvector(np.array([r["root"] for r in res]))
    u = model.field("u")
    assert [list(mode_id(v, ell, u)) for v in vecs] == ka["expected"]


# ---------------------------------------------------------------------------------------------
# kernel parity against the oracle on identical inputs
# ---------------------------------------------------------------------------------------------
def each_det_path(h):
    """Forces every discriminant kernel in turn (all must give the same bits): fused, split, quad,
    hex with 16 and with 32 lanes per frequency, the column-owner kernel (4 lanes), det_mix with 50 % column owners
    (hex_lanes = -50: the launch large batches get; Colloc6/8 and batches under 32 frequencies fall back to det_hex);
    restores the automatic choice afterwards."""
    for path, lanes in ((1, 0), (2, 0), (3, 0), (4, 16), (4, 32), (4, 4), (4, -50)):
        h.set_option("path", path)
        h.set_option("hex_lanes", lanes)
        yield f"path {path} lanes {lanes}"
    h.set_option("path", 0)
    h.set_option("hex_lanes", 0)


@pytest.mark.parametrize("ell", [0, 1, 2])
@pytest.mark.parametrize("scheme", SCHEMES)
def test_det_bit_exact_both_paths(oracle, oracle_models, scheme, ell):
    model = oracle_models("zams")
    o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
    h = hip_from_oracle(o)
    freqs = np.concatenate([oracle.linspace(0.7, 30.0, 61), [3.3047719, 10.0, 24.999]])
    ref = o.det(freqs, threads=16)
    for path in each_det_path(h):
        got = h.det(freqs)
        assert same_bits(ref, got), f"path {path}"


@pytest.mark.parametrize("scheme", ["colloc4", "magnus6", "colloc8"])
@pytest.mark.parametrize("ell,m", [(1, 1), (2, -1), (3, 2), (2, 0)])
def test_det_uniform_rotation(oracle, golden_dir, scheme, ell, m):
    # BASELINE config 4: tams with set-rotation-constant 0.1 (parity unpinned by the reference's
    # tests for rot != 0: the oracle is the yardstick)
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    model.set_rotation_constant(0.1)
    o = oracle.Solver(model, ell, m, scheme, model.field("r_coord"))
    h = hip_from_oracle(o)
    freqs = oracle.linspace(1.0, 25.0, 40)
    ref = o.det(freqs, threads=16)
    for path in each_det_path(h):
        assert same_bits(ref, h.det(freqs)), f"path {path}"


@pytest.mark.parametrize("scheme", ["colloc2", "colloc6", "magnus4", "magnus8"])
def test_det_differential_rotation(oracle, golden_dir, scheme):
    # non-uniform rotation profile -> per-point evaluation of the rotation terms
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_zams.npz"))
    x = model.field("r_coord")
    model.set_field("rot", 0.05 + 0.1 * np.exp(-3.0 * x) - 0.02 * x)
    o = oracle.Solver(model, 2, 1, scheme, x)
    h = hip_from_oracle(o)
    freqs = oracle.linspace(1.0, 20.0, 33)
    ref = o.det(freqs, threads=16)
    for path in each_det_path(h):
        assert same_bits(ref, h.det(freqs)), f"path {path}"


@pytest.mark.parametrize("scheme", ["colloc4", "magnus2"])
def test_scan_and_eigenvectors_bit_exact(oracle, oracle_models, scheme):
    # BASELINE config 1 shape: zams, ell=2, relative precision 1e-8
    model = oracle_models("zams")
    o = oracle.Solver(model, 2, 0, scheme, model.field("r_coord"))
    h = hip_from_oracle(o)
    freqs = oracle.linspace(1.0, 25.0, 200)
    ulp = oracle.relative_to_ulp(1e-8)
    rres, rdets, rinv = o.scan_and_optimize(freqs, ulp, threads=16)
    gres, gdets, ginv = h.scan_and_optimize(freqs, ulp)
    assert same_bits(rdets, gdets)
    assert rinv == ginv
    assert [r["index"] for r in rres] == [g["index"] for g in gres]        # bracket indices
    for k in ("lo_x", "lo_f", "hi_x", "hi_f", "root"):
        assert same_bits([r[k] for r in rres], [g[k] for g in gres]), k
    assert [r["evals"] for r in rres] == [g["evals"] for g in gres]
    roots = np.array([g["root"] for g in gres])
    rdet, rvec = o.eigenvector(roots, threads=16)
    gdet, gvec = h.eigenvector(roots)
    assert same_bits(rdet, gdet)
    assert same_bits(rvec, gvec)


def test_polytrope_model_config3_shape(oracle):
    # BASELINE config 3 inputs at reduced size: n=3 polytrope, Colloc4 + Magnus2, ell=2
    model = oracle.Model.polytrope(3.0, 5.0 / 3.0, 0.01)
    grid = model.grid(2)  # resample = 2
    for scheme in ("colloc4", "magnus2"):
        o = oracle.Solver(model, 2, 0, scheme, grid)
        h = hip_from_oracle(o)
        freqs = oracle.linspace(1.0, 25.0, 300)
        assert same_bits(o.det(freqs, threads=16), h.det(freqs))


def test_inverse_grid_g_modes(oracle, golden_dir):
    # BASELINE config 5 scan shape: period-linear grid (--inverse) in the dense g-mode range
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    model.set_rotation_constant(0.1)
    o = oracle.Solver(model, 2, 0, "colloc4", model.field("r_coord"))
    h = hip_from_oracle(o)
    freqs = oracle.rev_linspace(0.5, 1.0, 64)
    ulp = oracle.relative_to_ulp(1e-8)
    rres, rdets, _ = o.scan_and_optimize(freqs, ulp, threads=16)
    gres, gdets, _ = h.scan_and_optimize(freqs, ulp)
    assert same_bits(rdets, gdets)
    assert [r["index"] for r in rres] == [g["index"] for g in gres]
    assert same_bits([r["root"] for r in rres], [g["root"] for g in gres])


# ---------------------------------------------------------------------------------------------
# edge cases
# ---------------------------------------------------------------------------------------------
def test_empty_and_tiny_batches(oracle, oracle_models):
    model = oracle_models("zams")
    o = oracle.Solver(model, 1, 0, "colloc4", model.field("r_coord"))
    h = hip_from_oracle(o)
    assert h.det(np.array([])).size == 0
    res, dets, inv = h.scan_and_optimize(np.array([]), 1)
    assert res == [] and dets.size == 0 and not inv
    res, dets, inv = h.scan_and_optimize(np.array([5.0]), 1)
    assert res == [] and same_bits(dets, o.det(5.0))
    for n in (1, 2, 3, 5, 63, 64, 65, 130):
        f = oracle.linspace(2.0, 9.0, n) if n > 1 else np.array([2.0])
        for path in each_det_path(h):
            assert same_bits(o.det(f, threads=16), h.det(f)), (n, path)


@pytest.mark.parametrize("npts", [2, 3, 4, 6])
def test_tiny_grids(oracle, npts):
    # 1..5 intervals: exercises the loop remainders of the ordered-product kernel
    model = oracle.Model.poly0(5.0 / 3.0)
    grid = oracle.linspace(0.0, 1.0, npts)
    for scheme in ("colloc2", "magnus4", "colloc8"):
        for ell in (0, 2):
            o = oracle.Solver(model, ell, 0, scheme, grid)
            h = hip_from_oracle(o)
            f = oracle.linspace(1.0, 6.0, 7)
            ref = o.det(f)
            for path in each_det_path(h):
                assert same_bits(ref, h.det(f)), (scheme, ell, path)
            rdet, rvec = o.eigenvector(f[:3])
            gdet, gvec = h.eigenvector(f[:3])
            assert same_bits(rdet, gdet) and same_bits(rvec, gvec)


def test_non_finite_discriminants_and_invalid_flag(oracle, oracle_models):
    # frequency 0 with ell != 0 divides by zero exactly like the reference; the scan must drop the
    # pair and raise the flag (reference src/dynamic_interface.rs:130-137,153-157)
    model = oracle_models("zams")
    o = oracle.Solver(model, 2, 0, "colloc2", model.field("r_coord"))
    h = hip_from_oracle(o)
    f = np.array([0.0, 1.0, 2.0, 3.0, 4.0])
    ref, got = o.det(f), h.det(f)
    assert np.array_equal(np.isnan(ref), np.isnan(got)) and np.array_equal(np.isinf(ref), np.isinf(got))
    fin = np.isfinite(ref)
    assert same_bits(ref[fin], got[fin])
    rres, _, rinv = o.scan_and_optimize(f, 1)
    gres, _, ginv = h.scan_and_optimize(f, 1)
    assert rinv == ginv
    assert same_bits([r["root"] for r in rres], [g["root"] for g in gres])


def test_capacity_and_argument_errors(oracle, oracle_models):
    import storm_amd
    model = oracle_models("zams")
    o = oracle.Solver(model, 0, 0, "colloc2", model.field("r_coord"))
    h = hip_from_oracle(o)
    with pytest.raises(storm_amd.StormError) as e:
        h.scan_and_optimize(oracle.linspace(1.0, 25.0, 25), 1, cap=3)
    assert e.value.code == -5
    with pytest.raises(storm_amd.StormError) as e:
        h.scan_and_optimize(oracle.linspace(1.0, 25.0, 25), 0)
    assert e.value.code == -1
    with pytest.raises(storm_amd.StormError):
        h.set_option("no_such_option", 1)
    # the handle is still usable after errors
    assert same_bits(h.det(np.array([3.0])), o.det(3.0))


def test_chunked_staging_gives_identical_results(oracle, oracle_models):
    model = oracle_models("zams")
    o = oracle.Solver(model, 2, 0, "magnus4", model.field("r_coord"))
    h = hip_from_oracle(o)
    f = oracle.linspace(1.0, 25.0, 100)
    h.set_option("path", 2)
    ref = h.det(f)
    h.set_option("steps_bytes_max", 7 * o.n_int * 128)  # 7 frequencies per chunk
    assert same_bits(ref, h.det(f))
    roots = f[:9]
    d1, v1 = h.eigenvector(roots)
    h.set_option("steps_bytes_max", 1 << 32)
    d2, v2 = h.eigenvector(roots)
    assert same_bits(d1, d2) and same_bits(v1, v2)
    assert same_bits(ref, o.det(f, threads=16))


# ---------------------------------------------------------------------------------------------
# full BASELINE sizes: size-independent properties + sampled oracle comparison
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("scheme,ell", [("magnus6", 1), ("colloc8", 3)])
def test_full_size_scan_config2(oracle, golden_dir, scheme, ell):
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
    h = hip_from_oracle(o)
    n = 10000
    f = oracle.linspace(1.0, 25.0, n)
    ulp = oracle.relative_to_ulp(1e-8)
    gres, gdets, ginv = h.scan_and_optimize(f, ulp)
    # (1) the two kernel paths agree bit for bit on the whole grid
    for path in each_det_path(h):
        assert same_bits(gdets, h.det(f)), path
    h.set_option("path", 0)
    # (2) sampled comparison with the oracle (512 of 10^4 points) and every bracket end point
    rng = np.random.default_rng(1234)
    idx = np.unique(np.concatenate([rng.choice(n, 512, replace=False),
                                    [g["index"] for g in gres], [g["index"] + 1 for g in gres]]))
    assert same_bits(o.det(f[idx], threads=16), gdets[idx])
    # (3) brackets are exactly the adjacent finite sign changes, ascending, roots inside
    sgn = np.signbit(gdets)
    expect = [i for i in range(n - 1) if sgn[i] != sgn[i + 1] and np.isfinite(gdets[i]) and np.isfinite(gdets[i + 1])]
    assert [g["index"] for g in gres] == expect
    roots = np.array([g["root"] for g in gres])
    assert np.all(np.diff(roots) > 0)
    for g in gres:
        assert f[g["index"]] <= g["lo_x"] <= g["root"] <= g["hi_x"] <= f[g["index"] + 1]
        assert np.signbit(g["lo_f"]) != np.signbit(g["hi_f"])
        assert oracle.lib() and (np.float64(g["hi_x"]).view(np.uint64) - np.float64(g["lo_x"]).view(np.uint64)) <= ulp
    # (4) refinement is bit-identical with the oracle's on every bracket (cheap: ~10 evals each)
    for g in gres[:: max(1, len(gres) // 8)]:
        r = o.optimize(f[g["index"]], gdets[g["index"]], f[g["index"] + 1], gdets[g["index"] + 1], ulp)
        assert same_bits([r["root"], r["lo_x"], r["hi_x"]], [g["root"], g["lo_x"], g["hi_x"]]) and r["evals"] == g["evals"]
    # (5) eigenvectors: residual-free check against the oracle for a few modes, finite for all
    gdet, gvec = h.eigenvector(roots)
    assert np.all(np.isfinite(gvec))
    sel = np.arange(len(roots))[:: max(1, len(roots) // 4)]
    rdet, rvec = o.eigenvector(roots[sel], threads=16)
    assert same_bits(rdet, gdet[sel]) and same_bits(rvec, gvec[sel])


def test_host_scan_command(host, host_models, oracle, oracle_models):
    # StormState::scan mirror (reference src/main.rs:793-838): scan + eigenvector per mode
    hm, om = host_models("zams"), oracle_models("zams")
    grid = hm.grid(1)
    res, vecs = host.scan(hm, grid, 1, 0, 2.0, 12.0, 50, inverse=False, precision=1e-8, scheme="colloc4")
    o = oracle.Solver(om, 1, 0, "colloc4", om.grid(1))
    rres, _, _ = o.scan_and_optimize(oracle.linspace(2.0, 12.0, 50), oracle.relative_to_ulp(1e-8), threads=16)
    assert same_bits([r["root"] for r in rres], [g["root"] for g in res])
    _, rvec = o.eigenvector(np.array([r["root"] for r in rres]), threads=16)
    assert same_bits(rvec, vecs)


@pytest.mark.parametrize("scheme", ["colloc8", "magnus8", "colloc4"])
def test_batches_sent_in_pieces_of_wave_rounds(oracle, oracle_models, scheme):
    # option "hex_rounds": a batch with more waves than the GPU has slots goes out as several launches of whole wave
    # rounds (one-wave-per-SIMD schemes; for the others the option changes nothing): 3000 frequencies at 32 lanes each
    # are 1500 waves for 1024 slots.  Same bits as one launch; a random sample is checked against the oracle.
    model = oracle_models("zams")
    o = oracle.Solver(model, 2, 0, scheme, model.field("r_coord"))
    h = hip_from_oracle(o)
    freqs = oracle.linspace(0.7, 30.0, 3001)[:-1]
    h.set_option("path", 4)
    results = {}
    for lanes in (16, 32):
        h.set_option("hex_lanes", lanes)
        for rounds in (0, 1, 2):
            h.set_option("hex_rounds", rounds)
            results[(lanes, rounds)] = h.det(freqs)
    first = results[(16, 0)]
    for key, got in results.items():
        assert np.array_equal(got.view(np.uint64), first.view(np.uint64)), key
    pick = np.random.default_rng(5).choice(freqs.size, 64, replace=False)
    ref = o.det(freqs[pick], threads=16)
    assert np.array_equal(first[pick].view(np.uint64), ref.view(np.uint64))
    with pytest.raises(Exception):
        h.set_option("hex_rounds", -1)
