"""GPU parity of storm_sweep (several systems per launch) against the CPU oracle run
system by system, and against one storm_solver handle per system: discriminants, brackets (points, roots,
evals, grid indices), the invalid-number flags, eigenvectors and post-processing, all bit-exact.  The reference
has no batched object (each `scan` line builds its own ErasedSolver, src/main.rs:815-831), so the contract is
"identical to the per-system calls"."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def same_brackets(ref, got):
    keys = ("lo_x", "lo_f", "hi_x", "hi_f", "root")
    if len(ref) != len(got):
        return False
    for r, g in zip(ref, got):
        if any(bits(r[k]) != bits(g[k]) for k in keys) or r["evals"] != g["evals"]:
            return False
        if "index" in r and "index" in g and r["index"] != g["index"]:
            return False
    return True


def rotating(oracle, golden_dir, name, rot):
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, f"gsm_{name}.npz"))
    if rot is not None:
        if np.isscalar(rot):
            model.set_rotation_constant(rot)
        else:
            model.set_field("rot", rot(model.field("r_coord")))
    return model


def sweep_from_oracle(solvers):
    import storm_amd
    from storm_amd import ffi
    o0 = solvers[0]
    return ffi.HipSweep(o0.scheme, [(o.ell, o.m, o.system()) for o in solvers], o0.n_int, o0.p)


# (ell, m, rotation): every kernel class (ell = 0, uniform rotation, differential rotation), interleaved so the
# internal grouping by class has to permute them
MIXED = [(2, 0, None), (0, 0, 0.1), (1, 1, 0.1), (2, -1, lambda x: 0.05 + 0.1 * np.exp(-3.0 * x)), (0, 0, None),
         (3, 2, 0.07), (1, -1, lambda x: 0.02 + 0.03 * x), (2, 2, 0.1)]


@pytest.mark.parametrize("scheme", ["colloc4", "colloc2", "magnus2", "colloc6", "colloc8", "magnus4", "magnus6", "magnus8"])
def test_sweep_scan_bit_exact(oracle, golden_dir, scheme):
    models = [rotating(oracle, golden_dir, "zams", rot) for _, _, rot in MIXED]
    grid = models[0].field("r_coord")
    solvers = [oracle.Solver(mdl, ell, m, scheme, grid) for mdl, (ell, m, _) in zip(models, MIXED)]
    # ragged frequency lists (lengths not multiples of 4, one empty, one single point)
    grids = [oracle.linspace(1.0 + 0.1 * k, 11.0 + k, 37 + 5 * k) for k in range(len(MIXED))]
    grids[4] = np.empty(0)
    grids[6] = np.array([3.25])
    ulp = oracle.relative_to_ulp(1e-8)
    sweep = sweep_from_oracle(solvers)
    dets = sweep.det(grids)
    got = sweep.scan_and_optimize(grids, ulp)
    n_modes = 0
    for k, o in enumerate(solvers):
        ref, rdets, rinv = o.scan_and_optimize(grids[k], ulp, threads=16)
        assert same_bits(dets[k], rdets), (scheme, k, "det")
        br, sdets, inv = got[k]
        assert same_bits(sdets, rdets), (scheme, k, "scan det")
        assert inv == rinv, (scheme, k)
        assert same_brackets(ref, br), (scheme, k, "brackets")
        n_modes += len(ref)
    assert n_modes > 10
    st = sweep.stats()
    assert st["refine_rounds"] > 0 and st["refine_evals"] == sum(b["evals"] for br, _, _ in got for b in br)
    if scheme in ("colloc4", "magnus2"):
        # eigenvectors of every mode of every system in one pass
        roots = [np.array([b["root"] for b in br]) for br, _, _ in got]
        vecs = sweep.eigenvector(roots)
        for k, o in enumerate(solvers):
            if roots[k].size == 0:
                assert vecs[k][1].shape[0] == 0
                continue
            rdet, rvec = o.eigenvector(roots[k], threads=16)
            assert same_bits(vecs[k][1], rvec), (scheme, k, "eigenvectors")
            assert same_bits(vecs[k][0], rdet), (scheme, k, "eigen det")


def test_sweep_equals_separate_handles_and_fast_mode(oracle, golden_dir):
    """Same bits as one storm_solver per system, also in the FMA build (where the oracle is not the yardstick; there
    the single handle is pinned to the kernel the sweep uses, because contraction differs from kernel to kernel)."""
    import storm_amd
    models = [rotating(oracle, golden_dir, "tams", rot) for rot in (0.0, 0.05, 0.1)]
    grid = models[0].field("r_coord")
    cfg = [(0, 2, 0), (1, 2, 1), (2, 1, -1), (1, 0, 0), (2, 3, 0)]
    solvers = [oracle.Solver(models[i], ell, m, "colloc4", grid) for i, ell, m in cfg]
    grids = [oracle.linspace(1.0, 6.0 + k, 120 + 7 * k) for k in range(len(cfg))]
    ulp = oracle.relative_to_ulp(1e-8)
    for arith in (0, 1):
        sweep = sweep_from_oracle(solvers)
        sweep.set_option("arithmetic", arith)
        if arith:
            sweep.set_option("segments", -1)  # the kernels a single handle can be pinned to: same bits
        got = sweep.scan_and_optimize(grids, ulp)
        roots = [np.array([b["root"] for b in br]) for br, _, _ in got]
        vecs = sweep.eigenvector(roots)
        if arith:
            # the FMA build's default for a sweep: small batches (these scans, every refinement round) and all
            # eigenvectors go through segment transfer matrices.  The tolerance contract of a single handle: the same
            # brackets, roots within 1e-10, eigenvectors within 1e-7 of their largest component.
            sweep.set_option("segments", 0)
            sweep.stats(reset=True)
            got_seg = sweep.scan_and_optimize(grids, ulp)
            seg = sweep.eigenvector(roots)
            assert sweep.stats()["segment_fallbacks"] == 0
            for k in range(len(cfg)):
                a, b = got_seg[k][0], got[k][0]
                assert len(a) == len(b) and got_seg[k][2] == got[k][2], k
                for x, y in zip(a, b):
                    assert abs(x["root"] - y["root"]) <= 1e-10 * abs(y["root"]), (k, x["root"], y["root"])
                    assert y["lo_x"] <= x["root"] <= y["hi_x"] or abs(x["root"] - y["root"]) <= 1e-10 * abs(y["root"])
                scale = np.max(np.abs(got[k][1]))
                assert np.max(np.abs(got_seg[k][1] - got[k][1])) <= 1e-8 * scale, k
                for u, v in zip(seg[k][1], vecs[k][1]):
                    assert np.max(np.abs(u - v)) <= 1e-7 * np.max(np.abs(v)), (k, np.max(np.abs(u - v)) / np.max(np.abs(v)))
            # the guard: with the growth bound at 2^0 every transfer matrix is "too large", the pass is redone by the
            # sequential elimination and counted -- same bits as with the segment path switched off
            sweep.set_option("seg_growth_log2", 0)
            sweep.stats(reset=True)
            guarded = sweep.eigenvector(roots)
            assert sweep.stats()["segment_fallbacks"] == sum(r.size for r in roots)
            for k in range(len(cfg)):
                assert same_bits(guarded[k][1], vecs[k][1]) and same_bits(guarded[k][0], vecs[k][0]), k
            sweep.set_option("seg_growth_log2", 24)
            sweep.set_option("segments", -1)
        for k, o in enumerate(solvers):
            h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
            h.set_option("arithmetic", arith)
            if arith:
                h.set_option("path", 4)  # FMA contraction differs between kernels: compare like with like
                h.set_option("segments", -1)  # (a single handle's eigenvectors would take the segment path; a sweep's do not)
            ref, rdets, rinv = h.scan_and_optimize(grids[k], ulp)
            assert same_bits(got[k][1], rdets) and got[k][2] == rinv, (arith, k)
            assert same_brackets(ref, got[k][0]), (arith, k)
            rdet, rvec = h.eigenvector(roots[k])
            assert same_bits(vecs[k][1], rvec) and same_bits(vecs[k][0], rdet), (arith, k)
        for lanes in (16, 32):
            sweep.set_option("hex_lanes", lanes)
            again = sweep.det(grids)
            for k in range(len(cfg)):
                if arith:
                    # FMA build: the 16-lane kernel sums the ordered product in v_mfma_f64_4x4x4 (fused, the hardware's
                    # order), the 32-lane kernel in v_fma_f64: the fast build's tolerance contract, not the same bits
                    scale = np.max(np.abs(got[k][1]))
                    assert np.max(np.abs(again[k] - got[k][1])) <= 1e-9 * scale, (arith, lanes, k)
                    assert np.array_equal(np.signbit(again[k]), np.signbit(got[k][1])), (arith, lanes, k)
                else:
                    assert same_bits(again[k], got[k][1]), (arith, lanes, k)


def test_sweep_modes_through_host_mirror(oracle, golden_dir):
    """SolverSweep of the host mirror (one shared interpolation per model) + post-processing of all modes in one pass,
    against the oracle's per-system scan -> eigenvector -> post-process chain."""
    from storm_amd import host
    rots = [0.0, 0.05, 0.1]
    lm = [(0, 0), (1, 0), (2, 1), (2, -2)]
    hmodels, omodels = [], []
    for rot in rots:
        hm = host.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_zams.npz"))
        om = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_zams.npz"))
        hm.set_rotation_constant(rot)
        om.set_rotation_constant(rot)
        hmodels.append(hm)
        omodels.append(om)
    grid = hmodels[0].field("r_coord")
    systems = [(hmodels[i], ell, m) for i in range(len(rots)) for ell, m in lm]
    sweep = host.SolverSweep(systems, "colloc4", grid)
    assert sweep.n_sys == 12 and sweep.n_int == grid.size - 1
    grids = [host.linspace(2.0, 14.0, 90)] * sweep.n_sys
    res = sweep.scan_and_optimize(grids, host.Precision.Relative(1e-8))
    roots = [np.array([b["root"] for b in br]) for br, _ in res]
    posts = [host.PostProcessor(hmodels[i].eval(grid)) for i in range(len(rots))]
    per_sys_posts = [posts[i] for i in range(len(rots)) for _ in lm]
    ids_only = sweep.modes(per_sys_posts, roots)
    full = sweep.modes(per_sys_posts, roots, want_fields=True)
    vecs = sweep.eigenvector(roots)
    u = 0
    for i in range(len(rots)):
        for ell, m in lm:
            o = oracle.Solver(omodels[i], ell, m, "colloc4", grid)
            ref, _, rinv = o.scan_and_optimize(grids[u], oracle.relative_to_ulp(1e-8), threads=16)
            assert same_brackets(ref, res[u][0]) and res[u][1] == rinv, u
            _, rvec = o.eigenvector(roots[u], threads=16)
            assert same_bits(vecs[u], rvec), u
            rp = omodels[i].postprocess(ell, m, roots[u], rvec, threads=16)
            assert rp["ids"].tolist() == ids_only[u]["ids"].tolist() == full[u]["ids"].tolist(), u
            assert rp["n_nodes"].tolist() == full[u]["n_nodes"].tolist(), u
            for a, b in zip(rp["nodes"], full[u]["nodes"]):
                assert same_bits(a, b), u
            assert same_bits(rp["fields"], full[u]["fields"]), u
            u += 1


def test_sweep_chunked_eigenvectors(oracle, golden_dir):
    """A staging cap smaller than the mode count: chunks cut through systems and classes."""
    models = [rotating(oracle, golden_dir, "zams", rot) for _, _, rot in MIXED[:5]]
    grid = models[0].field("r_coord")
    solvers = [oracle.Solver(mdl, ell, m, "colloc2", grid) for mdl, (ell, m, _) in zip(models, MIXED)]
    sweep = sweep_from_oracle(solvers)
    grids = [oracle.linspace(2.0, 16.0, 60)] * len(solvers)
    got = sweep.scan_and_optimize(grids, 4)
    roots = [np.array([b["root"] for b in br]) for br, _, _ in got]
    assert sum(r.size for r in roots) > 12
    per_mode = (sweep.n_int * 16 + sweep.n_int * 22 + 6 + (sweep.n_int + 1) * 4) * 8
    sweep.set_option("steps_bytes_max", 5 * per_mode)
    vecs = sweep.eigenvector(roots)
    for k, o in enumerate(solvers):
        _, rvec = o.eigenvector(roots[k], threads=16)
        assert same_bits(vecs[k][1], rvec), k


def test_sweep_argument_errors(oracle, oracle_models):
    import ctypes as C
    import storm_amd
    from storm_amd import ffi
    model = oracle_models("zams")
    grid = model.field("r_coord")
    a = oracle.Solver(model, 1, 0, "colloc4", grid)
    b = oracle.Solver(model, 1, 0, "colloc2", grid)
    c = oracle.Solver(model, 1, 0, "colloc4", grid[::2])
    L = ffi.load_library()
    with pytest.raises(storm_amd.StormError, match="share the scheme"):
        descs = (ffi.SystemDesc * 2)()
        keep = []
        for k, o in enumerate((a, b)):
            s = o.system()
            arrs = [np.ascontiguousarray(s[q], dtype=np.float64) for q in ("u", "v_gamma", "a_star", "c1", "rot", "delta_over_x")]
            keep.append(arrs)
            descs[k] = ffi.SystemDesc(o.scheme, o.ell, o.m, o.n_int, o.p, *[x.ctypes.data_as(C.POINTER(C.c_double)) for x in arrs],
                                      s["inner_c1"], s["inner_rot"], s["outer_u"])
        h = C.c_void_p()
        ffi._check(L.storm_sweep_create(descs, 2, C.byref(h)))
    with pytest.raises(ValueError):
        ffi.HipSweep(a.scheme, [(1, 0, a.system()), (1, 0, c.system())], a.n_int, a.p)
    with pytest.raises(storm_amd.StormError, match="at least one"):
        ffi.HipSweep(a.scheme, [], a.n_int, a.p)
    sweep = ffi.HipSweep(a.scheme, [(1, 0, a.system())], a.n_int, a.p)
    with pytest.raises(storm_amd.StormError, match="unknown option"):
        sweep.set_option("path", 1)
    # bad offsets
    off = np.array([1, 3], dtype=np.uintp)
    f = np.array([1.0, 2.0, 3.0])
    out = np.empty(3)
    rc = L.storm_sweep_det(sweep._h, off.ctypes.data_as(C.POINTER(C.c_size_t)), f.ctypes.data_as(C.POINTER(C.c_double)),
                           out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc != 0 and b"offsets[0]" in L.storm_last_error()
    # capacity: the offsets are still filled in
    grid_f = oracle.linspace(1.0, 12.0, 80)
    off = np.array([0, grid_f.size], dtype=np.uintp)
    ooff = np.zeros(2, dtype=np.uintp)
    br = (ffi.Bracket * 1)()
    rc = L.storm_sweep_scan_and_optimize(sweep._h, off.ctypes.data_as(C.POINTER(C.c_size_t)),
                                         grid_f.ctypes.data_as(C.POINTER(C.c_double)), 4, br, 1,
                                         ooff.ctypes.data_as(C.POINTER(C.c_size_t)), None, None, None)
    ref, _, _ = a.scan_and_optimize(grid_f, 4, threads=8)
    assert rc != 0 and int(ooff[1]) == len(ref) > 1


@pytest.mark.parametrize("name,ell,m,rot,scheme", [
    ("zams", 0, 0, None, "colloc4"), ("zams", 2, 1, 0.1, "colloc2"), ("tams", 1, -1, 0.05, "magnus2"),
    ("zams", 3, 0, lambda x: 0.05 + 0.1 * np.exp(-3.0 * x), "colloc6")])
def test_lane_per_mode_eigen_kernels(oracle, golden_dir, name, ell, m, rot, scheme):
    """The lane-per-mode elimination / back-substitution (picked for launches of >= 1920 modes, forced here) against the
    oracle: 150 frequencies (three waves, the last one ragged), roots and non-roots alike."""
    import storm_amd
    model = rotating(oracle, golden_dir, name, rot)
    o = oracle.Solver(model, ell, m, scheme, model.field("r_coord"))
    f = oracle.linspace(0.7, 21.3, 150)
    rdet, rvec = o.eigenvector(f, threads=16)
    h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
    for layout in (2, 1):
        h.set_option("eigen_layout", layout)
        det, vec = h.eigenvector(f)
        assert same_bits(det, rdet), layout
        assert same_bits(vec, rvec), layout


def test_lane_per_mode_tiny_grids_and_sweep(oracle):
    """n_int = 1, 2, 3, 5 (the 4-deep factor prefetch of the back-substitution clamps) and the sweep path with
    per-lane systems."""
    import storm_amd
    from storm_amd import ffi
    model = oracle.Model.poly0(5.0 / 3.0)
    for npts in (2, 3, 4, 6, 9):
        grid = oracle.linspace(0.0, 1.0, npts)
        solvers = [oracle.Solver(model, ell, 0, "colloc2", grid) for ell in (0, 1, 2, 3)]
        f = [oracle.linspace(1.0 + ell, 9.0, 5 + ell) for ell in range(4)]
        sweep = ffi.HipSweep(solvers[0].scheme, [(o.ell, o.m, o.system()) for o in solvers], solvers[0].n_int, solvers[0].p)
        sweep.set_option("eigen_layout", 2)
        got = sweep.eigenvector(f)
        for k, o in enumerate(solvers):
            rdet, rvec = o.eigenvector(f[k], threads=4)
            assert same_bits(got[k][1], rvec), (npts, k)
            assert same_bits(got[k][0], rdet), (npts, k)


def test_full_size_configs_4_and_5_as_one_sweep(oracle, golden_dir):
    """BASELINE configs 4 and 5 (tams, rot = 0.1, Colloc4, 10^4 frequencies per scan; config 5 on the period-linear
    grid) as ONE sweep of nine systems, eigenvectors of all ~190 modes with both kernel layouts.  Checked against one
    storm_solver per system (every bit of every result) and against the oracle on a sample."""
    import storm_amd
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    model.set_rotation_constant(0.1)
    grid = model.field("r_coord")
    lin, inv = oracle.linspace(1.0, 25.0, 10000), oracle.rev_linspace(0.5, 1.0, 10000)
    cfg = [(1, -1, lin), (1, 1, lin), (2, -1, lin), (2, 1, lin), (3, -1, lin), (3, 1, lin), (0, 0, inv), (2, 0, inv), (4, 0, inv)]
    solvers = [oracle.Solver(model, ell, m, "colloc4", grid) for ell, m, _ in cfg]
    grids = [g for _, _, g in cfg]
    ulp = oracle.relative_to_ulp(1e-8)
    sweep = sweep_from_oracle(solvers)
    got = sweep.scan_and_optimize(grids, ulp)
    roots = [np.array([b["root"] for b in br]) for br, _, _ in got]
    n_modes = sum(r.size for r in roots)
    assert n_modes >= 150, n_modes
    vecs = sweep.eigenvector(roots)          # wave per mode at this count
    sweep.set_option("eigen_layout", 2)
    lane_vecs = sweep.eigenvector(roots)     # what a launch of >= 1920 modes takes
    for a, b in zip(vecs, lane_vecs):
        assert same_bits(a[0], b[0]) and same_bits(a[1], b[1])
    rng = np.random.default_rng(17)
    for k, o in enumerate(solvers):
        h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
        ref, rdets, rinv = h.scan_and_optimize(grids[k], ulp)
        assert same_bits(got[k][1], rdets) and got[k][2] == rinv, k
        assert same_brackets(ref, got[k][0]), k
        if roots[k].size:
            rdet, rvec = h.eigenvector(roots[k])
            assert same_bits(vecs[k][1], rvec) and same_bits(vecs[k][0], rdet), k
            sel = rng.choice(roots[k].size, min(3, roots[k].size), replace=False)
            odet, ovec = o.eigenvector(roots[k][sel], threads=16)
            assert same_bits(vecs[k][1][sel], ovec) and same_bits(vecs[k][0][sel], odet), k
        idx = np.unique(np.concatenate([rng.choice(10000, 128, replace=False),
                                        np.array([b["index"] for b in got[k][0]], dtype=np.int64)])).astype(np.int64)
        assert same_bits(o.det(grids[k][idx], threads=16), got[k][1][idx]), k
    assert got[6][0] == []                   # no radial mode below the fundamental


def test_resident_fields_and_coupling_integrals(oracle, golden_dir):
    """post-process + perturb-deformed with the fields staying in HBM: storm_sweep_modes_resident, then
    storm_sweep_coupling_matrices on a subset of the modes (two systems, interleaved order), against the host-fields
    entry point and the oracle."""
    import storm_amd
    from storm_amd import ffi
    model = rotating(oracle, golden_dir, "zams", 0.08)
    grid = model.field("r_coord")
    cfg = [(1, 1), (0, 0), (3, 1), (2, 0), (2, 1)]
    solvers = [oracle.Solver(model, ell, m, "colloc4", grid) for ell, m in cfg]
    sweep = sweep_from_oracle(solvers)
    grids = [oracle.linspace(2.0, 13.0, 70)] * len(cfg)
    got = sweep.scan_and_optimize(grids, oracle.relative_to_ulp(1e-9))
    roots = [np.array([b["root"] for b in br]) for br, _, _ in got]
    post = storm_amd.HipPostproc({k: model.field(k) for k in storm_amd.MODEL_FIELDS})
    posts = [post] * len(cfg)
    full = sweep.modes(posts, roots, want_fields=True)
    res = sweep.modes_resident(posts, roots)
    off = np.concatenate([[0], np.cumsum([r.size for r in roots])])
    for u in range(len(cfg)):
        assert res[u]["ids"].tolist() == full[u]["ids"].tolist()
    metric = model.perturb_structure(0.08)
    # the m = 1 modes, systems 4, 0, 2 in that order (not the resident order)
    pick = [(4, k) for k in range(roots[4].size)] + [(0, k) for k in range(roots[0].size)] + [(2, k) for k in range(roots[2].size)]
    idx = np.array([off[u] + k for u, k in pick])
    ells = np.array([cfg[u][0] for u, _ in pick], dtype=np.uint64)
    freqs = np.array([roots[u][k] for u, k in pick])
    fields = np.stack([full[u]["fields"][k] for u, k in pick])
    mats = sweep.coupling_matrices_resident(post, idx, ells, 1, metric)
    host_path = post.coupling_matrices(ells, freqs, fields, 1, metric)
    ref, _ = model.coupling_matrices(ells, freqs, fields, 1, metric, threads=16)
    assert idx.size >= 8
    for k in ref:
        assert same_bits(mats[k], host_path[k]) and same_bits(mats[k], ref[k]), k
    # a later modes call drops the resident set
    sweep.modes(posts, roots)
    with pytest.raises(storm_amd.StormError, match="no resident fields"):
        sweep.coupling_matrices_resident(post, idx, ells, 1, metric)
    with pytest.raises(storm_amd.StormError, match="staging cap"):
        sweep.set_option("steps_bytes_max", 1 << 20)
        sweep.modes_resident(posts, roots)


def test_sweep_of_many_systems_uploads_in_slabs(oracle, golden_dir):
    """900 systems (more than one 256 MB upload slab of zams/Colloc2 planes), all three kernel classes interleaved."""
    from storm_amd import ffi
    models = [rotating(oracle, golden_dir, "zams", rot) for rot in (None, 0.05, lambda x: 0.02 + 0.03 * x)]
    grid = models[0].field("r_coord")
    combos = [(0, 0), (1, 0), (1, 1), (2, -1), (2, 2), (3, 0)]
    distinct = [oracle.Solver(mdl, ell, m, "colloc2", grid) for mdl in models for ell, m in combos]
    systems = [(distinct[k % len(distinct)].ell, distinct[k % len(distinct)].m, distinct[k % len(distinct)].system()) for k in range(900)]
    sweep = ffi.HipSweep(distinct[0].scheme, systems, distinct[0].n_int, distinct[0].p)
    freqs = [np.array([2.0 + 0.01 * k, 5.5, 9.25 - 0.003 * k]) for k in range(900)]
    got = sweep.det(freqs)
    for k in list(range(0, 900, 37)) + [899]:
        assert same_bits(got[k], distinct[k % len(distinct)].det(freqs[k])), k
    for k in range(len(distinct), 900):        # the middle frequency is the same for every copy of a system
        assert bits(got[k][1]) == bits(got[k % len(distinct)][1]), k


def test_sweep_invalid_numbers_are_per_system(oracle, oracle_models):
    """A non-finite discriminant (frequency 0 with ell != 0 divides by zero like the reference) drops the pair and raises
    the flag of THAT system only (reference src/dynamic_interface.rs:130-137,153-157)."""
    model = oracle_models("zams")
    grid = model.field("r_coord")
    solvers = [oracle.Solver(model, ell, 0, "colloc2", grid) for ell in (2, 1, 0, 3)]
    grids = [np.array([0.0, 1.0, 2.0, 3.0, 4.0, 5.0]), oracle.linspace(1.0, 6.0, 9), np.array([0.0, 2.0, 4.0, 6.0]),
             np.array([1.5, 0.0, 2.5, 3.5, 4.5])]
    sweep = sweep_from_oracle(solvers)
    got = sweep.scan_and_optimize(grids, 1)
    flags = []
    for k, o in enumerate(solvers):
        ref, rdets, rinv = o.scan_and_optimize(grids[k], 1)
        br, dets, inv = got[k]
        assert np.array_equal(np.isnan(rdets), np.isnan(dets)) and np.array_equal(np.isinf(rdets), np.isinf(dets)), k
        fin = np.isfinite(rdets)
        assert same_bits(rdets[fin], dets[fin]) and inv == rinv and same_brackets(ref, br), k
        flags.append(inv)
    assert flags[0] and not flags[1]      # the sweep does not smear one system's warning over the others
