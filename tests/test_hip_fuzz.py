"""Seeded sweep over the corners the fixed-size tests do not visit: random sub-grids of a fixture (2 to ~300 points, so every
ragged tail of every kernel: intervals per wave, segments, the last block of a plane), random degree / order / rotation
profile / scheme, random frequency counts and every discriminant path, single handle and the device bracket machine.
Parity build: bit-exact against the oracle (discriminants, bracket lists, eigenvectors).  FMA build: the contract of
tests/test_hip_fast_mode.py on the same inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCHEMES = ["colloc2", "colloc4", "colloc6", "colloc8", "magnus2", "magnus4", "magnus6", "magnus8"]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same_bits(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def random_case(rng, oracle, oracle_models):
    name = ["zams", "tams"][int(rng.integers(2))]
    model = oracle_models(name)
    full = model.field("r_coord")
    n_pts = int(rng.choice([2, 3, 4, 5, 8, 9, 17, 33, 63, 64, 65, 66, 97, 128, 129, 200, 257, 301]))
    if rng.random() < 0.5:
        pick = np.sort(rng.choice(np.arange(1, full.size - 1), size=n_pts - 2, replace=False))
        grid = np.concatenate([[full[0]], full[pick], [full[-1]]])          # model points: exact grid hits
    else:
        inner = np.sort(rng.uniform(full[0], full[-1], size=n_pts - 2))
        grid = np.concatenate([[full[0]], inner, [full[-1]]])               # interpolated abscissae
        grid = np.unique(grid)
    kind = int(rng.integers(3))
    if kind == 0:
        model.set_rotation_constant(0.0)
    elif kind == 1:
        model.set_rotation_constant(float(rng.uniform(0.01, 0.3)))
    else:
        x = full / full[-1]
        model.set_field("rot", 0.05 + 0.2 * rng.random() * (1.0 - 0.5 * x * x))   # differential rotation
    ell = int(rng.integers(0, 5))
    m = int(rng.integers(-ell, ell + 1)) if ell else 0
    scheme = SCHEMES[int(rng.integers(8))]
    o = oracle.Solver(model, ell, m, scheme, grid)
    n_freq = int(rng.choice([1, 2, 3, 4, 5, 7, 16, 63, 64, 65, 130, 257]))
    lo = float(rng.uniform(0.8, 6.0))
    f = np.sort(rng.uniform(lo, lo + rng.uniform(0.5, 20.0), size=n_freq))
    return o, f, (name, n_pts, kind, ell, m, scheme, n_freq)


@pytest.mark.parametrize("seed", range(6))
def test_random_corners_parity_build(oracle, oracle_models, seed):
    import storm_amd
    rng = np.random.default_rng(1000 + seed)
    for _ in range(12):
        o, f, what = random_case(rng, oracle, oracle_models)
        ref = o.det(f, threads=8)
        h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
        for path in (0, 2, 4, 3, 1):
            h.set_option("path", path)
            assert same_bits(h.det(f), ref), (what, "path", path)
        h.set_option("path", 0)
        if f.size >= 2 and np.all(np.isfinite(ref)):
            ulp = storm_amd.relative_to_ulp(1e-9)
            rres, rdets, rinv = o.scan_and_optimize(f, ulp, threads=8)
            for dev in (1, 0):
                h.set_option("device_brackets", dev)
                got, dets, inv = h.scan_and_optimize(f, ulp)
                assert same_bits(dets, rdets) and inv == rinv, (what, dev)
                assert [b["index"] for b in got] == [b["index"] for b in rres], (what, dev)
                for key in ("lo_x", "lo_f", "hi_x", "hi_f", "root"):
                    assert same_bits([b[key] for b in got], [b[key] for b in rres]), (what, dev, key)
                assert [b["evals"] for b in got] == [b["evals"] for b in rres], (what, dev)
            if rres:
                roots = np.array([b["root"] for b in rres])[:5]
                rdet, rvec = o.eigenvector(roots, threads=8)
                gdet, gvec = h.eigenvector(roots)
                assert same_bits(gdet, rdet) and same_bits(gvec, rvec), what
        h.close()
    oracle_models("zams").set_rotation_constant(0.0)
    oracle_models("tams").set_rotation_constant(0.0)


@pytest.mark.parametrize("seed", range(6))
def test_random_corners_fma_build(oracle, oracle_models, seed):
    """|det_fast - det_ref| <= 1e-9 of the largest |det_ref| among a grid point and its neighbours (the collocation schemes on
    interpolated, very uneven random sub-grids: 1e-8, their stage systems are worse conditioned there), identical signs where
    the reference value is not at rounding level, for the automatic path and the forced segment path."""
    import storm_amd
    rng = np.random.default_rng(2000 + seed)
    for _ in range(16):
        o, f, what = random_case(rng, oracle, oracle_models)
        ref = o.det(f, threads=8)
        # (grids of a handful of points put the whole star into one or two intervals: |A| h ~ 10^2-10^4 there, the stage
        # systems are conditioned accordingly and the tolerance contract is not about such discretisations; the parity test
        # above covers them bit for bit)
        if what[1] < 33 or not np.all(np.isfinite(ref)):
            continue
        h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
        h.set_option("arithmetic", 1)
        scale = np.maximum.reduce([np.roll(np.abs(ref), 1), np.abs(ref), np.roll(np.abs(ref), -1)]) if f.size > 2 else np.full(f.size, np.max(np.abs(ref)))
        tol = 1e-8 if what[5].startswith("colloc") else 1e-9
        for segments in (0, 32):
            h.set_option("segments", segments)
            got = h.det(f)
            assert np.all(np.abs(got - ref) <= tol * scale), (what, segments, float(np.max(np.abs(got - ref) / scale)))
            clear = np.abs(ref) > 1e-6 * scale
            assert np.array_equal(np.signbit(got[clear]), np.signbit(ref[clear])), (what, segments)
        h.close()
    oracle_models("zams").set_rotation_constant(0.0)
    oracle_models("tams").set_rotation_constant(0.0)


@pytest.mark.parametrize("seed", range(4))
def test_random_sweeps_equal_the_oracle_system_by_system(oracle, golden_dir, seed):
    """storm_sweep on random sets of systems (1-9 systems of random degree, order and rotation profile -- so a random split
    over the three kernel classes --, one shared random sub-grid, ragged frequency lists with empty ones): discriminants,
    brackets and eigenvectors of every system are the oracle's, bit for bit (parity build)."""
    import os
    from storm_amd import ffi
    import storm_amd
    rng = np.random.default_rng(3000 + seed)
    for _ in range(5):
        name = ["zams", "tams"][int(rng.integers(2))]
        n_sys = int(rng.integers(1, 10))
        models, specs = [], []
        for _k in range(n_sys):
            mdl = oracle.Model.from_gsm_npz(os.path.join(golden_dir, f"gsm_{name}.npz"))
            kind = int(rng.integers(3))
            if kind == 1:
                mdl.set_rotation_constant(float(rng.uniform(0.01, 0.3)))
            elif kind == 2:
                x = mdl.field("r_coord") / mdl.field("r_coord")[-1]
                mdl.set_field("rot", 0.05 + 0.2 * rng.random() * (1.0 - 0.5 * x * x))
            ell = int(rng.integers(0, 4))
            models.append(mdl)
            specs.append((ell, int(rng.integers(-ell, ell + 1)) if ell else 0))
        full = models[0].field("r_coord")
        n_pts = int(rng.choice([2, 5, 17, 64, 65, 130, 301]))
        pick = np.sort(rng.choice(np.arange(1, full.size - 1), size=n_pts - 2, replace=False))
        grid = np.concatenate([[full[0]], full[pick], [full[-1]]])
        scheme = SCHEMES[int(rng.integers(8))]
        solvers = [oracle.Solver(mdl, ell, m, scheme, grid) for mdl, (ell, m) in zip(models, specs)]
        lists = []
        for _k in range(n_sys):
            n_f = int(rng.choice([0, 1, 2, 3, 5, 8, 31, 70]))
            lo = float(rng.uniform(0.8, 5.0))
            lists.append(np.sort(rng.uniform(lo, lo + 15.0, size=n_f)))
        sweep = ffi.HipSweep(scheme, [(o.ell, o.m, o.system()) for o in solvers], solvers[0].n_int, solvers[0].p)
        what = (name, n_sys, specs, n_pts, scheme, [x.size for x in lists])
        dets = sweep.det(lists)
        refs = [o.det(x, threads=4) if x.size else np.empty(0) for o, x in zip(solvers, lists)]
        for k in range(n_sys):
            assert same_bits(dets[k], refs[k]), (what, k)
        if all(np.all(np.isfinite(r)) for r in refs):
            ulp = storm_amd.relative_to_ulp(1e-9)
            got = sweep.scan_and_optimize(lists, ulp)
            roots = []
            for k, o in enumerate(solvers):
                rres, rdets, rinv = o.scan_and_optimize(lists[k], ulp, threads=4) if lists[k].size else ([], np.empty(0), False)
                res, d, inv = got[k]
                assert same_bits(d, rdets) and bool(inv) == bool(rinv), (what, k)
                # (bits, not ==: on a grid of a few points a refinement can end in NaN -- in the oracle and here alike)
                assert same_bits([b["root"] for b in res], [b["root"] for b in rres]) and [b["evals"] for b in res] == [b["evals"] for b in rres], (what, k)
                rr = np.array([b["root"] for b in rres])
                roots.append(rr[np.isfinite(rr)][:3])
            vec = sweep.eigenvector(roots)
            for k, o in enumerate(solvers):
                if roots[k].size:
                    rdet, rvec = o.eigenvector(roots[k], threads=4)
                    assert same_bits(vec[k][0], rdet) and same_bits(vec[k][1], rvec), (what, k)
        sweep.close()
