"""GPU tests of the C-ABI surface beyond plain parity: device-pointer entry point, error codes,
thread safety, counters, and the full-size shapes of BASELINE configs 3-5."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype != np.float64:
        return np.array_equal(a, b)
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def hip_from_oracle(o):
    import storm_amd
    return storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)


def test_device_pointer_entry_point(oracle, oracle_models):
    import torch
    model = oracle_models("zams")
    o = oracle.Solver(model, 2, 0, "colloc4", model.field("r_coord"))
    h = hip_from_oracle(o)
    f = oracle.linspace(1.0, 25.0, 777)
    d_f = torch.from_numpy(f).cuda()
    d_out = torch.empty_like(d_f)
    torch.cuda.synchronize()
    for path in (0, 1, 2, 3, 4):
        h.set_option("path", path)
        d_out.zero_()
        torch.cuda.synchronize()
        h.det_device(d_f.data_ptr(), f.size, d_out.data_ptr())          # library stream, synchronous
        assert np.array_equal(bits(d_out.cpu().numpy()), bits(o.det(f, threads=16))), path
    # caller-provided stream: asynchronous, ordered with the caller's other work on that stream
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        d_out.zero_()
        h.det_device(d_f.data_ptr(), f.size, d_out.data_ptr(), stream=st.cuda_stream)
        doubled = d_out * 2.0
    st.synchronize()
    assert np.array_equal(bits(doubled.cpu().numpy()), bits(o.det(f, threads=16) * 2.0))


def test_singular_stage_matrix_is_reported(oracle):
    """A collocation stage whose (jump - I) is exactly singular: the reference panics in
    try_inverse().unwrap() (src/stepper.rs:88); the C ABI returns STORM_ERR_SINGULAR."""
    import storm_amd
    n_int = 5
    sys_ = dict(u=np.zeros(n_int), v_gamma=np.zeros(n_int), a_star=np.zeros(n_int), c1=np.ones(n_int),
                rot=np.zeros(n_int), delta_over_x=np.ones(n_int), inner_c1=1.0, inner_rot=0.0, outer_u=3.0)
    h = storm_amd.HipSolver("colloc2", 0, 0, sys_, n_int, 1)
    for path in (1, 2, 3, 4):
        h.set_option("path", path)
        with pytest.raises(storm_amd.StormError) as e:
            h.det(np.array([1.0, 2.0, 3.0]))
        assert e.value.code == -3, path
    # a regular system on the same handle type still works afterwards
    sys_["delta_over_x"] = np.full(n_int, 0.01)
    h2 = storm_amd.HipSolver("colloc2", 0, 0, sys_, n_int, 1)
    assert np.all(np.isfinite(h2.det(np.array([1.0, 2.0]))))


def test_concurrent_calls_on_one_handle_and_many_handles(oracle, oracle_models):
    """The reference calls its closures from rayon workers (Sync); calls on one handle are
    serialised inside the library, independent handles run concurrently."""
    model = oracle_models("zams")
    o = oracle.Solver(model, 1, 0, "magnus4", model.field("r_coord"))
    f = oracle.linspace(1.0, 20.0, 96)
    ref = o.det(f, threads=16)
    shared = hip_from_oracle(o)
    own = [hip_from_oracle(o) for _ in range(4)]
    errors = []

    def work(k):
        try:
            for _ in range(5):
                a = shared.det(f[k::4])
                b = own[k].det(f)
                if not np.array_equal(bits(a), bits(ref[k::4])) or not np.array_equal(bits(b), bits(ref)):
                    errors.append(k)
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors


def test_counters(oracle, oracle_models):
    model = oracle_models("zams")
    o = oracle.Solver(model, 0, 0, "colloc2", model.field("r_coord"))
    h = hip_from_oracle(o)
    h.stats(reset=True)
    res, _, _ = h.scan_and_optimize(oracle.linspace(1.0, 25.0, 25), 1)
    st = h.stats()
    assert st["det_evals"] == 25 + sum(r["evals"] for r in res)
    assert st["refine_evals"] == sum(r["evals"] for r in res)
    assert st["refine_rounds"] == max(r["evals"] for r in res)
    h.eigenvector(np.array([r["root"] for r in res]))
    assert h.stats()["eigenvectors"] == len(res)


@pytest.mark.parametrize("scheme,ell,ulp_rel", [("colloc4", 2, 1e-8), ("magnus2", 0, None), ("colloc6", 1, 1e-12)])
def test_device_state_machine_equals_host_state_machine(oracle, oracle_models, scheme, ell, ulp_rel):
    """Option device_brackets: 1 (default) = sign-swap compaction + InverseQuadratic state machine as kernels, 0 = the host
    loop of round 1.  Same brackets, end points, roots, evals and counters, and both equal the oracle's."""
    model = oracle_models("tams")
    o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
    f = oracle.linspace(1.0, 25.0, 700)
    ulp = 1 if ulp_rel is None else oracle.relative_to_ulp(ulp_rel)
    ref, rdets, rinv = o.scan_and_optimize(f, ulp, threads=16)
    out = {}
    for dev in (0, 1):
        h = hip_from_oracle(o)
        h.set_option("device_brackets", dev)
        h.stats(reset=True)
        res, dets, inv = h.scan_and_optimize(f, ulp)
        st = h.stats()
        assert same_bits(dets, rdets) and inv == rinv
        assert [b["index"] for b in res] == [b["index"] for b in ref]
        for key in ("lo_x", "lo_f", "hi_x", "hi_f", "root"):
            assert same_bits([b[key] for b in res], [b[key] for b in ref]), (dev, key)
        assert [b["evals"] for b in res] == [b["evals"] for b in ref]
        out[dev] = (st["det_evals"], st["refine_evals"], st["refine_rounds"])
        assert st["refine_evals"] == sum(b["evals"] for b in ref) and st["refine_rounds"] == max(b["evals"] for b in ref)
    assert out[0] == out[1]


def test_full_size_config3_polytrope_1e5_frequencies(oracle):
    """BASELINE config 3 on one GPU: synthetic 10^4-point n=3 polytrope, ell=2, 10^5 frequencies,
    Colloc4 (CLI default) and Magnus2.  Sampled oracle comparison + size-independent properties."""
    # step chosen so that the Lane-Emden integration yields ~10^4 points (xi_1 = 6.8968...)
    model = oracle.Model.polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
    grid = model.grid(1)
    assert 9990 <= grid.size <= 10010
    n = 100000
    f = oracle.linspace(1.0, 25.0, n)
    rng = np.random.default_rng(7)
    for scheme in ("colloc4", "magnus2"):
        o = oracle.Solver(model, 2, 0, scheme, grid)
        h = hip_from_oracle(o)
        ulp = oracle.relative_to_ulp(1e-8)
        gres, gdets, ginv = h.scan_and_optimize(f, ulp)          # auto path: lane-per-frequency kernel
        idx = np.unique(np.concatenate([rng.choice(n, 256, replace=False), np.array([g["index"] for g in gres], dtype=np.int64),
                                        np.array([g["index"] + 1 for g in gres], dtype=np.int64)])).astype(np.int64)
        ref = o.det(f[idx], threads=16)
        if scheme == "magnus2":
            # On this grid the outermost intervals have |A| ~ 1e8 (V -> inf at the surface): the T18
            # exponential overflows in the reference arithmetic as well, every discriminant is NaN and the
            # scan reports "invalid numbers" with no brackets.  Parity = the same non-finite pattern.
            assert np.array_equal(np.isnan(ref), np.isnan(gdets[idx]))
            assert np.all(np.isnan(gdets)) and ginv and gres == []
            continue
        assert np.array_equal(bits(ref), bits(gdets[idx])), scheme
        sgn = np.signbit(gdets)
        fin = np.isfinite(gdets)
        expect = np.nonzero((sgn[:-1] != sgn[1:]) & fin[:-1] & fin[1:])[0].tolist()
        assert [g["index"] for g in gres] == expect
        assert len(gres) > 10
        h.set_option("path", 4)
        sub = slice(40000, 40000 + 4096)
        assert np.array_equal(bits(h.det(f[sub])), bits(gdets[sub]))
        for g in gres[:: max(1, len(gres) // 6)]:
            r = o.optimize(f[g["index"]], gdets[g["index"]], f[g["index"] + 1], gdets[g["index"] + 1], ulp)
            assert np.array_equal(bits([r["root"]]), bits([g["root"]])) and r["evals"] == g["evals"]


def test_full_size_config4_rotating(oracle, golden_dir):
    """BASELINE config 4: tams with rot = 0.1, ell = 1..3, m = +-1, Colloc4, 10^4 frequencies."""
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    model.set_rotation_constant(0.1)
    grid = model.field("r_coord")
    f = oracle.linspace(1.0, 25.0, 10000)
    rng = np.random.default_rng(11)
    ulp = oracle.relative_to_ulp(1e-8)
    for ell, m in ((1, -1), (2, 1), (3, -1)):
        o = oracle.Solver(model, ell, m, "colloc4", grid)
        h = hip_from_oracle(o)
        gres, gdets, _ = h.scan_and_optimize(f, ulp)
        idx = np.unique(np.concatenate([rng.choice(f.size, 256, replace=False),
                                        np.array([g["index"] for g in gres], dtype=np.int64)])).astype(np.int64)
        assert np.array_equal(bits(o.det(f[idx], threads=16)), bits(gdets[idx])), (ell, m)
        roots = np.array([g["root"] for g in gres])
        sel = np.arange(len(roots))[:: max(1, len(roots) // 3)]
        rdet, rvec = o.eigenvector(roots[sel], threads=16)
        gdet, gvec = h.eigenvector(roots)
        assert np.array_equal(bits(rdet), bits(gdet[sel])) and np.array_equal(bits(rvec), bits(gvec[sel]))


def test_full_size_config5_g_modes(oracle, golden_dir):
    """BASELINE config 5 scan part: rot = 0.1, ell = 0, 2, 4, period-linear grid (--inverse) of 10^4
    points in the dense g-mode range, then eigenvectors of the modes found."""
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    model.set_rotation_constant(0.1)
    grid = model.field("r_coord")
    f = oracle.rev_linspace(0.5, 1.0, 10000)
    rng = np.random.default_rng(13)
    ulp = oracle.relative_to_ulp(1e-8)
    for ell in (0, 2, 4):
        o = oracle.Solver(model, ell, 0, "colloc4", grid)
        h = hip_from_oracle(o)
        gres, gdets, ginv = h.scan_and_optimize(f, ulp)
        idx = np.unique(np.concatenate([rng.choice(f.size, 256, replace=False),
                                        np.array([g["index"] for g in gres], dtype=np.int64)])).astype(np.int64)
        assert np.array_equal(bits(o.det(f[idx], threads=16)), bits(gdets[idx])), ell
        if ell == 0:
            assert gres == []      # no radial mode below the fundamental
        else:
            assert len(gres) >= 5  # g modes of the TAMS model in [0.5, 1]
        if gres:
            roots = np.array([g["root"] for g in gres])
            sel = np.arange(len(roots))[:: max(1, len(roots) // 3)]
            _, rvec = o.eigenvector(roots[sel], threads=16)
            _, gvec = h.eigenvector(roots)
            assert np.array_equal(bits(rvec), bits(gvec[sel]))


def test_handle_per_scan_from_many_threads_and_memory_cache(oracle, golden_dir):
    # the rotation-sweep pattern: every thread creates its own solver + post-processing handle,
    # scans, identifies the modes and destroys both; buffers and streams are recycled through the library's cache.
    # Results must equal the sequential ones bit for bit, and trimming the cache must not disturb later calls.
    import storm_amd
    from storm_amd import ffi
    jobs = [((ell, m), 0.01 + 0.003 * k) for k, (ell, m) in enumerate([(0, 0), (1, 1), (2, -1), (2, 0), (3, 0), (1, -1)] * 3)]
    freqs = oracle.linspace(2.0, 14.0, 120)
    ulp = oracle.relative_to_ulp(1e-8)

    def run(job):
        (ell, m), rot = job
        model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_zams.npz"))
        model.set_rotation_constant(rot)
        o = oracle.Solver(model, ell, m, "colloc4", model.field("r_coord"))
        h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
        post = storm_amd.HipPostproc({k: model.field(k) for k in storm_amd.MODEL_FIELDS})
        res, dets, _ = h.scan_and_optimize(freqs, ulp)
        got = h.modes(post, np.array([b["root"] for b in res]), want_vec=True)
        h.close()
        post.close()
        return dets, got["vec"], got["fields"], got["ids"]

    sequential = [run(j) for j in jobs]
    assert ffi.memory_cache_stats()[0] > 0            # destroyed handles left their blocks in the cache
    out = [None] * len(jobs)

    def worker(k):
        out[k] = run(jobs[k])

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for a, b in zip(sequential, out):
        for x, y in zip(a, b):
            assert same_bits(x, y)
    ffi.memory_cache_trim()
    assert ffi.memory_cache_stats() == (0, 0)
    again = run(jobs[0])
    for x, y in zip(sequential[0], again):
        assert same_bits(x, y)
    ffi.memory_cache_limit(0, 0)                       # nothing may stay cached: every release goes back to HIP
    run(jobs[1])
    assert ffi.memory_cache_stats() == (0, 0)
    ffi.memory_cache_limit(16 << 30, 2 << 30)


def test_buffers_of_every_size_band_are_recycled_through_the_slabs(oracle, oracle_models):
    # The library carves its device and pinned blocks out of slabs (three size bands: below 1 MiB, 1-64 MiB, above).
    # Three handles take turns with batches whose buffers grow and shrink through all bands (800 B to 240 MB: 3 x 10^7
    # frequencies would be too slow to check, so the large band comes from eigenvector staging of 40 modes at 2 MB each);
    # every result must stay bit-identical to the oracle's whatever block a buffer landed in, idle bytes must show in the
    # statistics and a trim must give everything back.
    from storm_amd import ffi
    model = oracle_models("tams")
    grid = model.field("r_coord")
    solvers = []
    for ell, scheme in ((1, "colloc4"), (2, "magnus2"), (3, "colloc2")):
        o = oracle.Solver(model, ell, 0, scheme, grid)
        solvers.append((o, hip_from_oracle(o)))
    ffi.memory_cache_trim()
    for n in (100, 40000, 700, 200000, 5000, 100):
        for o, h in solvers:
            f = oracle.linspace(1.0 + 0.01 * o.ell, 24.0, n)
            got = h.det(f)
            pick = np.unique(np.linspace(0, n - 1, 97).astype(int))
            assert same_bits(got[pick], o.det(f[pick], threads=16)), (o.scheme, n)
    roots = oracle.linspace(2.0, 20.0, 40)
    for o, h in solvers[:2]:
        det, vec = h.eigenvector(roots)
        rdet, rvec = o.eigenvector(roots, threads=16)
        assert same_bits(det, rdet) and same_bits(vec, rvec)
    for _, h in solvers:
        h.close()
    dev, pinned = ffi.memory_cache_stats()
    assert dev >= (64 << 20) and pinned > 0        # the slabs stay, with everything in them idle
    ffi.memory_cache_trim()
    assert ffi.memory_cache_stats() == (0, 0)
    o, _ = solvers[0]
    h = hip_from_oracle(o)                          # and the library carries on from nothing
    f = oracle.linspace(1.0, 24.0, 300)
    assert same_bits(h.det(f), o.det(f, threads=16))
    h.close()


def test_eigenvectors_into_a_pinned_result_block(oracle, oracle_models):
    # a caller's pinned (page-locked) result block is filled by DMA, a pageable one through the library's staging block:
    # same bits either way, and only the requested part of the block is written
    import torch
    model = oracle_models("zams")
    o = oracle.Solver(model, 2, 0, "colloc4", model.field("r_coord"))
    h = hip_from_oracle(o)
    roots = oracle.linspace(3.0, 12.0, 7)
    rdet, rvec = o.eigenvector(roots, threads=8)
    length = (o.n_int + 1) * 4
    pinned = torch.full((9 * length,), -1.0, dtype=torch.float64).pin_memory().numpy()
    det, vec = h.eigenvector(roots, out=pinned)
    assert same_bits(det, rdet) and same_bits(vec, rvec)
    assert np.all(pinned[7 * length:] == -1.0)
    pageable = np.full(9 * length, -1.0)
    det2, vec2 = h.eigenvector(roots, out=pageable)
    assert same_bits(det2, rdet) and same_bits(vec2, rvec) and np.all(pageable[7 * length:] == -1.0)
    with pytest.raises(ValueError):
        h.eigenvector(roots, out=np.empty(5))
    h.close()
