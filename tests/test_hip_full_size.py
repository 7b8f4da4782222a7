"""GPU parity tests at BASELINE.json's FULL sizes, against the oracle (not against another build of the kernels):

  config 1  zams, ell = 2, Magnus2, 1000 frequencies
  config 2  tams, ell = 1..3 x {Magnus6, Colloc8}, 10^4 frequencies -- all six scans
  config 3  10^4-point n = 3 polytrope AND Polytrope0 on linspace(0, 1, 10^4), ell = 2, Colloc4 (+ Magnus2),
            10^5 frequencies -- the bench headline

For each: the parity build must equal the oracle bit for bit on EVERY grid point (discriminants, bracket indices, end
points, roots, evals); the FMA build ("arithmetic" = 1) must keep north_star's contract against the ORACLE's values:
identical sign pattern and bracket index list, |det_fast - det_ref| <= 1e-9 x the local scale, roots within 1e-10.
The oracle passes cost 10^8..10^9 evaluations each (10-20 s on 16 host threads), so they are shared per module.
Also here: the two halves of scan_and_optimize as separate C-ABI calls, the caller-stream entry point right after
create, and the strong-scaling path with every rank driving the HIP library.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DET_TOL, ROOT_TOL = 1e-9, 1e-10
XI1_N3 = 6.89684861937


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def same_bits(a, b):
    return np.array_equal(bits(a), bits(b))


def local_scale(d):
    a = np.abs(d)
    return np.maximum.reduce([np.roll(a, 1), a, np.roll(a, -1)])


def hip_from_oracle(o, arithmetic=0):
    import storm_amd
    h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
    h.set_option("arithmetic", arithmetic)
    return h


def check_parity_full(o, f, ulp, ref):
    """Parity build vs the oracle's full scan `ref` = (brackets, dets, invalid): everything bit-exact."""
    rres, rdets, rinv = ref
    h = hip_from_oracle(o, 0)
    gres, gdets, ginv = h.scan_and_optimize(f, ulp)
    assert same_bits(rdets, gdets)
    assert rinv == ginv
    assert [r["index"] for r in rres] == [g["index"] for g in gres]
    for k in ("lo_x", "lo_f", "hi_x", "hi_f", "root"):
        assert same_bits([r[k] for r in rres], [g[k] for g in gres]), k
    assert [r["evals"] for r in rres] == [g["evals"] for g in gres]
    return h, gres


def check_fast_full(o, f, ulp, ref, det_tol=DET_TOL):
    """FMA build vs the oracle's full scan: north_star's tolerance contract."""
    rres, rdets, rinv = ref
    h = hip_from_oracle(o, 1)
    fres, fdets, finv = h.scan_and_optimize(f, ulp)
    assert finv == rinv
    fin = np.isfinite(rdets)
    assert np.array_equal(fin, np.isfinite(fdets))
    assert np.array_equal(np.signbit(rdets[fin]), np.signbit(fdets[fin]))                    # sign changes bit-exact
    assert [b["index"] for b in fres] == [b["index"] for b in rres]                          # bracket indices bit-exact
    err = np.abs(fdets[fin] - rdets[fin]) / local_scale(np.where(fin, rdets, 0.0))[fin]
    assert np.all(err <= det_tol), err.max()
    rr, fr = np.array([b["root"] for b in rres]), np.array([b["root"] for b in fres])
    if rr.size:
        assert np.all(np.abs(fr - rr) <= ROOT_TOL * np.abs(rr)), np.max(np.abs(fr - rr) / np.abs(rr))
    return h, fres


# ---------------------------------------------------------------------------------------------- config 3
@pytest.fixture(scope="module")
def config3_polytrope(oracle):
    model = oracle.Model.polytrope(3.0, 5.0 / 3.0, XI1_N3 / 9999.0)
    grid = model.grid(1)
    assert 9990 <= grid.size <= 10010
    o = oracle.Solver(model, 2, 0, "colloc4", grid)
    f = oracle.linspace(1.0, 25.0, 100000)
    ulp = oracle.relative_to_ulp(1e-8)
    return o, f, ulp, o.scan_and_optimize(f, ulp, threads=16)     # 10^9 evaluations on the CPU


def test_config3_polytrope_full_grid_parity_build_bit_exact(config3_polytrope):
    o, f, ulp, ref = config3_polytrope
    assert len(ref[0]) > 10
    h, gres = check_parity_full(o, f, ulp, ref)
    roots = np.array([g["root"] for g in gres])
    sel = np.arange(len(roots))[:: max(1, len(roots) // 4)]
    rdet, rvec = o.eigenvector(roots[sel], threads=16)
    gdet, gvec = h.eigenvector(roots)
    assert same_bits(rdet, gdet[sel]) and same_bits(rvec, gvec[sel])


def one_ulp_data_sensitivity(oracle, o, f, roots):
    """How far the ORACLE's own results move when one input field moves by ONE ulp per point (seeded random signs): the
    conditioning of the reference's value with respect to its data.  Returns (discriminants: max |d' - d| / local scale over
    `f`, eigenvectors: max-norm relative change at `roots`)."""
    ref = o.det(f, threads=16)
    scale = local_scale(ref)
    _, rvec = o.eigenvector(roots, threads=16)
    rng = np.random.default_rng(1)
    det_s, vec_s = 0.0, 0.0
    for name in ("a_star", "gamma1"):
        m2 = oracle.Model.polytrope(3.0, 5.0 / 3.0, XI1_N3 / 9999.0)
        v = m2.field(name)
        up = rng.integers(0, 2, v.size).astype(bool)
        m2.set_field(name, np.where(up, np.nextafter(v, np.inf), np.nextafter(v, -np.inf)))
        o2 = oracle.Solver(m2, o.ell, o.m, o.scheme, m2.grid(1))
        det_s = max(det_s, float(np.max(np.abs(o2.det(f, threads=16) - ref) / scale)))
        _, v2 = o2.eigenvector(roots, threads=16)
        vec_s = max(vec_s, float(np.max(np.max(np.abs(v2 - rvec), axis=1) / np.max(np.abs(rvec), axis=1))))
    return det_s, vec_s


def test_config3_polytrope_full_grid_fast_build_brackets_identical(oracle, config3_polytrope):
    o, f, ulp, ref = config3_polytrope
    # The bar on THIS model is derived, not chosen: towards the surface of the 10^4-point n = 3 polytrope V and A* grow like
    # 1/(1 - x) (1.8e8 at the last point) and the reference's own discriminant is determined by its input data only to a few
    # 1e-7 -- ONE ulp on a_star or gamma1 moves the oracle's value by 3-4e-7 of the local scale (one ulp on the frequency:
    # 7e-13).  No arithmetic that rounds differently can agree better than that, whatever its formulation:
    # profiles/r03_attribution_config3.txt has the FMA build with each re-formulation (LU stage solve, Newton reciprocal,
    # Laplace-expansion inverse, folded wrapper) switched off in turn and with the reference's formulas under contraction
    # alone -- 4.0e-7 to 5.8e-7 every time, 4.0e-7 as shipped.  So: bar = max(1e-9, 4 x the measured one-ulp sensitivity).
    # The factor is multiplicative in the discriminant, so it vanishes at a root: signs and bracket indices are identical
    # and the roots agree to 1.1e-12 (bound asserted: 1e-10).
    roots = np.array([b["root"] for b in ref[0]])
    sel = np.arange(len(roots))[:: max(1, len(roots) // 4)]
    det_s, vec_s = one_ulp_data_sensitivity(oracle, o, oracle.linspace(1.0, 25.0, 2000), roots[sel])
    assert 1e-8 < det_s < 2e-6, det_s          # (if this model ever becomes well-conditioned the plain contract applies)
    h, fres = check_fast_full(o, f, ulp, ref, det_tol=max(DET_TOL, 4.0 * det_s))
    _, rvec = o.eigenvector(roots[sel], threads=16)
    _, fvec = h.eigenvector(roots[sel])
    err = np.max(np.abs(fvec - rvec), axis=1) / np.max(np.abs(rvec), axis=1)
    assert np.all(err <= max(1e-7, 4.0 * vec_s)), (err.max(), vec_s)   # (1e-7 on the other models)


@pytest.mark.parametrize("scheme", ["colloc4", "magnus2"])
def test_config3_polytrope0_full_grid(oracle, scheme):
    """SURVEY 8(d) config 3's second model: Polytrope0 on linspace(0, 1, 10^4), 10^5 frequencies.  Magnus2 is FINITE here
    (analytic coefficients; on the n = 3 polytrope V -> inf at the surface overflows the exponential in the reference
    arithmetic too, tests/test_hip_api.py).  The GPU runs the full grid; the oracle a 4096-point sample, every bracket end
    and the refinement of every bracket (a second full 10^9-evaluation CPU pass would double the suite's time)."""
    model = oracle.Model.poly0(5.0 / 3.0)
    o = oracle.Solver(model, 2, 0, scheme, oracle.linspace(0.0, 1.0, 10000))
    n = 100000
    f = oracle.linspace(1.0, 25.0, n)
    ulp = oracle.relative_to_ulp(1e-8)
    h = hip_from_oracle(o, 0)
    gres, gdets, ginv = h.scan_and_optimize(f, ulp)
    rng = np.random.default_rng(21)
    idx = np.unique(np.concatenate([rng.choice(n, 4096, replace=False), [g["index"] for g in gres],
                                    [g["index"] + 1 for g in gres]])).astype(np.int64)
    ref = o.det(f[idx], threads=16)
    assert same_bits(ref, gdets[idx])
    assert np.all(np.isfinite(gdets)) and not ginv and len(gres) >= 10
    sgn = np.signbit(gdets)
    assert [g["index"] for g in gres] == np.nonzero(sgn[:-1] != sgn[1:])[0].tolist()
    for g in gres:
        r = o.optimize(f[g["index"]], gdets[g["index"]], f[g["index"] + 1], gdets[g["index"] + 1], ulp)
        assert same_bits([r["root"], r["lo_x"], r["hi_x"]], [g["root"], g["lo_x"], g["hi_x"]]) and r["evals"] == g["evals"]
    # FMA build: against the oracle on the sample, against the (sample-verified) parity build on the rest
    fast = hip_from_oracle(o, 1)
    fres, fdets, _ = fast.scan_and_optimize(f, ulp)
    assert np.array_equal(np.signbit(fdets[idx]), np.signbit(ref))
    assert np.array_equal(np.signbit(fdets), sgn) and [b["index"] for b in fres] == [g["index"] for g in gres]
    assert np.all(np.abs(fdets - gdets) <= DET_TOL * local_scale(gdets))
    gr, fr = np.array([g["root"] for g in gres]), np.array([b["root"] for b in fres])
    assert np.all(np.abs(fr - gr) <= ROOT_TOL * np.abs(gr))


# ---------------------------------------------------------------------------------------------- config 2
@pytest.mark.parametrize("scheme", ["magnus6", "colloc8"])
@pytest.mark.parametrize("ell", [1, 2, 3])
def test_config2_all_six_scans_full_grid(oracle, golden_dir, scheme, ell):
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    o = oracle.Solver(model, ell, 0, scheme, model.field("r_coord"))
    f = oracle.linspace(1.0, 25.0, 10000)
    ulp = oracle.relative_to_ulp(1e-8)
    ref = o.scan_and_optimize(f, ulp, threads=16)       # 6.2e7 evaluations
    assert len(ref[0]) > 10
    check_parity_full(o, f, ulp, ref)
    check_fast_full(o, f, ulp, ref)


# ---------------------------------------------------------------------------------------------- configs 4 and 5
def rotating_tams(oracle, golden_dir):
    model = oracle.Model.from_gsm_npz(os.path.join(golden_dir, "gsm_tams.npz"))
    model.set_rotation_constant(0.1)        # `set-rotation-constant 0.1`, src/main.rs:785-788
    return model


@pytest.mark.parametrize("m", [-1, 1])
@pytest.mark.parametrize("ell", [1, 2, 3])
def test_config4_rotating_full_grid_both_builds(oracle, golden_dir, ell, m):
    """BASELINE config 4: tams with rot = 0.1, ell = 1..3, m = -1, +1, Colloc4, 10^4 frequencies: the rot != 0 branch of
    add_frequency and of the inner boundary (src/system/adiabatic.rs:136-148,164-168).  Parity build bit-exact on every
    point; FMA build: identical sign pattern and bracket index list, roots <= 1e-10, against the ORACLE."""
    model = rotating_tams(oracle, golden_dir)
    o = oracle.Solver(model, ell, m, "colloc4", model.field("r_coord"))
    f = oracle.linspace(1.0, 25.0, 10000)
    ulp = oracle.relative_to_ulp(1e-8)
    ref = o.scan_and_optimize(f, ulp, threads=16)
    assert len(ref[0]) > 10
    check_parity_full(o, f, ulp, ref)
    check_fast_full(o, f, ulp, ref)


@pytest.mark.parametrize("ell", [0, 2, 4])
def test_config5_dense_g_mode_range_inverse_grid_both_builds(oracle, golden_dir, ell):
    """BASELINE config 5's scans: tams with rot = 0.1, ell = 0, 2, 4, m = 0, Colloc4, the period-linear grid
    rev_linspace(0.5, 1.0, 10^4) of `scan --inverse` (src/main.rs:816-817,1236-1238) over the dense g-mode range -- where
    a discriminant at rounding level on a grid point is likeliest.  Same bars as config 4."""
    model = rotating_tams(oracle, golden_dir)
    o = oracle.Solver(model, ell, 0, "colloc4", model.field("r_coord"))
    f = oracle.rev_linspace(0.5, 1.0, 10000)
    ulp = oracle.relative_to_ulp(1e-8)
    ref = o.scan_and_optimize(f, ulp, threads=16)
    assert len(ref[0]) == {0: 0, 2: 9, 4: 19}[ell]   # (ell = 0 has no g modes: its scan comes back empty on both sides)
    check_parity_full(o, f, ulp, ref)
    check_fast_full(o, f, ulp, ref)


# ---------------------------------------------------------------------------------------------- config 1
def test_config1_zams_magnus2_1000_frequencies(oracle, oracle_models):
    model = oracle_models("zams")
    o = oracle.Solver(model, 2, 0, "magnus2", model.field("r_coord"))
    f = oracle.linspace(1.0, 25.0, 1000)
    ulp = oracle.relative_to_ulp(1e-8)
    ref = o.scan_and_optimize(f, ulp, threads=16)
    h, gres = check_parity_full(o, f, ulp, ref)
    check_fast_full(o, f, ulp, ref)
    roots = np.array([g["root"] for g in gres])
    rdet, rvec = o.eigenvector(roots, threads=16)
    gdet, gvec = h.eigenvector(roots)
    assert same_bits(rdet, gdet) and same_bits(rvec, gvec)


# ---------------------------------------------------------------------------------------------- C-ABI additions
def test_scan_brackets_then_optimize_equals_scan_and_optimize(oracle, oracle_models):
    import storm_amd
    model = oracle_models("zams")
    o = oracle.Solver(model, 1, 0, "colloc4", model.field("r_coord"))
    h = hip_from_oracle(o)
    f = oracle.linspace(1.0, 25.0, 300)
    ulp = oracle.relative_to_ulp(1e-8)
    whole, wdets, winv = h.scan_and_optimize(f, ulp)
    found, fdets, finv = h.scan_brackets(f)
    assert same_bits(wdets, fdets) and winv == finv
    assert [b["index"] for b in found] == [b["index"] for b in whole]
    assert all(np.isnan(b["root"]) and b["evals"] == 0 for b in found)
    assert same_bits([b["lo_x"] for b in found], f[[b["index"] for b in found]])
    # refine them in two unequal shares, shuffled: every bracket is independent of its neighbours
    order = np.random.default_rng(3).permutation(len(found))
    a, b = order[: len(order) // 3], order[len(order) // 3:]
    got = {}
    for share in (a, b):
        for k, r in zip(share, h.optimize_brackets([found[k] for k in share], ulp)):
            got[int(k)] = r
    for k, w in enumerate(whole):
        for key in ("lo_x", "lo_f", "hi_x", "hi_f", "root"):
            assert same_bits([w[key]], [got[k][key]]), (k, key)
        assert w["evals"] == got[k]["evals"]
    assert h.optimize_brackets([], ulp) == []
    with pytest.raises(storm_amd.StormError) as e:      # equal signs: the reference panics
        h.optimize_brackets([dict(lo_x=1.0, lo_f=1.0, hi_x=2.0, hi_f=2.0)], ulp)
    assert e.value.code == -6
    with pytest.raises(storm_amd.StormError) as e:
        h.optimize_brackets([dict(lo_x=1.0, lo_f=float("nan"), hi_x=2.0, hi_f=2.0)], ulp)
    assert e.value.code == -6


def test_caller_stream_immediately_after_create(oracle, oracle_models):
    """The planes are uploaded asynchronously on the library's stream by storm_solver_create; a launch on a caller's
    stream right after create must wait for them (an event, not a host synchronisation).  Both kernel families:
    large batch (det_hex) and small batch (step-matrix staging shared between streams)."""
    import torch
    model = oracle_models("tams")
    o = oracle.Solver(model, 2, 0, "colloc4", model.field("r_coord"))
    system = o.system()
    import storm_amd
    for n in (2048, 48):
        f = oracle.linspace(1.0, 25.0, n)
        ref = o.det(f, threads=16)
        d_f = torch.from_numpy(f).cuda()
        torch.cuda.synchronize()
        for _ in range(3):
            st = torch.cuda.Stream()
            d_out = torch.zeros_like(d_f)
            torch.cuda.synchronize()
            h = storm_amd.HipSolver(o.scheme, o.ell, o.m, system, o.n_int, o.p)     # create ...
            with torch.cuda.stream(st):
                h.det_device(d_f.data_ptr(), n, d_out.data_ptr(), stream=st.cuda_stream)   # ... and launch at once
            # the same handle on its own stream while the caller's stream may still be running
            own = h.det(f[: max(1, n // 2)])
            st.synchronize()
            assert same_bits(d_out.cpu().numpy(), ref)
            assert same_bits(own, ref[: max(1, n // 2)])
            h.close()


# ---------------------------------------------------------------------------------------------- strong scaling, HIP on every rank
def _strong_hip_worker(rank, world, port, backend, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from storm_amd import distributed as D, ffi, host
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_zams.npz"))
        es = host.ErasedSolver(model, 2, 0, "colloc4", model.field("r_coord"))
        h = ffi.HipSolver.borrow(es.raw_handle(), owner=es)
        f = host.linspace(1.0, 25.0, 4000)
        vecs = []

        def eig(brackets):
            if brackets:
                vecs.append(h.eigenvector(np.array([b["root"] for b in brackets]))[1])

        where = torch.device("cuda", dev) if backend == "nccl" else "cpu"
        # deal=False first (one collective, every rank refines its own slice's brackets; its eigenvectors are not kept)
        own, inv1, own_mine = D.strong_scan_and_optimize(h, f, ffi.relative_to_ulp(1e-8), device=where, records=True, deal=False)
        np.save(os.path.join(out_dir, f"own{rank}.npy"), own)
        np.save(os.path.join(out_dir, f"ownmine{rank}.npy"), own_mine)
        res, inv, mine = D.strong_scan_and_optimize(h, f, ffi.relative_to_ulp(1e-8), device=where, after_refine=eig)
        np.save(os.path.join(out_dir, f"hip{rank}.npy"), D.pack(res, 0))
        np.save(os.path.join(out_dir, f"vec{rank}.npy"), vecs[0] if vecs else np.zeros((0, 0)))
        np.save(os.path.join(out_dir, f"idx{rank}.npy"), np.array([b["index"] for b in mine], dtype=np.int64))
    finally:
        dist.destroy_process_group()


def test_strong_scaling_with_hip_on_every_rank(tmp_path, oracle, oracle_models):
    """Two ranks, each driving libstorm_hip (RCCL when two GPUs are visible, else gloo with both ranks on the one GPU):
    the gathered bracket list and the dealt eigenvectors must equal the oracle's single-process results bit for bit."""
    import torch
    import torch.multiprocessing as mp
    from storm_amd import distributed as D
    world = 2
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_strong_hip_worker, args=(world, port, backend, str(tmp_path)), nprocs=world, join=True)
    model = oracle_models("zams")
    o = oracle.Solver(model, 2, 0, "colloc4", model.field("r_coord"))
    ref, _, _ = o.scan_and_optimize(oracle.linspace(1.0, 25.0, 4000), oracle.relative_to_ulp(1e-8), threads=16)
    ref_rec = D.pack(ref, 0)
    assert len(ref) > 10
    _, rvec = o.eigenvector(np.array([r["root"] for r in ref]), threads=16)
    for r in range(world):
        got = np.load(tmp_path / f"hip{r}.npy")
        assert same_bits(got, ref_rec), f"rank {r}"
        share = D.deal(len(ref), world, r)
        assert np.load(tmp_path / f"idx{r}.npy").tolist() == [ref[k]["index"] for k in share]
        assert same_bits(np.load(tmp_path / f"vec{r}.npy"), rvec[share])
        # the one-collective protocol: the same list on every rank, and the rank's own slice's part of it
        assert same_bits(np.load(tmp_path / f"own{r}.npy"), ref_rec), f"rank {r}"
        lo, hi = D.shard_bounds(4000, world, r)
        assert same_bits(np.load(tmp_path / f"ownmine{r}.npy"), ref_rec[(ref_rec[:, 6] >= lo) & (ref_rec[:, 6] < hi)]), f"rank {r}"


def test_exchange_path_over_rccl_one_rank():
    """The N > 1 exchange (storm_amd/distributed.py: pinned mirrors, device tensors, all_gather_into_tensor) with RCCL as
    the backend.  One rank is all a one-GPU box can give RCCL (it refuses two ranks on one device); the communicator, the
    collective form and the buffers are the ones N ranks use.  Own process: the test session must not own a process group."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(36000 + os.getpid() % 2000))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rccl_one_rank.py")], env=env, capture_output=True,
                       text=True, timeout=150)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ok " in r.stdout and "gather_brackets over RCCL" in r.stdout
