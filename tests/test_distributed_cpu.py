"""world_size-2 (and 3) gloo tests of the frequency-slice sharding, on CPU.  The discriminant
backend here is the oracle (test-only): what is under test is the slicing, the halo, and the
bracket gather, which must reproduce the single-process bracket list exactly."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_freq, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import storm_ref
    from storm_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if n_freq % 2:  # the odd case also takes the second exchange (more brackets on a rank than the first one holds)
            D.GATHER_CAP = 1
        model = storm_ref.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_zams.npz"))
        solver = storm_ref.Solver(model, 1, 0, "colloc2", model.field("r_coord"))
        freqs = storm_ref.linspace(1.0, 25.0, n_freq)
        res, inv, local = D.sharded_scan_and_optimize(solver, freqs, 1, device="cpu")
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), D.pack(res, 0))
        np.save(os.path.join(out_dir, f"local{rank}.npy"), np.array([len(local), int(inv)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_freq", [(2, 80), (3, 41)])
def test_sharded_scan_matches_single_process(tmp_path, oracle, oracle_models, world, n_freq):
    from storm_amd import distributed as D
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, n_freq, str(tmp_path)), nprocs=world, join=True)
    model = oracle_models("zams")
    solver = oracle.Solver(model, 1, 0, "colloc2", model.field("r_coord"))
    ref, _, _ = solver.scan_and_optimize(oracle.linspace(1.0, 25.0, n_freq), 1, threads=8)
    ref_rec = D.pack(ref, 0)
    total_local = 0
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert np.array_equal(got.view(np.uint64), ref_rec.view(np.uint64)), f"rank {r}"
        total_local += int(np.load(tmp_path / f"local{r}.npy")[0])
    assert total_local == len(ref)  # every bracket found by exactly one rank


class _OracleHalves:
    """Test-only stand-in for HipSolver.scan_brackets / optimize_brackets: the oracle (what is under test is the slicing,
    the two gathers and the cyclic deal of the brackets)."""

    def __init__(self, solver):
        self.s = solver

    def scan_brackets(self, freqs):
        d = self.s.det(freqs)
        sgn, fin = np.signbit(d), np.isfinite(d)
        swap = (sgn[:-1] != sgn[1:]) | np.isnan(d[:-1]) | np.isnan(d[1:])
        ok = fin[:-1] & fin[1:]
        res = [dict(lo_x=freqs[i], lo_f=d[i], hi_x=freqs[i + 1], hi_f=d[i + 1], root=float("nan"), evals=0, index=int(i))
               for i in np.nonzero(swap & ok)[0]]
        return res, d, bool(np.any(swap & ~ok))

    def optimize_brackets(self, brackets, ulp):
        out = []
        for b in brackets:
            r = self.s.optimize(b["lo_x"], b["lo_f"], b["hi_x"], b["hi_f"], ulp)
            r["index"] = b["index"]
            out.append(r)
        return out


def _strong_worker(rank, world, port, n_freq, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import storm_ref
    from storm_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = storm_ref.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_zams.npz"))
        solver = _OracleHalves(storm_ref.Solver(model, 1, 0, "colloc2", model.field("r_coord")))
        freqs = storm_ref.linspace(1.0, 25.0, n_freq)
        seen = []
        res, inv, mine = D.strong_scan_and_optimize(solver, freqs, 1, device="cpu", after_refine=seen.extend)
        assert [b["index"] for b in seen] == [b["index"] for b in mine]
        np.save(os.path.join(out_dir, f"strong{rank}.npy"), D.pack(res, 0))
        np.save(os.path.join(out_dir, f"mine{rank}.npy"), np.array([b["index"] for b in mine], dtype=np.int64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_freq", [(2, 80), (3, 41), (2, 3)])
def test_strong_scaling_scan_deals_brackets_and_matches_single_process(tmp_path, oracle, oracle_models, world, n_freq):
    from storm_amd import distributed as D
    port = 33500 + (os.getpid() % 2000) + world + n_freq
    mp.spawn(_strong_worker, args=(world, port, n_freq, str(tmp_path)), nprocs=world, join=True)
    model = oracle_models("zams")
    solver = oracle.Solver(model, 1, 0, "colloc2", model.field("r_coord"))
    ref, _, _ = solver.scan_and_optimize(oracle.linspace(1.0, 25.0, n_freq), 1, threads=8)
    ref_rec = D.pack(ref, 0)
    dealt = []
    for r in range(world):
        got = np.load(tmp_path / f"strong{r}.npy")
        assert np.array_equal(got.view(np.uint64), ref_rec.view(np.uint64)), f"rank {r}"
        mine = np.load(tmp_path / f"mine{r}.npy").tolist()
        assert mine == [ref[k]["index"] for k in D.deal(len(ref), world, r)]   # cyclic by global bracket index
        dealt += mine
    assert sorted(dealt) == [b["index"] for b in ref]  # every bracket refined by exactly one rank


class _OracleWhole(_OracleHalves):
    """... plus the one-call form the deal=False protocol takes (scan and refinement of a slice in one go)."""

    def scan_and_optimize(self, freqs, ulp):
        found, dets, inv = self.scan_brackets(freqs)
        return self.optimize_brackets(found, ulp), dets, inv


def _owner_worker(rank, world, port, n_freq, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import storm_ref
    from storm_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = storm_ref.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_zams.npz"))
        solver = _OracleWhole(storm_ref.Solver(model, 1, 0, "colloc2", model.field("r_coord")))
        freqs = storm_ref.linspace(1.0, 25.0, n_freq)
        timings = {}
        for records in (False, True):
            seen = []
            res, inv, mine = D.strong_scan_and_optimize(solver, freqs, 1, device="cpu", after_refine=seen.append, timings=timings,
                                                        records=records, deal=False)
            assert len(seen) == 1 and seen[0] is mine      # the eigenvector hook ran once, on this rank's own brackets
        assert set(timings) <= set(D.PHASES) and {"slice_scan", "after_refine", "gather_refined"} <= set(timings)
        np.save(os.path.join(out_dir, f"own{rank}.npy"), res)
        np.save(os.path.join(out_dir, f"ownmine{rank}.npy"), mine)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_freq", [(2, 80), (3, 41), (2, 3), (3, 2)])
def test_strong_scaling_owner_refines_one_collective(tmp_path, oracle, oracle_models, world, n_freq):
    """deal=False: every rank refines what its slice found and ONE all_gather (in flight during the eigenvector hook) puts
    the list together: the single-process record array on every rank, bit for bit, and each rank's own part of it."""
    from storm_amd import distributed as D
    port = 37500 + (os.getpid() % 2000) + world + n_freq
    mp.spawn(_owner_worker, args=(world, port, n_freq, str(tmp_path)), nprocs=world, join=True)
    model = oracle_models("zams")
    solver = oracle.Solver(model, 1, 0, "colloc2", model.field("r_coord"))
    ref, _, _ = solver.scan_and_optimize(oracle.linspace(1.0, 25.0, n_freq), 1, threads=8)
    ref_rec = D.pack(ref, 0)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"own{r}.npy").view(np.uint64), ref_rec.view(np.uint64)), f"rank {r}"
        lo, hi = D.shard_bounds(n_freq, world, r)
        part = ref_rec[(ref_rec[:, 6] >= lo) & (ref_rec[:, 6] < hi)]
        assert np.array_equal(np.load(tmp_path / f"ownmine{r}.npy").view(np.uint64), part.view(np.uint64)), f"rank {r}"


class _OracleRecords(_OracleHalves):
    """... and for ffi.HipSolver.scan_brackets_records / optimize_records (the array form the RCCL path takes)."""

    def scan_brackets_records(self, freqs, offset=0):
        from storm_amd import distributed as D
        res, _, inv = self.scan_brackets(freqs)
        return D.pack(res, offset), inv

    def optimize_records(self, rec, ulp):
        from storm_amd import distributed as D
        return D.pack(self.optimize_brackets(D.unpack(rec), ulp), 0)


def _records_worker(rank, world, port, n_freq, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import storm_ref
    from storm_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = storm_ref.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_zams.npz"))
        solver = _OracleRecords(storm_ref.Solver(model, 1, 0, "colloc2", model.field("r_coord")))
        freqs = storm_ref.linspace(1.0, 25.0, n_freq)
        timings = {}
        for _ in range(2):   # twice: the second scan reuses the exchange buffers of the first
            seen = []
            rec, inv, mine = D.strong_scan_and_optimize(solver, freqs, 1, device="cpu", after_refine=seen.append,
                                                        timings=timings, records=True)
            assert isinstance(rec, np.ndarray) and rec.shape[1] == D.RECORD and seen[0] is mine
        assert set(timings) == set(D.PHASES) and all(v >= 0.0 for v in timings.values())
        np.save(os.path.join(out_dir, f"rec{rank}.npy"), rec)
        np.save(os.path.join(out_dir, f"recmine{rank}.npy"), mine)
    finally:
        dist.destroy_process_group()


def test_strong_scaling_record_arrays_and_phase_timings(tmp_path, oracle, oracle_models):
    from storm_amd import distributed as D
    world, n_freq = 2, 80
    port = 36500 + (os.getpid() % 2000)
    mp.spawn(_records_worker, args=(world, port, n_freq, str(tmp_path)), nprocs=world, join=True)
    model = oracle_models("zams")
    solver = oracle.Solver(model, 1, 0, "colloc2", model.field("r_coord"))
    ref, _, _ = solver.scan_and_optimize(oracle.linspace(1.0, 25.0, n_freq), 1, threads=8)
    ref_rec = D.pack(ref, 0)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"rec{r}.npy").view(np.uint64), ref_rec.view(np.uint64)), f"rank {r}"
        assert np.array_equal(np.load(tmp_path / f"recmine{r}.npy").view(np.uint64), ref_rec[r::world].view(np.uint64))


def test_shard_bounds_cover_every_pair_once():
    from storm_amd import distributed as D
    for n in (0, 1, 2, 7, 64, 100001):
        for world in (1, 2, 3, 8):
            pairs = []
            for r in range(world):
                lo, hi = D.shard_bounds(n, world, r)
                upper = min(hi + 1, n)
                pairs += [(i, i + 1) for i in range(lo, upper - 1)] if n < 1000 else []
                assert 0 <= lo <= hi <= n
            if n < 1000:
                assert pairs == [(i, i + 1) for i in range(max(0, n - 1))]


class _OracleSweep:
    """Test-only stand-in for SolverSweep: the oracle, system by system (what is under test is the dealing and the gather)."""

    def __init__(self, storm_ref, model, systems):
        grid = model.field("r_coord")
        self.solvers = [storm_ref.Solver(model, ell, m, "colloc2", grid) for ell, m in systems]

    def scan(self, grids):
        return [s.scan_and_optimize(g, 1)[0] for s, g in zip(self.solvers, grids)]


_SYSTEMS = [(0, 0), (1, 0), (1, 1), (2, 0), (2, -1)]


def _sweep_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import storm_ref
    from storm_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = storm_ref.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_zams.npz"))
        model.set_rotation_constant(0.05)
        grids = [storm_ref.linspace(1.0 + 0.5 * k, 14.0, 30 + k) for k in range(len(_SYSTEMS))]
        res = D.sharded_sweep_scan(lambda idx: _OracleSweep(storm_ref, model, [_SYSTEMS[u] for u in idx]), len(_SYSTEMS),
                                   grids, lambda sw, g: sw.scan(g))
        np.save(os.path.join(out_dir, f"sweep{rank}.npy"),
                np.array([[u, b["lo_x"], b["hi_x"], b["root"], b["evals"]] for u, br in enumerate(res) for b in br]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sweep_matches_single_process(tmp_path, oracle, world):
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_sweep_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    model = oracle.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_zams.npz"))
    model.set_rotation_constant(0.05)
    grids = [oracle.linspace(1.0 + 0.5 * k, 14.0, 30 + k) for k in range(len(_SYSTEMS))]
    ref = _OracleSweep(oracle, model, _SYSTEMS).scan(grids)
    ref_rows = np.array([[u, b["lo_x"], b["hi_x"], b["root"], b["evals"]] for u, br in enumerate(ref) for b in br])
    assert ref_rows.shape[0] > 5
    for r in range(world):
        got = np.load(tmp_path / f"sweep{r}.npy")
        assert np.array_equal(got.view(np.uint64), ref_rows.view(np.uint64)), f"rank {r}"


def test_shard_systems_partition():
    from storm_amd import distributed as D
    for n in (0, 1, 5, 8, 64, 1001 * 8):
        for world in (1, 2, 3, 8):
            parts = [D.shard_systems(n, world, r) for r in range(world)]
            assert sum(parts, []) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
