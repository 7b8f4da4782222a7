"""Frequency-slice sharding of a scan over the GPUs of one node (one process per GPU).

The reference runs one rayon task per scan frequency (src/dynamic_interface.rs:125-127); the two
`collect()` barriers there are the only global points.  Here every rank evaluates a contiguous
slice of the frequency grid plus ONE halo point on the right, so every adjacent pair (i, i+1) is
owned by exactly the rank that owns i; brackets are refined where they are found and the
(tiny, latency-bound) bracket lists are gathered with one all_gather of padded records behind a
header row (a second one only if a rank found more than GATHER_CAP) -- RCCL over xGMI on GPUs (`backend="nccl"`), gloo in the CPU tests.  There is no
collective inside the data path.
"""
import numpy as np
import torch
import torch.distributed as dist

RECORD = 7  # lo_x, lo_f, hi_x, hi_f, root, evals, global index


def shard_bounds(n, world, rank):
    """Owned index range [lo, hi) of rank `rank`; the slice to evaluate is [lo, min(hi + 1, n))."""
    per = -(-n // world) if n else 0
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


def local_slice(freqs, world, rank):
    lo, hi = shard_bounds(len(freqs), world, rank)
    return lo, freqs[lo:min(hi + 1, len(freqs))]


def pack(brackets, offset):
    rec = np.zeros((len(brackets), RECORD), dtype=np.float64)
    for k, b in enumerate(brackets):
        rec[k] = (b["lo_x"], b["lo_f"], b["hi_x"], b["hi_f"], b["root"], float(b["evals"]), float(b["index"] + offset))
    return rec


def unpack(rec):
    return [dict(lo_x=r[0], lo_f=r[1], hi_x=r[2], hi_f=r[3], root=r[4], evals=int(r[5]), index=int(r[6]))
            for r in rec]


GATHER_CAP = 64  # bracket records per rank that fit the first exchange (a scan finds a few dozen modes in all)


class _Exchange:
    """The buffers of one all_gather shape, allocated once per (device, shape) and reused from scan to scan: the send and
    receive tensors on the collective's device and, for a GPU, pinned host mirrors, so an exchange is one async H2D copy,
    the collective and one async D2H copy on torch's current stream, then one stream synchronisation."""

    def __init__(self, device, world, shape):
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.h_send = torch.zeros(shape, dtype=torch.float64, pin_memory=self.on_gpu)
        self.h_recv = torch.zeros((world,) + tuple(shape), dtype=torch.float64, pin_memory=self.on_gpu)
        self.send = torch.zeros(shape, dtype=torch.float64, device=self.device) if self.on_gpu else self.h_send
        self.recv = torch.zeros((world,) + tuple(shape), dtype=torch.float64, device=self.device) if self.on_gpu else self.h_recv
        self.parts = None
        self.send_np, self.recv_np = self.h_send.numpy(), self.h_recv.numpy()

    def start(self):
        """Puts the all_gather of send_np (filled by the caller) in flight and returns: host work done before finish()
        -- the eigenvectors of this rank's modes on the solver's own stream -- overlaps the exchange."""
        if self.on_gpu:
            self.send.copy_(self.h_send, non_blocking=True)
        if _single_tensor_form():
            self.work = dist.all_gather_into_tensor(self.recv, self.send, async_op=True)
        else:
            if self.parts is None:
                self.parts = list(self.recv.unbind(0))   # views into recv: the list form writes in place
            self.work = dist.all_gather(self.parts, self.send, async_op=True)

    def finish(self):
        """-> recv_np [world, *shape] of the exchange start() put in flight"""
        self.work.wait()     # (a GPU backend: torch's current stream waits for the collective's)
        self.work = None
        if self.on_gpu:
            self.h_recv.copy_(self.recv, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        return self.recv_np

    def run(self):
        """all_gather of send_np (filled by the caller) -> recv_np [world, *shape]"""
        self.start()
        return self.finish()


_EXCHANGES = {}
_FORM = {}


def _single_tensor_form():
    """Which all_gather the process group's backend gets, decided ONCE per process group before the first collective: the
    single-tensor form on RCCL (`nccl`), the list form elsewhere (gloo in the CPU tests).  No try/except around a
    collective: an error inside one is a real error, and answering it with a different collective would leave the ranks in
    mismatched calls."""
    key = dist.get_backend()
    if key not in _FORM:
        _FORM[key] = key == "nccl" and hasattr(dist, "all_gather_into_tensor")
    return _FORM[key]


def _exchange(device, shape):
    key = (str(device), dist.get_world_size(), tuple(shape), id(dist.group.WORLD))
    ex = _EXCHANGES.get(key)
    if ex is None:
        if len(_EXCHANGES) > 16:   # (shapes other than the two standing ones are rare: do not keep them all)
            _EXCHANGES.clear()
        ex = _EXCHANGES[key] = _Exchange(device, dist.get_world_size(), shape)
    return ex


def _all_gather_rows(buf, device):
    """all_gather of one equally shaped float64 array per rank -> array [world, *buf.shape] on the host (a view of the
    exchange's receive buffer: valid until the next exchange of the same shape)."""
    ex = _exchange(device, buf.shape)
    ex.send_np[...] = buf
    return ex.run()


def gather_brackets(local_records, invalid, device, meanwhile=None):
    """all_gather of per-rank bracket records -> (global records in ascending order, any_invalid).

    ONE collective in the usual case: every rank sends GATHER_CAP padded records behind a header row (count, invalid
    flag); only if some rank found more is a second exchange made, sized by the largest count.  `meanwhile()` runs while
    the first exchange is in flight."""
    n = local_records.shape[0]
    ex = _exchange(device, (GATHER_CAP + 1, RECORD))
    buf = ex.send_np
    buf[0, 0], buf[0, 1] = float(n), 1.0 if invalid else 0.0
    if n:
        buf[1:1 + min(n, GATHER_CAP)] = local_records[:GATHER_CAP]
    ex.start()
    if meanwhile is not None:
        meanwhile()
    got = ex.finish()
    counts = got[:, 0, 0].astype(np.int64)
    any_invalid = bool(got[:, 0, 1].any())
    if counts.max(initial=0) > GATHER_CAP:
        cap = int(counts.max())
        big = np.zeros((cap, RECORD))
        big[:n] = local_records
        got = _all_gather_rows(big, device)
        parts = [got[r, :counts[r]] for r in range(got.shape[0])]
    else:
        parts = [got[r, 1:1 + counts[r]] for r in range(got.shape[0])]
    return (np.concatenate(parts, axis=0) if parts else np.zeros((0, RECORD))), any_invalid


def sharded_scan_and_optimize(solver, freqs, ulp, device="cpu"):
    """scan_and_optimize of the whole grid, sharded over the ranks of the default process group.

    `solver` needs .scan_and_optimize(freqs, ulp) -> (brackets with 'index', dets, invalid).
    Returns (all brackets in ascending order, any_invalid, this rank's local brackets)."""
    freqs = np.ascontiguousarray(freqs, dtype=np.float64)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        res, _, inv = solver.scan_and_optimize(freqs, ulp)
        return res, inv, res
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, mine = local_slice(freqs, world, rank)
    res, _, inv = solver.scan_and_optimize(mine, ulp) if len(mine) else ([], None, False)
    rec, any_inv = gather_brackets(pack(res, lo), inv, device)
    return unpack(rec), any_inv, res


def deal(n_brackets, world, rank):
    """Global bracket indices rank `rank` refines: dealt out cyclically by global bracket index, so a slice
    of the grid that is dense in modes (g modes) does not serialise on the rank that scanned it."""
    return list(range(rank, n_brackets, world))


PHASES = ("slice_scan", "gather_found", "refine", "after_refine", "gather_refined")


def strong_scan_and_optimize(solver, freqs, ulp, device="cpu", after_refine=None, timings=None, records=False, deal=True):
    """scan_and_optimize of ONE frequency grid split over the ranks of the default process group (strong scaling).

    Every rank evaluates a contiguous slice (+1 halo point) and finds its sign swaps.  Then one of two protocols:

    deal=True: the unrefined brackets are gathered (one all_gather of padded records behind a header row),
    dealt out cyclically, refined where they were dealt (`solver.optimize_brackets`), and the refined records are gathered
    again (one all_gather: the deal fixes every rank's count); `after_refine(my_brackets)` runs between refinement and
    the last gather (the eigenvectors of this rank's modes).  Two collectives; a slice that is dense in modes does not
    serialise on the rank that scanned it.

    deal=False: every rank refines the brackets of its own slice in the same library call that scanned it
    (`solver.scan_and_optimize`: nothing crosses PCIe between the scan and the refinement), then ONE all_gather of the
    refined records, in flight while `after_refine(my_brackets)` computes this rank's eigenvectors.  One collective and no
    wait between scan and refinement: what a scan with a few dozen modes wants (refinement rounds and eigenvectors are
    latency-bound: their time hardly depends on how many brackets a rank holds).  Same records either way, bit for bit.

    Returns (all brackets in ascending order, any_invalid, this rank's brackets) -- lists of dicts, or with `records` the
    [n, RECORD] arrays themselves (no per-bracket Python at all).  `timings`: a dict that receives this rank's seconds per
    phase (PHASES), accumulated over calls (deal=False: slice_scan includes the refinement, after_refine runs inside
    gather_refined's flight and is listed on its own, gather_refined is what is left to wait for).

    `solver` needs .scan_brackets(freqs) -> (brackets with 'index', dets, invalid) and
    .optimize_brackets(brackets, ulp) -> brackets (deal=False: .scan_and_optimize(freqs, ulp) -> the same triple, refined);
    if it has the *_records forms (ffi.HipSolver) those are used instead and the records go from the C struct array to the
    exchange buffer as numpy columns."""
    import time
    freqs = np.ascontiguousarray(freqs, dtype=np.float64)
    fast = hasattr(solver, "scan_brackets_records") and hasattr(solver, "optimize_records")
    clock = [time.perf_counter()]

    def lap(phase):
        if timings is not None:
            now = time.perf_counter()
            timings[phase] = timings.get(phase, 0.0) + now - clock[0]
            clock[0] = now

    def scan(f, offset):
        if len(f) < 2:
            return np.zeros((0, RECORD)), False
        if fast:
            return solver.scan_brackets_records(f, offset)
        found, _, inv = solver.scan_brackets(f)
        return pack(found, offset), inv

    def refine(rec):
        if not rec.shape[0]:
            return rec
        if fast:
            return solver.optimize_records(rec, ulp)
        return pack(solver.optimize_brackets(unpack(rec), ulp), 0)

    give = (lambda rec: rec) if records else unpack
    if not dist.is_initialized() or dist.get_world_size() == 1:
        rec, inv = scan(freqs, 0)
        lap("slice_scan")
        rec = refine(rec)
        lap("refine")
        out = give(rec)
        if after_refine is not None:
            after_refine(out)
        lap("after_refine")
        return out, inv, out
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, mine = local_slice(freqs, world, rank)
    if not deal:
        if len(mine) < 2:
            my, inv = np.zeros((0, RECORD)), False
        elif hasattr(solver, "scan_and_optimize_records"):
            my, inv = solver.scan_and_optimize_records(mine, ulp, lo)
        else:
            res, _, inv = solver.scan_and_optimize(mine, ulp)
            my = pack(res, lo)
        lap("slice_scan")
        my_out = give(my)

        def eigenvectors():
            lap("gather_refined")      # (the few microseconds of putting the exchange in flight)
            if after_refine is not None:
                after_refine(my_out)
            lap("after_refine")

        out, any_inv = gather_brackets(my, inv, device, meanwhile=eigenvectors)
        lap("gather_refined")
        return give(out), any_inv, my_out
    found, inv = scan(mine, lo)
    lap("slice_scan")
    rec, any_inv = gather_brackets(found, inv, device)
    lap("gather_found")
    n_all = rec.shape[0]
    my = refine(rec[rank::world])          # the cyclic deal (deal())
    lap("refine")
    my_out = give(my)
    if after_refine is not None:
        after_refine(my_out)
    lap("after_refine")
    # second exchange: every rank's count follows from the deal, so one all_gather of padded records is enough
    cap = max(1, -(-n_all // world))
    ex = _exchange(device, (cap, RECORD))
    ex.send_np[:my.shape[0]] = my
    got = ex.run()
    out = np.empty((n_all, RECORD))
    for r in range(world):
        cnt = len(range(r, n_all, world))
        out[r::world] = got[r, :cnt]
    lap("gather_refined")
    return give(out), any_inv, my_out


# ---- rotation sweeps (storm_sweep): systems are independent, so they are dealt out whole ----
def shard_systems(n_sys, world, rank):
    """Indices of the systems rank `rank` runs: contiguous blocks (systems of one rotation rate sit next to each other
    and share their model interpolation), sizes differing by at most one."""
    base, extra = divmod(n_sys, world)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


def sharded_sweep_scan(make_sweep, n_sys, freq_grids, scan, device="cpu"):
    """Runs scan_and_optimize of n_sys systems over the ranks of the default process group and returns, on every rank,
    the per-system bracket lists of ALL systems (caller order).

    make_sweep(indices) builds this rank's sweep object over the given systems; scan(sweep, grids) returns one bracket
    list (dicts with lo_x, lo_f, hi_x, hi_f, root, evals) per system.  The only collective is the gather of the
    results: two all_gathers (counts, padded records), as in gather_brackets."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        mine = list(range(n_sys))
        return scan(make_sweep(mine), [freq_grids[u] for u in mine]) if n_sys else []
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = shard_systems(n_sys, world, rank)
    local = scan(make_sweep(mine), [freq_grids[u] for u in mine]) if mine else []
    rows = [(float(u), b["lo_x"], b["lo_f"], b["hi_x"], b["hi_f"], b["root"], float(b["evals"]))
            for u, br in zip(mine, local) for b in br]
    rec = np.array(rows, dtype=np.float64).reshape(len(rows), RECORD)
    allrec, _ = gather_brackets(rec, False, device)
    out = [[] for _ in range(n_sys)]
    for r in allrec:
        out[int(r[0])].append(dict(lo_x=r[1], lo_f=r[2], hi_x=r[3], hi_f=r[4], root=r[5], evals=int(r[6])))
    return out
