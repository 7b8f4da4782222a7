"""What one rank of a strong-scaled headline scan does, timed on one GPU (no collectives): the slice of ceil(1e5/G)+1
frequencies through scan_brackets, the refinement of every G-th bracket, their eigenvectors.  usage: strong_slice_timing.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import storm_amd
from storm_amd import host, ffi, distributed as D
model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
grid = model.grid(1)
es = host.ErasedSolver(model, 2, 0, "colloc4", grid)
h = ffi.HipSolver.borrow(es.raw_handle(), owner=es)
h.set_option("arithmetic", 1)
freqs = host.linspace(1.0, 25.0, 100000)
ulp = storm_amd.relative_to_ulp(1e-8)
allb, _, _ = h.scan_brackets(freqs)
buf = np.empty(64 * (es.n_int + 1) * 4)
for G in (1, 2, 4, 8):
    best = None
    for rank in range(G):
        lo, mine = D.local_slice(freqs, G, rank)
        share = [allb[j] for j in D.deal(len(allb), G, rank)]
        for rep in range(4):
            t0 = time.perf_counter()
            h.scan_brackets(mine)
            t1 = time.perf_counter()
            my = h.optimize_brackets(share, ulp) if share else []
            t2 = time.perf_counter()
            if my:
                h.eigenvector(np.array([b["root"] for b in my]), out=buf)
            t3 = time.perf_counter()
            row = (t3 - t0, t1 - t0, t2 - t1, t3 - t2)
            if rep and (best is None or row[0] > best[0]):   # the slowest rank sets the step (first repetition: warm-up)
                best = row
    print(f"G={G}: slowest rank {best[0]*1e3:.2f} ms per step (scan of the slice {best[1]*1e3:.2f}, refinement {best[2]*1e3:.2f}, "
          f"eigenvectors {best[3]*1e3:.2f}) -> {1e5 * es.n_int / best[0] / 1e10:.2f}e10 evals/s without the two collectives", flush=True)
    # the one-collective protocol (strong_scan_and_optimize(deal=False), what bench.py --gpus N runs): scan + refinement of
    # the slice in ONE library call, then the eigenvectors of the slice's own modes
    own = None
    for rank in range(G):
        lo, mine = D.local_slice(freqs, G, rank)
        for rep in range(4):
            t0 = time.perf_counter()
            rec, _ = h.scan_and_optimize_records(mine, ulp, lo)
            t1 = time.perf_counter()
            if rec.shape[0]:
                h.eigenvector(rec[:, 4].copy(), out=buf)
            t2 = time.perf_counter()
            row = (t2 - t0, t1 - t0, t2 - t1, rec.shape[0])
            if rep and (own is None or row[0] > own[0]):
                own = row
    print(f"G={G}: one-collective protocol: slowest rank {own[0]*1e3:.2f} ms per step (scan + refinement of the slice {own[1]*1e3:.2f}, "
          f"eigenvectors of its {own[3]} modes {own[2]*1e3:.2f}) -> {1e5 * es.n_int / own[0] / 1e10:.2f}e10 evals/s without the collective", flush=True)
