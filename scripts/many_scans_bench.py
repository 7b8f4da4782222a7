"""The rotation-sweep workload of the reference's batch scripts (scripts/generate-batch-input-deformed.py:19-35):
many small scans, each on its own (ell, m, rotation rate) system: solver construction + scan + refinement +
eigenvectors + mode ids per scan.  usage: python scripts/many_scans_bench.py [--scans 64] [--steps 300] [--threads 1 4 8 16]"""
import argparse
import os
import sys

# one hardware queue per concurrent scan (HIP's default of 4 makes streams share queues: a scan's latency-bound
# eigenvector kernel then blocks the scans behind it); must be set before the HIP runtime starts
if int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scans", type=int, default=64)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--scheme", default="colloc4")
ap.add_argument("--keep", action="store_true", help="keep every handle alive until the end (no frees inside the timed region)")
ap.add_argument("--threads", type=int, nargs="*", default=[1, 4, 8, 16])
args = ap.parse_args()

npz = os.path.join(ROOT, "tests", "golden", "gsm_tams.npz")
combos = [(0, 0), (2, 0), (4, 0), (1, 1), (1, -1), (2, 1), (2, -1), (3, 0)]
jobs = [(combos[k % len(combos)], 0.01 + 0.002 * (k // len(combos))) for k in range(args.scans)]
base = host.Model.from_gsm_npz(npz)
grid = base.field("r_coord")
freqs = host.linspace(1.0, 25.0, args.steps)
phase = {}


keep = []
models = []                                         # model loading (HDF5 in the reference) is not what is measured
for (ell, m), rot in jobs:
    md = host.Model.from_gsm_npz(npz)               # each job owns its model (set-rotation-constant mutates it)
    md.set_rotation_constant(rot)
    models.append(md)
job_ids = list(range(len(jobs)))


def one(k):
    (ell, m), rot = jobs[k]
    t0 = time.perf_counter()
    model = models[k]
    t1 = time.perf_counter()
    s = host.ErasedSolver(model, ell, m, args.scheme, grid)
    t2 = time.perf_counter()
    res, _ = s.scan_and_optimize(freqs, host.Precision.Relative(1e-8))
    t3 = time.perf_counter()
    n = len(res)
    if n:
        pp = host.PostProcessor(model.eval(grid))
        t4 = time.perf_counter()
        pp.modes(s, np.array([r["root"] for r in res]))
        t5 = time.perf_counter()
    else:
        t4 = t5 = t3
    if args.keep:
        keep.append((s, pp if n else None))
    return n, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)


one(0)
for th in args.threads:
    t0 = time.perf_counter()
    if th == 1:
        out = [one(j) for j in job_ids]
    else:
        with ThreadPoolExecutor(max_workers=th) as pool:
            out = list(pool.map(one, job_ids))
    wall = time.perf_counter() - t0
    modes = sum(o[0] for o in out)
    ph = np.array([o[1] for o in out]).mean(axis=0) * 1e3
    print(f"threads {th:3d}: {wall * 1e3:8.1f} ms for {len(jobs)} scans x {args.steps} frequencies = {wall / len(jobs) * 1e3:6.2f} ms/scan, "
          f"{modes} modes, {modes / wall:7.0f} modes/s | mean per-scan phases (ms): model {ph[0]:.2f} solver {ph[1]:.2f} "
          f"scan {ph[2]:.2f} postproc-create {ph[3]:.2f} modes {ph[4]:.2f}", flush=True)
