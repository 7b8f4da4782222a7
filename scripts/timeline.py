"""Phase timeline of the concurrent 6-scan bench step (diagnostic)."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import storm_amd
from storm_amd import host
from concurrent.futures import ThreadPoolExecutor
model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
grid = model.field("r_coord")
CASES = [(ell, scheme) for scheme in ("colloc8", "magnus6") for ell in (3, 2, 1)]
ARITH = int(sys.argv[1]) if len(sys.argv) > 1 else 0
solvers = [host.ErasedSolver(model, ell, 0, sch, grid) for ell, sch in CASES]
[s.set_option("arithmetic", ARITH) for s in solvers]
freqs = host.linspace(1.0, 25.0, 10000)
prec = host.Precision.Relative(1e-8)
T0 = [0.0]
def job(k):
    t = [time.perf_counter()]
    res, inv = solvers[k].scan_and_optimize(freqs, prec); t.append(time.perf_counter())
    v = solvers[k].eigenvector(np.array([r["root"] for r in res])); t.append(time.perf_counter())
    return [round((x - T0[0]) * 1e3, 1) for x in t]
pool = ThreadPoolExecutor(6)
for rep in range(3):
    T0[0] = time.perf_counter()
    out = list(pool.map(job, range(6)))
    total = (time.perf_counter() - T0[0]) * 1e3
    print(f"rep {rep}: total {total:.1f} ms  [start, scan+refine done, eig done]")
    for c, o in zip(CASES, out):
        print("   ", c, o)
