"""Wall time of the bracket refinement alone (storm_optimize_brackets on the 22 brackets of the bench headline's scan: config 3,
10^4-point n=3 polytrope, ell=2, Colloc4, precision 1e-8), state machine on the host (round 1) vs on the device, per arithmetic
build and with the segment path on/off; and of the eigenvectors of the 22 modes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from storm_amd import host, ffi
model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
grid = model.grid(1)
freqs = host.linspace(1.0, 25.0, 100000)
ulp = ffi.relative_to_ulp(1e-8)
es = host.ErasedSolver(model, 2, 0, "colloc4", grid)
h = ffi.HipSolver.borrow(es.raw_handle(), owner=es)
found, _, _ = h.scan_brackets(freqs)
for arith, seg in ((0, 0), (1, -1), (1, 0)):
    for dev in (0, 1):
        h.set_option("arithmetic", arith); h.set_option("segments", seg); h.set_option("device_brackets", dev)
        res = h.optimize_brackets(found, ulp)
        roots = np.array([b["root"] for b in res])
        h.eigenvector(roots)
        ts, te = [], []
        for _ in range(20):
            h.stats(reset=True)
            t0 = time.perf_counter(); h.optimize_brackets(found, ulp); t1 = time.perf_counter()
            st = h.stats()
            h.eigenvector(roots); t2 = time.perf_counter()
            ts.append(t1 - t0); te.append(t2 - t1)
        print(f"arith={arith} segments={seg} device_brackets={dev}: refinement of {len(found)} brackets {1e3*min(ts):.3f} ms (median {1e3*np.median(ts):.3f}), "
              f"launches {st['launches']}, rounds {st['refine_rounds']}, evals {st['refine_evals']} | eigenvectors {1e3*min(te):.3f} ms")
