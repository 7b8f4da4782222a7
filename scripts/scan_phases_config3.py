"""Phase times of ONE scan of the bench headline run alone (config 3: 10^4-point n=3 polytrope, ell=2, Colloc4, 10^5 frequencies):
det pass, the refinement part of scan_and_optimize, eigenvectors -- for both arithmetic builds, with the bracket state machine on
the host (round 1) or on the device, and the segment path on or off."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from storm_amd import host
model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
grid = model.grid(1)
freqs = host.linspace(1.0, 25.0, 100000)
prec = host.Precision.Relative(1e-8)
for arith, dev, seg in ((0, 0, 0), (0, 1, 0), (1, 0, -1), (1, 1, -1), (1, 0, 0), (1, 1, 0)):
    s = host.ErasedSolver(model, 2, 0, "colloc4", grid)
    s.set_option("arithmetic", arith); s.set_option("device_brackets", dev); s.set_option("segments", seg)
    s.det(freqs); res, _ = s.scan_and_optimize(freqs, prec); roots = np.array([r["root"] for r in res]); s.eigenvector(roots)
    best = None
    for rep in range(5):
        s.stats(reset=True)
        t0 = time.perf_counter(); s.det(freqs); t1 = time.perf_counter()
        res, _ = s.scan_and_optimize(freqs, prec); t2 = time.perf_counter()
        s.eigenvector(roots); t3 = time.perf_counter()
        st = s.stats()
        row = (1e3 * (t2 - t1 - (t1 - t0)), 1e3 * (t1 - t0), 1e3 * (t3 - t2), st["launches"], st["refine_rounds"], st["refine_evals"])
        best = row if best is None or row[0] < best[0] else best
    print(f"arith={arith} device_brackets={dev} segments={seg}: det {best[1]:.2f} ms | refinement part {best[0]:.3f} ms | "
          f"eigenvectors({len(roots)}) {best[2]:.2f} ms | launches {best[3]} rounds {best[4]} refine evals {best[5]}")
