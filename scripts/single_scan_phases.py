"""Phase times of ONE scan run alone (diagnostic): det pass, refinement, eigenvectors."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import host
model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
grid = model.field("r_coord")
freqs = host.linspace(1.0, 25.0, 10000)
prec = host.Precision.Relative(1e-8)
for scheme in ("colloc8", "magnus6"):
    for arith in (0, 1):
        s = host.ErasedSolver(model, 3, 0, scheme, grid)
        s.set_option("arithmetic", arith)
        s.set_option("profile", 1)
        s.det(freqs); res, _ = s.scan_and_optimize(freqs, prec); roots = np.array([r["root"] for r in res]); s.eigenvector(roots)
        s.stats(reset=True); s.kernel_times(reset=True)
        t0 = time.perf_counter(); s.det(freqs); t1 = time.perf_counter()
        res, _ = s.scan_and_optimize(freqs, prec); t2 = time.perf_counter()
        v = s.eigenvector(roots); t3 = time.perf_counter()
        st = s.stats()
        print(f"{scheme} arith={arith}: det {1e3*(t1-t0):.2f} ms | scan+refine {1e3*(t2-t1):.2f} ms (refine part {1e3*(t2-t1-(t1-t0)):.2f}, rounds {st['refine_rounds']}, evals {st['refine_evals']}) | eigen {1e3*(t3-t2):.2f} ms ({len(roots)} modes)")
        print("     kernels:", {k: (round(ms, 2), c) for k, (ms, c) in s.kernel_times().items() if c})
