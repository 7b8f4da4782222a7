"""Wall time of the phases of one scan of the bench alone on the GPU (tams, ell=3, 10^4 frequencies, precision 1e-8):
the scan (discriminant launch + refinement rounds) and the eigenvectors.  usage: scan_phases.py scheme"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import storm_amd
from storm_amd import host
scheme = sys.argv[1] if len(sys.argv) > 1 else "colloc8"
model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
s = host.ErasedSolver(model, 3, 0, scheme, model.field("r_coord"))
s.set_option("profile", 1)
f = host.linspace(1.0, 25.0, 10000)
prec = host.Precision.Relative(1e-8)
for rep in range(4):
    s.kernel_times(reset=True)
    t0 = time.perf_counter()
    d = s.det(f)
    t1 = time.perf_counter()
    res, _ = s.scan_and_optimize(f, prec)
    t2 = time.perf_counter()
    roots = np.array([r["root"] for r in res])
    v = s.eigenvector(roots)
    t3 = time.perf_counter()
    kt = {k: (round(ms, 3), cnt) for k, (ms, cnt) in s.kernel_times(reset=True).items() if cnt}
    print(f"{scheme} rep {rep}: det {1e3*(t1-t0):.2f} ms | scan_and_optimize {1e3*(t2-t1):.2f} ms | eigenvector({roots.size}) {1e3*(t3-t2):.2f} ms | kernels {kt}")
