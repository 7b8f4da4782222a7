"""Runs storm_eigenvector_batch on the modes of one tams scan a few times, for rocprofv3 passes.
usage: profile_eigen.py [scheme] [reps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import host
scheme = sys.argv[1] if len(sys.argv) > 1 else "colloc4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
grid = model.field("r_coord")
s = host.ErasedSolver(model, 2, 0, scheme, grid)
res, _ = s.scan_and_optimize(host.linspace(1.0, 25.0, 2000), host.Precision.Relative(1e-8))
roots = np.array([r["root"] for r in res])
for _ in range(reps):
    v = s.eigenvector(roots)
print("ok", roots.size, "modes", s.n_int, "intervals", float(v[0, -1]))
