import os, sys, time
import numpy as np
sys.path.insert(0, '.')
from storm_amd import host
model = host.Model.from_gsm_npz("tests/golden/gsm_tams.npz")
s = host.ErasedSolver(model, 3, 0, "magnus6", model.field("r_coord"))
s.set_option("profile", 1)
res, _ = s.scan_and_optimize(host.linspace(1.0, 25.0, 10000), host.Precision.Relative(1e-8))
roots = np.array([r["root"] for r in res])
for n in (1, 8, 31):
    s.eigenvector(roots[:n]); s.kernel_times(reset=True)
    t0 = time.perf_counter(); s.eigenvector(roots[:n]); t1 = time.perf_counter()
    print(n, "modes: wall", round(1e3*(t1-t0), 2), "ms; kernels", {k: round(v[0], 2) for k, v in s.kernel_times(reset=True).items() if v[1]})
