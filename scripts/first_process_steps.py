"""Per-step wall time of the headline scan from the start of a process (is the first process on a fresh box slower?)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import storm_amd
from storm_amd import host, ffi
model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
es = host.ErasedSolver(model, 2, 0, "colloc4", model.grid(1))
es.set_option("profile", 1)
h = ffi.HipSolver.borrow(es.raw_handle(), owner=es)
h.set_option("arithmetic", 1)
freqs = torch.from_numpy(host.linspace(1.0, 25.0, 100000)).pin_memory().numpy()
ulp = storm_amd.relative_to_ulp(1e-8)
buf = torch.empty(64 * (es.n_int + 1) * 4, dtype=torch.float64, pin_memory=True).numpy()
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, _, _ = h.scan_and_optimize(freqs, ulp, cap=8192, want_dets=False)
    t1 = time.perf_counter()
    h.eigenvector(np.array([b["root"] for b in res]), out=buf)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1))
kt = h.kernel_times(reset=True)
print("steps scan/eig ms:", " ".join(f"{a*1e3:.1f}/{b*1e3:.2f}" for a, b in ts))
print("kernel avg ms", {k: round(v[0] / v[1], 3) for k, v in kt.items() if v[1]})
