"""One rank's step of a strong-scaled headline scan at G = 8 (slice of 12 501 frequencies, 3 of 22 brackets, their eigenvectors)
a few times, for a rocprofv3 --kernel-trace --memory-copy-trace timeline; prints the host-side wall times of the three calls.
usage: slice_timeline.py [G] [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import storm_amd
from storm_amd import host, ffi, distributed as D
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
es = host.ErasedSolver(model, 2, 0, "colloc4", model.grid(1))
h = ffi.HipSolver.borrow(es.raw_handle(), owner=es)
h.set_option("arithmetic", 1)
freqs = torch.from_numpy(host.linspace(1.0, 25.0, 100000)).pin_memory().numpy()
ulp = storm_amd.relative_to_ulp(1e-8)
allb, _, _ = h.scan_brackets(freqs)
buf = torch.empty(64 * (es.n_int + 1) * 4, dtype=torch.float64, pin_memory=True).numpy()
rank = G - 1 if G > 1 else 0
lo, mine = D.local_slice(freqs, G, 0)
share = [allb[j] for j in D.deal(len(allb), G, 0)]
mark = torch.zeros(64, device="cuda")
for rep in range(reps):
    mark.add_(1.0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.scan_brackets(mine, want_dets=False)
    t1 = time.perf_counter()
    my = h.optimize_brackets(share, ulp) if share else []
    t2 = time.perf_counter()
    if my:
        h.eigenvector(np.array([b["root"] for b in my]), out=buf)
    t3 = time.perf_counter()
    print(f"rep {rep}: scan {1e3*(t1-t0):.3f} ms, refine {1e3*(t2-t1):.3f} ms ({len(share)} brackets), eigenvectors {1e3*(t3-t2):.3f} ms", flush=True)
mark.add_(1.0); torch.cuda.synchronize()
