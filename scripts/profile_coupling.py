"""Runs storm_coupling_matrices a few times on M modes (default 512) for rocprofv3 passes.
usage: profile_coupling.py [M] [reps] [arithmetic] [sorted: 1 = modes grouped by degree, as scans deliver them] [tile of the
matrix-pipe walk: 0 (automatic), 16, 32]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import host, ffi
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
arith = int(sys.argv[3]) if len(sys.argv) > 3 else 0
model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
model.set_rotation_constant(0.1)
grid = model.field("r_coord")
on_grid = model.eval(grid)
pp = host.PostProcessor(on_grid)
fields, ells, freqs = [], [], []
for ell in (0, 2):
    s = host.ErasedSolver(model, ell, 0, "colloc4", grid)
    res, _ = s.scan_and_optimize(host.linspace(1.0, 25.0, 2000), host.Precision.Relative(1e-8))
    roots = np.array([r["root"] for r in res])
    got = pp.modes(s, roots, want_fields=True)
    fields.append(got["fields"]); ells += [ell] * roots.size; freqs += list(roots)
fields, ells, freqs = np.concatenate(fields), np.array(ells, dtype=np.uint64), np.array(freqs)
sel = np.arange(M) % freqs.size
if len(sys.argv) > 4 and int(sys.argv[4]):
    sel = sel[np.argsort(ells[sel], kind="stable")]
x = grid
metric = {"alpha": 1e-3 * x ** 2, "dalpha": 2e-3 * x ** 2, "ddalpha": 4e-3 * x ** 2,
          "beta": -1.2e-3 * x ** 2, "dbeta": -2.4e-3 * x ** 2, "ddbeta": -4.8e-3 * x ** 2}
raw = ffi.HipPostproc({k: on_grid.field(k) for k in ffi.MODEL_FIELDS})
raw.set_option("arithmetic", arith)
if len(sys.argv) > 5:
    raw.set_option("coupling_matrix_tile", int(sys.argv[5]))
f, e, fl = freqs[sel], ells[sel], np.ascontiguousarray(fields[sel])
for _ in range(reps):
    out = raw.coupling_matrices(e, f, fl, 0, metric)
import hashlib
digest = hashlib.md5(b"".join(np.ascontiguousarray(out[k]).tobytes() for k in ffi.COUPLING_MATRICES)).hexdigest()
print("ok", M, raw.last_kernel_ms(), "ms", float(out["d_squared"][0, 0]), digest)
