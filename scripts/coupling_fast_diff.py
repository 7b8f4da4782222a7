"""Largest difference between the FMA build and the parity build of the coupling integrals, relative to the largest
magnitude of each matrix (tams, modes of ell = 0, 2, 4; the tolerance of tests/test_hip_coupling.py is 1e-11)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import host, ffi  # noqa: E402

model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
model.set_rotation_constant(0.1)
grid = model.field("r_coord")
on_grid = model.eval(grid)
pp = host.PostProcessor(on_grid)
fields, ells, freqs = [], [], []
for ell in (0, 2, 4):
    s = host.ErasedSolver(model, ell, 0, "colloc4", grid)
    res, _ = s.scan_and_optimize(host.linspace(1.0, 25.0, 2000), host.Precision.Relative(1e-8))
    roots = np.array([r["root"] for r in res])
    got = pp.modes(s, roots, want_fields=True)
    fields.append(got["fields"]); ells += [ell] * roots.size; freqs += list(roots)
fields, ells, freqs = np.ascontiguousarray(np.concatenate(fields)), np.array(ells, dtype=np.uint64), np.array(freqs)
x = grid
metric = {"alpha": 1e-3 * x ** 2, "dalpha": 2e-3 * x ** 2, "ddalpha": 4e-3 * x ** 2,
          "beta": -1.2e-3 * x ** 2, "dbeta": -2.4e-3 * x ** 2, "ddbeta": -4.8e-3 * x ** 2}
raw = ffi.HipPostproc({k: on_grid.field(k) for k in ffi.MODEL_FIELDS})
ref = raw.coupling_matrices(ells, freqs, fields, 0, metric)
raw.set_option("arithmetic", 1)
for tile in (8, 16):
    raw.set_option("coupling_tile", tile)
    got = raw.coupling_matrices(ells, freqs, fields, 0, metric)
    print(tile, {k: float(np.max(np.abs(got[k] - ref[k])) / (np.max(np.abs(ref[k])) or 1.0)) for k in ref}, flush=True)
