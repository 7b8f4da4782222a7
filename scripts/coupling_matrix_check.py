"""The matrix-pipe walk of the coupling integrals (FMA build, 16 x 16 tiles of like degrees) against the vector walk (8 x 8
tiles) on modes sorted by degree: largest difference relative to each matrix's largest magnitude, and the kernel times.
usage: python scripts/coupling_matrix_check.py [M ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import host, ffi  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [256, 1024]
model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
model.set_rotation_constant(0.1)
grid = model.field("r_coord")
on_grid = model.eval(grid)
pp = host.PostProcessor(on_grid)
fields, ells, freqs = [], [], []
for ell in (0, 1, 2, 3, 4):
    s = host.ErasedSolver(model, ell, 0, "colloc4", grid)
    res, _ = s.scan_and_optimize(host.linspace(1.0, 25.0, 2000), host.Precision.Relative(1e-8))
    roots = np.array([r["root"] for r in res])
    got = pp.modes(s, roots, want_fields=True)
    fields.append(got["fields"]); ells += [ell] * roots.size; freqs += list(roots)
fields, ells, freqs = np.concatenate(fields), np.array(ells, dtype=np.uint64), np.array(freqs)
x = grid
metric = {"alpha": 1e-3 * x ** 2, "dalpha": 2e-3 * x ** 2, "ddalpha": 4e-3 * x ** 2,
          "beta": -1.2e-3 * x ** 2, "dbeta": -2.4e-3 * x ** 2, "ddbeta": -4.8e-3 * x ** 2}
raw = ffi.HipPostproc({k: on_grid.field(k) for k in ffi.MODEL_FIELDS})
raw.set_option("arithmetic", 1)
for M in sizes:
    pool = np.arange(freqs.size)
    if os.environ.get("ONLY_ELL"):  # e.g. 2 (one degree: every full tile takes the matrix-pipe walk) or 0,4
        pool = pool[np.isin(ells, [int(v) for v in os.environ["ONLY_ELL"].split(",")])]
    sel = pool[np.arange(M) % pool.size]
    sel = sel[np.argsort(ells[sel], kind="stable")]
    f, e, fl = freqs[sel], ells[sel], np.ascontiguousarray(fields[sel])
    out = {}
    for tile in (8, 16):
        raw.set_option("coupling_tile", tile)
        raw.coupling_matrices(e, f, fl, 0, metric)
        out[tile] = raw.coupling_matrices(e, f, fl, 0, metric)
        print(f"M={M} tile={tile}: kernel {raw.last_kernel_ms():.3f} ms", flush=True)
    a, b = out[8], out[16]
    names = sorted(a.keys()) if isinstance(a, dict) else range(len(a))
    for k in names:
        u, v = np.asarray(a[k]), np.asarray(b[k])
        scale = np.max(np.abs(u))
        print(f"  {k}: max |diff| / max |value| = {np.max(np.abs(u - v)) / scale:.3e}  (max |value| {scale:.3e})", flush=True)
