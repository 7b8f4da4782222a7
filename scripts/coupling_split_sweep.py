"""The matrix-pipe walk of storm_coupling_matrices (FMA build) against its two knobs: tile (16 / 32) and the number of pieces
the grid is cut into per tile (option "coupling_split"; 0 = the library's plan).  Kernel time from HIP events inside the
library (walk + the launch that adds the pieces).  usage: python scripts/coupling_split_sweep.py [M ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import host, ffi  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [80, 240, 512, 1024, 2048]
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
fields, ells, freqs = np.concatenate(fields), np.array(ells, dtype=np.uint64), np.array(freqs)
print(f"{freqs.size} distinct modes of ell = 0, 2, 4 on {grid.size} points", flush=True)
x = grid
metric = {"alpha": 1e-3 * x ** 2, "dalpha": 2e-3 * x ** 2, "ddalpha": 4e-3 * x ** 2,
          "beta": -1.2e-3 * x ** 2, "dbeta": -2.4e-3 * x ** 2, "ddbeta": -4.8e-3 * x ** 2}
raw = ffi.HipPostproc({k: on_grid.field(k) for k in ffi.MODEL_FIELDS})
raw.set_option("arithmetic", 1)
raw.set_option("coupling_matrix", 1)
for M in sizes:
    sel = np.arange(M) % freqs.size
    f, e, fl = freqs[sel], ells[sel], np.ascontiguousarray(fields[sel])
    ref = None
    for tile, split in [(0, 0)] + [(t, s) for t in (16, 32) for s in (1, 2, 3, 4, 6, 8, 16, 32)]:
        raw.set_option("coupling_matrix_tile", tile)
        raw.set_option("coupling_split", split)
        out = raw.coupling_matrices(e, f, fl, 0, metric)
        best = 1e9
        for _ in range(3):
            raw.coupling_matrices(e, f, fl, 0, metric)
            best = min(best, raw.last_kernel_ms())
        mats = np.stack([np.asarray(out[k]) for k in sorted(out)]) if isinstance(out, dict) else np.asarray(out)
        if ref is None:
            ref = mats
        scale = np.abs(ref).reshape(ref.shape[0], -1).max(axis=1).reshape(-1, *([1] * (ref.ndim - 1)))
        diff = float(np.max(np.abs(mats - ref) / np.where(scale > 0, scale, 1.0)))
        flops = M * M * (grid.size - 1) * 148.0
        print(f"M={M:5d} tile={tile:2d} split={split:2d}  kernel {best:8.3f} ms  {flops / best / 1e9:8.2f} TFLOP/s ref-equivalent  "
              f"max diff to the plan's result {diff:.1e}", flush=True)
