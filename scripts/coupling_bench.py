"""Throughput of the coupling-integral kernel (storm_coupling_matrices): M modes of tams (replicated from one scan),
kernel time from HIP events inside the library.  usage: python scripts/coupling_bench.py [M ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import host, ffi  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [32, 128, 512]
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
# parity build: vector walk with 8 x 8 and 16 x 16 tiles; FMA build: the same two, then the matrix-pipe walk (tile = 0 here)
# (tile = -16 / -32 here: the matrix-pipe walk with that tile)
for tile, arith, M in [(t, a, m) for a in (0, 1) for t in ((8, 16) if a == 0 else (16, -16, -32)) for m in sizes]:
    raw.set_option("arithmetic", arith)
    raw.set_option("coupling_matrix", 1 if tile < 0 else 0)
    raw.set_option("coupling_matrix_tile", -tile if tile < 0 else 0)
    raw.set_option("coupling_tile", max(tile, 0))
    sel = np.arange(M) % freqs.size
    f, e, fl = freqs[sel], ells[sel], np.ascontiguousarray(fields[sel])
    raw.coupling_matrices(e, f, fl, 0, metric)
    t0 = time.perf_counter()
    raw.coupling_matrices(e, f, fl, 0, metric)
    wall = time.perf_counter() - t0
    ms = raw.last_kernel_ms()
    flops = M * M * (grid.size - 1) * 148.0
    print(f"tile={tile:2d} arith={arith} M={M:5d}  kernel {ms:9.3f} ms  {flops / ms / 1e9:8.2f} TFLOP/s (algorithmic, un-fused)   call {wall * 1e3:9.1f} ms", flush=True)
