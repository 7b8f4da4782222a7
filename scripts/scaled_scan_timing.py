"""Raw scan against scaled scan (option scaled_scan) on the bench headline grid, both builds: wall time per scan, bracket
index lists, largest root difference."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import storm_amd
from storm_amd import host, ffi
model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
grid = model.grid(1)
es = host.ErasedSolver(model, 2, 0, "colloc4", grid)
h = ffi.HipSolver.borrow(es.raw_handle(), owner=es)
f = host.linspace(1.0, 25.0, 100000)
ulp = storm_amd.relative_to_ulp(1e-8)
for arith in (1, 0):
    h.set_option("arithmetic", arith)
    h.set_option("scaled_scan", 0)
    raw, _, _ = h.scan_and_optimize(f, ulp, want_dets=False)
    t0 = time.perf_counter(); raw, _, _ = h.scan_and_optimize(f, ulp, want_dets=False); t_raw = time.perf_counter() - t0
    h.set_option("scaled_scan", 1)
    sc, _, _ = h.scan_and_optimize(f, ulp, want_dets=False)
    t0 = time.perf_counter(); sc, _, _ = h.scan_and_optimize(f, ulp, want_dets=False); t_sc = time.perf_counter() - t0
    same = [b["index"] for b in raw] == [b["index"] for b in sc]
    dr = max(abs(a["root"] - b["root"]) / a["root"] for a, b in zip(raw, sc)) if same and raw else None
    print(f"arith {arith}: raw scan {t_raw*1e3:.1f} ms, scaled scan {t_sc*1e3:.1f} ms, {len(sc)} brackets, same indices {same}, max root diff {dr}")
