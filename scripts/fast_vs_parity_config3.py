"""Parity vs FMA build on BASELINE config 3 (10^4-point n=3 polytrope, ell=2, Colloc4, 10^5 frequencies): discriminant and root differences."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storm_amd import host, ffi
import storm_amd
model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
grid = model.grid(1)
f = host.linspace(1.0, 25.0, 100000)
ulp = storm_amd.relative_to_ulp(1e-8)
for scheme in sys.argv[1:] or ["colloc4"]:
    es = host.ErasedSolver(model, 2, 0, scheme, grid)
    h = ffi.HipSolver.borrow(es.raw_handle(), owner=es)
    out = {}
    for a in (0, 1):
        h.set_option("arithmetic", a)
        out[a] = h.scan_and_optimize(f, ulp, cap=8192)
    d0, d1 = out[0][1], out[1][1]
    a = np.abs(d0)
    scale = np.maximum.reduce([np.roll(a, 1), a, np.roll(a, -1)])
    err = np.abs(d1 - d0) / scale
    print(scheme, "det rel err (local scale): max %.3e median %.3e" % (err.max(), np.median(err)), "signs equal", np.array_equal(np.signbit(d0), np.signbit(d1)))
    r0 = np.array([b["root"] for b in out[0][0]]); r1 = np.array([b["root"] for b in out[1][0]])
    print(" brackets", len(r0), len(r1), "root rel diff max %.3e" % np.max(np.abs(r1 - r0) / np.abs(r0)))
    print(" ratio d1/d0 - 1 at a few points:", ((d1 - d0) / d0)[[10, 1000, 30000, 30001, 30002, 60000, 99990]])
    for ulp2, nm in ((1, "ULP(1)"),):
        q0 = []; 
        for a in (0, 1):
            h.set_option("arithmetic", a)
            q0.append(np.array([b["root"] for b in h.scan_and_optimize(f, ulp2, cap=8192)[0]]))
        print(" roots at", nm, "rel diff max %.3e" % np.max(np.abs(q0[1] - q0[0]) / np.abs(q0[0])))
