import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import storm_ref as o
import storm_amd
o.build()
m = o.Model.from_gsm_npz('tests/golden/gsm_tams.npz')
s = o.Solver(m, 1, 0, "colloc4", m.field("r_coord"))
f = o.linspace(1.0, 25.0, 2000)
res, _, _ = s.scan_and_optimize(f, o.relative_to_ulp(1e-8), threads=16)
rr = np.array([r["root"] for r in res])
_, rvec = s.eigenvector(rr, threads=16)
h = storm_amd.HipSolver(s.scheme, s.ell, s.m, s.system(), s.n_int, s.p)
h.set_option("arithmetic", 1)
for seg in (16, 48, 256, 1024):
    h.set_option("segments", seg)
    _, g = h.eigenvector(rr)
    d = np.abs(g - rvec).reshape(len(rr), -1, 4).max(axis=2)
    mx = np.max(np.abs(rvec), axis=1)
    err = d.max(axis=1) / mx
    print("seg", seg, "err", " ".join("%.1e" % e for e in err[:10]))
    k = 0
    where = np.argmax(d[k])
    print("  mode0: worst point", where, "of", d.shape[1], "err at segment nodes", " ".join("%.1e" % (d[k][p] / mx[k]) for p in range(0, d.shape[1], ((seg + 15) // 16 * 16) * max(1, d.shape[1] // ((seg + 15) // 16 * 16) // 12))))
    print("  mode0 |y| profile", " ".join("%.1e" % v for v in np.abs(rvec[0].reshape(-1, 4)).max(axis=1)[::500]))
print("---- error profile, mode 0, seg 48")
h.set_option("segments", 48)
_, g = h.eigenvector(rr)
d = np.abs(g - rvec).reshape(len(rr), -1, 4)
y = np.abs(rvec).reshape(len(rr), -1, 4)
for p in list(range(0, 100, 4)) + list(range(100, 6200, 300)):
    print(p, "err", " ".join("%.1e" % v for v in d[0][p]), "|y|", " ".join("%.1e" % v for v in y[0][p]))
print("|L_i| max entries, omega = root0:", " ".join("%.1e" % np.max(np.abs(s.step_matrix(i, rr[0]))) for i in list(range(0, 60, 3)) + [100, 1000, 3000, 6000, 6198]))
