import sys, time; sys.path.insert(0,'.')
import numpy as np
from oracle import storm_ref as R
import storm_amd
m = R.Model.polytrope(3.0, 5.0/3.0, 6.89684861937/9999.0)
g = m.grid(1)
o = R.Solver(m, 2, 0, "colloc4", g)
f = R.linspace(1., 25., 100000)
def mk(): return storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
h = mk(); d = h.det(f); print("A first call det 1e5:", np.isfinite(d).sum(), d[:2])
d = h.det(f); print("A second call:", np.isfinite(d).sum())
h = mk(); time.sleep(0.5); d = h.det(f); print("B sleep then det:", np.isfinite(d).sum())
h = mk(); d = h.det(f[:1000]); d = h.det(f); print("C small first:", np.isfinite(d).sum())
h = mk(); h.set_option("path", 4); d = h.det(f); print("D hex first:", np.isfinite(d).sum())
h = mk(); h.set_option("path", 1); d = h.det(f[:65536]); print("E fused 65536 first:", np.isfinite(d).sum())
h = mk(); h.set_option("path", 1); d = h.det(f[:4096]); print("F fused 4096 first:", np.isfinite(d).sum())
