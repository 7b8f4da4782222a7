import sys, time
sys.path.insert(0, '.')
import numpy as np
from oracle import storm_ref as R
import storm_amd as S
m = R.Model.from_gsm_npz('tests/golden/gsm_zams.npz')
grid = m.field('r_coord')
freqs = R.linspace(1., 25., 25)
ok = True
for ell in (0, 2):
  for sch in R.SCHEMES:
    o = R.Solver(m, ell, 0, sch, grid)
    h = S.HipSolver(sch, ell, 0, o.system(), o.n_int, o.p)
    ref = o.det(freqs, threads=8)
    for path in (1, 2):
        h.set_option("path", path)
        t = time.time(); got = h.det(freqs); dt = time.time() - t
        same = np.array_equal(ref.view(np.uint64), got.view(np.uint64))
        ok &= same
        print(ell, sch, 'path', path, 'bit-exact' if same else 'MISMATCH', f'{dt*1e3:.1f} ms',
              '' if same else (ref[:3], got[:3]))
    h.set_option("path", 0)
    rres, rd, _ = o.scan_and_optimize(freqs, 1, threads=8)
    gres, gd, _ = h.scan_and_optimize(freqs, 1)
    same = [r['root'] for r in rres] == [g['root'] for g in gres] and [r['evals'] for r in rres] == [g['evals'] for g in gres]
    ok &= same
    print('   scan roots', len(gres), 'identical' if same else 'DIFFER', h.stats())
    roots = np.array([g['root'] for g in gres])
    rdet, rvec = o.eigenvector(roots, threads=8)
    gdet, gvec = h.eigenvector(roots)
    same = np.array_equal(rvec.view(np.uint64), gvec.view(np.uint64)) and np.array_equal(rdet.view(np.uint64), gdet.view(np.uint64))
    ok &= same
    print('   eigenvectors', 'bit-exact' if same else 'MISMATCH', np.abs(rvec-gvec).max())
print('ALL OK' if ok else 'FAILURES')
