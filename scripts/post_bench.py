"""Timing of the mode pipeline on one GPU: eigenvectors to the host vs eigenvectors + post-processing on the
device (storm_modes_batch).  usage: python scripts/post_bench.py [--scheme colloc4] [--ell 2] [--reps 5]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scheme", default="colloc4")
ap.add_argument("--ell", type=int, default=2)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--copies", type=int, default=1, help="repeat the root list this many times (bigger batches)")
args = ap.parse_args()

model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
grid = model.field("r_coord")
s = host.ErasedSolver(model, args.ell, 0, args.scheme, grid)
res, _ = s.scan_and_optimize(host.linspace(1.0, 25.0, 10000), host.Precision.Relative(1e-8))
roots = np.tile(np.array([r["root"] for r in res]), args.copies)
pp = host.PostProcessor(model.eval(grid))
npts = grid.size
print(f"{roots.size} modes, {npts} points, scheme {args.scheme}, ell {args.ell}", flush=True)


def timed(fn):
    fn()
    t = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return 1e3 * min(t)


t_eig = timed(lambda: s.eigenvector(roots))
t_ids = timed(lambda: pp.modes(s, roots, want_fields=False))
t_all = timed(lambda: pp.modes(s, roots, want_fields=True))
vecs = s.eigenvector(roots)
t_split = timed(lambda: pp.post_process(args.ell, 0, roots, vecs))
print(f"eigenvectors to host            {t_eig:8.3f} ms")
print(f"modes (ids + nodes only)        {t_ids:8.3f} ms")
print(f"modes (13 fields to host)       {t_all:8.3f} ms")
print(f"post-process of host vectors    {t_split:8.3f} ms")
s.set_option("profile", 1)
for want in (False, True):
    s.kernel_times(reset=True)
    for _ in range(args.reps):
        pp.modes(s, roots, want_fields=want)
    kt = s.kernel_times(reset=True)
    ms = kt["postprocess"][0] / args.reps
    bytes_alg = roots.size * npts * ((40 + 40) + (136 if want else 0))
    print(f"postprocess kernels (fields={want}): {ms:7.3f} ms per batch, algorithmic {bytes_alg / 1e6:.1f} MB "
          f"-> {bytes_alg / ms / 1e6:.1f} GB/s; eigen {kt['eigen'][0] / args.reps:.3f} ms, "
          f"build_steps {kt['build_steps'][0] / args.reps:.3f} ms")
