"""Per-kernel timing of the det paths on one GPU (HIP events inside the library).
usage: python scripts/kernel_bench.py [schemes...] [--n N] [--fused-n N]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import host  # noqa: E402

FLOPS = {"colloc2": 610.1, "colloc4": 1396.1, "colloc6": 2694.1, "colloc8": 5048.1,
         "magnus2": 1486.0, "magnus4": 1848.0, "magnus6": 2562.0, "magnus8": 3900.1}

ap = argparse.ArgumentParser()
ap.add_argument("schemes", nargs="*", default=list(FLOPS))
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--fused-n", type=int, default=65536)
ap.add_argument("--ell", type=int, default=2)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--arith", type=int, default=0)
ap.add_argument("--lanes", type=int, default=0)
args = ap.parse_args()

model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
grid = model.field("r_coord")
for scheme in args.schemes:
    s = host.ErasedSolver(model, args.ell, 0, scheme, grid)
    s.set_option("profile", 1)
    s.set_option("arithmetic", args.arith)
    s.set_option("hex_lanes", args.lanes)
    n_int = s.n_int
    out = [f"{scheme:8s}"]
    for path, n in ((4, args.n), (4, 10000), (4, 4 * args.n)):
        f = host.linspace(1.0, 25.0, n)
        s.set_option("path", path)
        s.det(f)
        s.kernel_times(reset=True)
        for _ in range(args.reps):
            s.det(f)
        kt = s.kernel_times(reset=True)
        for kind, (ms, cnt) in kt.items():
            if cnt:
                per = ms / args.reps
                tf = n * n_int * FLOPS[scheme] / (per * 1e-3) / 1e12 if kind in ("build_steps", "det_fused", "det_quad", "det_hex") else 0
                gbs = n * n_int * 128 / (per * 1e-3) / 1e9 if kind in ("build_steps", "chain") else 0
                out.append(f"{kind}[n={n}] {per:8.3f} ms" + (f" {tf:5.1f} TF" if tf else "") + (f" {gbs:6.0f} GB/s" if gbs else ""))
    print(" | ".join(out), flush=True)
