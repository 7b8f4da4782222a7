"""Automatic kernel choice (hex_lanes = 0: det_mix where mix_col_blocks takes the batch) against det_hex with 16 lanes, per
scheme and build, on tams.  usage: mix_auto_check.py scheme[,scheme...] [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from storm_amd import host
model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
ns = [int(a) for a in sys.argv[2:]] or [20000, 25000, 29000, 40000, 50000, 58000, 70000, 86000]
for scheme in sys.argv[1].split(","):
    s = host.ErasedSolver(model, 3, 0, scheme, model.field("r_coord"))
    s.set_option("profile", 1); s.set_option("hex_rounds", 0)
    for arith in (0, 1):
        s.set_option("arithmetic", arith)
        row = []
        for n in ns:
            f = host.linspace(1.0, 25.0, n)
            t = []
            for lanes in (0, 16):
                s.set_option("hex_lanes", lanes)
                s.det(f); s.kernel_times(reset=True)
                for _ in range(3):
                    s.det(f)
                t.append(sum(v[0] for v in s.kernel_times(reset=True).values()) / 3)
            row.append(f"{n}: {t[0]:.2f}/{t[1]:.2f}")
        print(f"{scheme} arith={arith} auto/hex16 ms  " + "  ".join(row), flush=True)
