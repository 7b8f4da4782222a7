"""det_hex (16 / 32 lanes per frequency) vs det_col (4 lanes) on the bench headline's system (config 3, Colloc4), n frequencies:
HIP-event kernel time per launch.  usage: lanes_compare.py [n] [scheme]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from storm_amd import host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
scheme = sys.argv[2] if len(sys.argv) > 2 else "colloc4"
model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
s = host.ErasedSolver(model, 2, 0, scheme, model.grid(1))
s.set_option("profile", 1); s.set_option("hex_rounds", 0)
f = host.linspace(1.0, 25.0, n)
for arith in (1, 0):
    s.set_option("arithmetic", arith)
    for lanes in (16, 4, 0, 32):
        s.set_option("hex_lanes", lanes)
        s.det(f); s.kernel_times(reset=True)
        for _ in range(3):
            s.det(f)
        ms, cnt = s.kernel_times(reset=True)["det_hex"]
        print(f"{scheme} n={n} arith={arith} lanes={lanes}: {ms / cnt:.3f} ms per batch")
