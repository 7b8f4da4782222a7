"""The dominant launch of bench.py's headline (BASELINE config 3: 10^4-point n=3 polytrope, ell=2, Colloc4, 10^5 frequencies in
ONE det_hex launch) a few times, for rocprofv3 --kernel-trace / --pmc passes.  usage: profile_config3.py [arithmetic 0|1] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (first: see tests/conftest.py)
from storm_amd import host
arith = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
s = host.ErasedSolver(model, 2, 0, "colloc4", model.grid(1))
s.set_option("arithmetic", arith)
s.set_option("hex_rounds", 0)
f = host.linspace(1.0, 25.0, 100000)
for _ in range(reps):
    d = s.det(f)
print("ok", d[:2])
