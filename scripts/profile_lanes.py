"""One det batch on the headline system with a forced lane count, for rocprofv3 --pmc.  usage: profile_lanes.py lanes arith n"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from storm_amd import host
lanes, arith, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
s = host.ErasedSolver(model, 2, 0, "colloc4", model.grid(1))
s.set_option("arithmetic", arith); s.set_option("hex_rounds", 0); s.set_option("path", 4); s.set_option("hex_lanes", lanes)
f = host.linspace(1.0, 25.0, n)
for _ in range(2):
    d = s.det(f)
print("ok")
