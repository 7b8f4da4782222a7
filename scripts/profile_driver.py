"""Runs only the dominant launch of bench.py (colloc8, ell=3, tams, 10^4 frequencies) a few times,
for rocprofv3 --pmc passes (HBM traffic) and kernel traces.  usage: profile_driver.py [path] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storm_amd import host
path = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
scheme = sys.argv[3] if len(sys.argv) > 3 else "colloc8"
model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
s = host.ErasedSolver(model, 3, 0, scheme, model.field("r_coord"))
s.set_option("path", path)
s.set_option("hex_rounds", 0)  # the whole batch in one launch, like the roofline leg of bench.py
f = host.linspace(1.0, 25.0, 10000)
for _ in range(reps):
    d = s.det(f)
print("ok", d[:2])
