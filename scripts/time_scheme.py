"""det kernel time of one scheme on tams (ell=3, 10^4 frequencies, one launch), both builds.
usage: time_scheme.py scheme[,scheme..] [n] [hex_lanes: 0 = automatic (default), 4, 16, 32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from storm_amd import host
FL = {"colloc2": 610.1, "colloc4": 1396.1, "colloc6": 2694.1, "colloc8": 5048.1, "magnus2": 1486.0, "magnus4": 1848.0, "magnus6": 2562.0, "magnus8": 3900.1}
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
model = host.Model.from_gsm_npz(os.path.join(ROOT, "tests", "golden", "gsm_tams.npz"))
for scheme in sys.argv[1].split(","):
    s = host.ErasedSolver(model, 3, 0, scheme, model.field("r_coord"))
    s.set_option("profile", 1); s.set_option("hex_rounds", 0); s.set_option("hex_lanes", lanes)
    f = host.linspace(1.0, 25.0, n)
    for arith in (0, 1):
        s.set_option("arithmetic", arith)
        s.det(f); s.kernel_times(reset=True)
        for _ in range(3):
            s.det(f)
        ms, cnt = s.kernel_times(reset=True)["det_hex"]
        t = ms / cnt
        print(f"{scheme} n={n} lanes={lanes} arith={arith}: {t:.3f} ms, {n * s.n_int * FL[scheme] / (t * 1e-3) / 1e12:.2f} TFLOP/s algorithmic")
