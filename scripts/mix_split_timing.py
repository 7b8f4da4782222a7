"""det launch time against the share of frequencies given to the column-owner body of det_mix (FMA build, headline system).
usage: [STORM_ARITH=0] mix_split_timing.py [n ...]      hex_lanes = -p forces det_mix with p % column owners (tuning knob)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
from storm_amd import host
model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
s = host.ErasedSolver(model, 2, 0, "colloc4", model.grid(1))
s.set_option("profile", 1); s.set_option("hex_rounds", 0); s.set_option("arithmetic", int(os.environ.get("STORM_ARITH", "1"))); s.set_option("segments", -1)
for n in [int(a) for a in sys.argv[1:]] or [25001, 50001, 62000, 75000]:
    f = host.linspace(1.0, 25.0, n)
    row = []
    for p in (0, 10, 20, 30, 40, 50, 60, 70, 80, 90):
        s.set_option("hex_lanes", -p if p else 16)
        s.det(f); s.kernel_times(reset=True)
        for _ in range(3):
            s.det(f)
        kt = s.kernel_times(reset=True)
        ms = sum(v[0] for v in kt.values()) / 3
        row.append(f"{p}%: {ms:.2f}")
    print(f"n={n}: " + "  ".join(row), flush=True)
