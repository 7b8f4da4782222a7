"""Flat model file for the C++ tools (tools/many_scans.cpp, tools/storm_batch.cpp) from a tests/golden GSM fixture:
u64 n, f64 R_star, f64 M_star, then r, M_r, rho, P, Gamma_1, N2, Omega_rot (n doubles each).
usage: python tools/make_model_bin.py [tams] [out.bin]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "tams"
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tools", f"{name}_model.bin")
d = np.load(os.path.join(ROOT, "tests", "golden", f"gsm_{name}.npz"))
with open(out, "wb") as f:
    f.write(np.uint64(int(d["n"])).tobytes())
    f.write(np.float64(float(d["R_star"])).tobytes())
    f.write(np.float64(float(d["M_star"])).tobytes())
    for k in ("r", "M_r", "rho", "P", "Gamma_1", "N2", "Omega_rot"):
        f.write(np.ascontiguousarray(d[k], dtype=np.float64).tobytes())
print(out)
