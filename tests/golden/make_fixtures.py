#!/usr/bin/env python3
"""Export the reference's GSM test models to .npz fixtures.

Runs only in the build container (needs /root/reference and h5dump); the
produced .npz files are committed so nothing on the GPU box reads the reference.
The datasets/attributes exported are exactly the ones DiscreteModel::from_gsm
reads (reference src/model/gsm.rs:71-81).  Pure data, no code from the reference.
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

H5DUMP = os.environ.get("H5DUMP", "/opt/conda/bin/h5dump")
REF = "/root/reference/test-data"
OUT = os.path.dirname(os.path.abspath(__file__))
DATASETS = ["r", "M_r", "rho", "P", "Gamma_1", "N2", "Omega_rot"]
MODELS = {
    "zams": "test-model-zams.GSM",
    "tams": "test-model-tams.GSM",
    "joel": "joel-test-model.GSM",
}


def attr(path, name):
    out = subprocess.run([H5DUMP, "-m", "%.17g", "-a", "/" + name, path],
                         check=True, capture_output=True, text=True).stdout
    m = re.search(r"\(0\):\s*([-+0-9.eEinfa]+)", out)
    if not m:
        raise RuntimeError(f"cannot parse attribute {name}: {out}")
    return m.group(1)


def dataset(path, name, n):
    with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
        subprocess.run([H5DUMP, "-d", "/" + name, "-b", "LE", "-o", tmp.name, path],
                       check=True, capture_output=True)
        a = np.fromfile(tmp.name, dtype="<f8")
    if a.size != n:
        raise RuntimeError(f"{name}: expected {n} values, got {a.size}")
    return a


def main():
    for key, fname in MODELS.items():
        path = os.path.join(REF, fname)
        n = int(attr(path, "n"))
        arrays = {d: dataset(path, d, n) for d in DATASETS}
        arrays["R_star"] = np.array(float(attr(path, "R_star")))
        arrays["M_star"] = np.array(float(attr(path, "M_star")))
        arrays["n"] = np.array(n, dtype=np.int64)
        out = os.path.join(OUT, f"gsm_{key}.npz")
        np.savez_compressed(out, **arrays)
        print(key, n, os.path.getsize(out), "bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
