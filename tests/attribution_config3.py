"""Attribution experiment for the FMA build's discriminant error on BASELINE config 3 (not collected by pytest; run on a GPU
box: python tests/attribution_config3.py > gpurun_out/attribution_config3.txt).  VERDICT round 2, item 2(b).

On the 10^4-point n = 3 polytrope (ell = 2, Colloc4) the FMA build's discriminant differs from the reference's by up to
4.4e-7 of the local scale, where every other model stays below 1e-9.  Two questions:
 (1) is that the conditioning of the reference's own value?  Measured with the ORACLE alone: perturb one input field
     (a_star, gamma1, v, u, c1) by ONE ulp per point (seeded random signs) and see how far the oracle's discriminant moves;
 (2) or is it one of the FMA build's re-formulations?  Variant libraries (scripts/build_fast_variant.sh) with the
     re-formulations switched off one at a time, and all at once (contraction only), each compared with the oracle.
Every variant runs in a child process of its own (STORM_HIP_LIBRARY)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
XI1_N3 = 6.89684861937
N_FREQ = 2000
VARIANTS = [("FMA build as shipped", None), ("IEEE division instead of v_rcp_f64 + 2 Newton steps", "ieee_div"),
            ("adjugate inverse instead of the LU stage solve", "no_lu"),
            ("the reference's 16 cofactors instead of the Laplace expansion", "ref_inverse"),
            ("wrapper as -(jump - I)^-1 (jump + I) on jump = c S instead of -I - (2/c)(S - I/c)^-1", "no_fold"),
            ("all four switches", "all_switches"),
            ("the reference's formulas, FMA contraction only", "contraction_only")]


def local_scale(d):
    a = np.abs(d)
    return np.maximum.reduce([np.roll(a, 1), a, np.roll(a, -1)])


def child(scheme):
    """discriminants of the FMA build of whatever library STORM_HIP_LIBRARY names"""
    from oracle import storm_ref as oracle
    import storm_amd
    model = oracle.Model.polytrope(3.0, 5.0 / 3.0, XI1_N3 / 9999.0)
    o = oracle.Solver(model, 2, 0, scheme, model.grid(1))
    h = storm_amd.HipSolver(o.scheme, o.ell, o.m, o.system(), o.n_int, o.p)
    h.set_option("arithmetic", 1)
    d = h.det(oracle.linspace(1.0, 25.0, N_FREQ))
    print("DETS " + json.dumps([float(x).hex() for x in d]))


def main():
    from oracle import storm_ref as oracle
    oracle.build()
    f = oracle.linspace(1.0, 25.0, N_FREQ)
    for scheme in ("colloc4",):
        model = oracle.Model.polytrope(3.0, 5.0 / 3.0, XI1_N3 / 9999.0)
        grid = model.grid(1)
        ref = oracle.Solver(model, 2, 0, scheme, grid).det(f, threads=16)
        scale = local_scale(ref)
        print(f"== {scheme}, n = 3 polytrope, {grid.size} points, ell = 2, {N_FREQ} frequencies on [1, 25]")
        print("(1) the oracle's own discriminant when ONE input field moves by one ulp per point (random signs, seed 1):")
        rng = np.random.default_rng(1)
        for name in ("v", "u", "a_star", "c1", "gamma1"):
            m2 = oracle.Model.polytrope(3.0, 5.0 / 3.0, XI1_N3 / 9999.0)
            v = m2.field(name)
            up = rng.integers(0, 2, v.size).astype(bool)
            m2.set_field(name, np.where(up, np.nextafter(v, np.inf), np.nextafter(v, -np.inf)))
            d = oracle.Solver(m2, 2, 0, scheme, grid).det(f, threads=16)
            e = np.abs(d - ref) / scale
            print(f"    {name:8s} max |d - ref| / local scale = {e.max():.3e}   median {np.median(e):.3e}")
        one = np.abs(oracle.Solver(model, 2, 0, scheme, grid).det(np.nextafter(f, 100.0), threads=16) - ref) / scale
        print(f"    frequency + one ulp: max {one.max():.3e}")
        print("(2) the FMA build and its variants against the oracle:")
        for label, name in VARIANTS:
            env = dict(os.environ)
            if name:
                env["STORM_HIP_LIBRARY"] = os.path.join(ROOT, "storm_amd", "lib", "variants", f"libstorm_hip_{name}.so")
                if not os.path.exists(env["STORM_HIP_LIBRARY"]):
                    print(f"    {label}: variant library not built")
                    continue
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", scheme], env=env, capture_output=True,
                               text=True, timeout=300)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("DETS ")]
            if r.returncode or not line:
                print(f"    {label}: failed\n{r.stderr[-500:]}")
                continue
            d = np.array([float.fromhex(x) for x in json.loads(line[0][5:])])
            e = np.abs(d - ref) / scale
            print(f"    {label:90s} max {e.max():.3e}  median {np.median(e):.3e}  signs equal {np.array_equal(np.signbit(d), np.signbit(ref))}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        main()
