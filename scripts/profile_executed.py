"""The launches whose executed FP64 work bench.py's roofline entries are priced on, in a fixed order, for rocprofv3 passes
(--kernel-trace for durations, --pmc for the FP64 instruction counters and HBM bytes): scripts/gpu_r3_executed.sh.

Every library call is preceded by a SENTINEL dispatch (one torch element-wise kernel), and the list of calls is written as a
manifest next to the profiler's output, so tools/executed_flops.py can cut the dispatch stream into calls without guessing from
kernel names or grid sizes.

usage: profile_executed.py <group> <manifest.json> [reps]
groups: headline  BASELINE config 3 (n = 3 polytrope, 10^4 points, ell = 2, Colloc4), both builds, the whole grid of 10^5
                  frequencies and the slices a strong-scaled run hands one rank at N = 2, 4, 8
        schemes   tams (6199 intervals), ell = 3, all eight schemes, both builds, 10^5 frequencies in one launch
        config2   tams, ell = 3, Colloc8 and Magnus6, 10^4 frequencies (bench.py's config-2 leg), both builds
        coupling  storm_coupling_matrices: M = 512 modes of one degree (bench.py's leg) and M = 1024 of two degrees, both builds
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from storm_amd import host, ffi  # noqa: E402

group, manifest_path = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
SCHEMES = ["colloc2", "colloc4", "colloc6", "colloc8", "magnus2", "magnus4", "magnus6", "magnus8"]
manifest = []
mark = torch.zeros(64, device="cuda")


def call(fn, **what):
    """sentinel, then one library call; `what` goes to the manifest"""
    mark.add_(1.0)
    torch.cuda.synchronize()
    fn()
    manifest.append(what)


def det_items(solver, label, scheme, n_list, n_int):
    solver.set_option("hex_rounds", 0)   # the whole batch in ONE launch of the dominant kernel
    for arith in (0, 1):
        solver.set_option("arithmetic", arith)
        for n in n_list:
            f = host.linspace(1.0, 25.0, n)
            for r in range(1 + reps):
                call(lambda: solver.det(f), family="det", group=group, label=label, scheme=scheme,
                     arithmetic="fast" if arith else "parity", n_freq=n, n_int=n_int, warm=(r == 0))


TAMS = os.path.join(ROOT, "tests", "golden", "gsm_tams.npz")
if group == "headline":
    model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, 6.89684861937 / 9999.0)
    s = host.ErasedSolver(model, 2, 0, "colloc4", model.grid(1))
    # (storm_amd.distributed.local_slice: ceil(n / G) points + 1 halo point per rank)
    det_items(s, "config3: n=3 polytrope, 10^4 points, ell=2", "colloc4", [100000, 50001, 25001, 12501], s.n_int)
elif group == "schemes":
    model = host.Model.from_gsm_npz(TAMS)
    for scheme in SCHEMES:
        s = host.ErasedSolver(model, 3, 0, scheme, model.field("r_coord"))
        det_items(s, "tams, ell=3", scheme, [100000], s.n_int)
        del s
elif group == "config2":
    model = host.Model.from_gsm_npz(TAMS)
    for scheme in ("colloc8", "magnus6"):
        s = host.ErasedSolver(model, 3, 0, scheme, model.field("r_coord"))
        det_items(s, "config2: tams, ell=3", scheme, [10000], s.n_int)
        del s
elif group == "coupling":
    model = host.Model.from_gsm_npz(TAMS)
    model.set_rotation_constant(0.1)
    grid = model.field("r_coord")
    on_grid = model.eval(grid)
    pp = host.PostProcessor(on_grid)
    fields, ells, freqs = [], [], []
    for ell in (0, 2):
        s = host.ErasedSolver(model, ell, 0, "colloc4", grid)
        res, _ = s.scan_and_optimize(host.linspace(1.0, 25.0, 2000), host.Precision.Relative(1e-8))
        roots = np.array([r["root"] for r in res])
        got = pp.modes(s, roots, want_fields=True)
        fields.append(got["fields"]); ells += [ell] * roots.size; freqs += list(roots)
    fields, ells, freqs = np.concatenate(fields), np.array(ells, dtype=np.uint64), np.array(freqs)
    x = grid
    metric = {"alpha": 1e-3 * x ** 2, "dalpha": 2e-3 * x ** 2, "ddalpha": 4e-3 * x ** 2,
              "beta": -1.2e-3 * x ** 2, "dbeta": -2.4e-3 * x ** 2, "ddbeta": -4.8e-3 * x ** 2}
    raw = ffi.HipPostproc({k: on_grid.field(k) for k in ffi.MODEL_FIELDS})
    two = np.flatnonzero(ells == 2)
    cases = [("M=512 modes of one degree (ell=2), as bench.py's coupling leg", 512, two[np.arange(512) % two.size]),
             ("M=1024 modes of ell=0,2 grouped by degree", 1024, None)]
    for label, M, sel in cases:
        if sel is None:
            sel = np.arange(M) % freqs.size
            sel = sel[np.argsort(ells[sel], kind="stable")]
        f, e, fl = freqs[sel], ells[sel], np.ascontiguousarray(fields[sel])
        for arith in (0, 1):
            raw.set_option("arithmetic", arith)
            for r in range(1 + reps):
                call(lambda: raw.coupling_matrices(e, f, fl, 0, metric), family="coupling", group=group, label=label,
                     arithmetic="fast" if arith else "parity", modes=M, n_int=int(grid.size - 1), warm=(r == 0))
else:
    raise SystemExit(f"unknown group {group}")
mark.add_(1.0)   # closing sentinel
torch.cuda.synchronize()
with open(manifest_path, "w") as fh:
    json.dump(manifest, fh)
print("ok", group, len(manifest), "calls")
