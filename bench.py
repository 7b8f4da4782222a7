#!/usr/bin/env python3
"""bench.py -- scan throughput of the HIP path on BASELINE.json's config 2.

A "step" is one pass of the hot path over the whole batch: for tams (6199 intervals),
ell = 1,2,3 x {Magnus6, Colloc8}: scan_and_optimize over linspace(1, 25, 10^4) at relative
precision 1e-8, then the eigenvector of every mode found (= StormState::scan, reference
src/main.rs:793-838, six times).  Model coefficients are resident in HBM before the timed
region (solvers are created outside it).  value = (frequency x interval) discriminant
evaluations per second over the whole job (scan + refinement evaluations, eigenvector time
included in the denominator), summed over all ranks.

N > 1 (weak scaling): the grid becomes linspace(1, 25, N*10^4); every rank evaluates a contiguous
slice (+1 halo point), brackets are gathered with RCCL (storm_amd/distributed.py).
"""
import argparse
import json
import os
import sys
import time

# The six scans of a step run on six HIP streams; HIP's default of 4 hardware queues per process
# would make two of them wait behind the others' kernels.  Must be set before HIP initialises.
# (The GPU box exports the default 4; 8 is well inside what the pool allows.)
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic flops per (frequency, interval) evaluation, reference form (add/sub/mul/div = 1,
# un-fused), frozen by the oracle's op-counting scalar (tests/test_oracle_golden.py,
# oracle/storm_ref.hpp Counted) for ell != 0 with uniform rotation.  See DESIGN.md.
FLOPS_PER_EVAL = {"colloc2": 610.1, "colloc4": 1396.1, "colloc6": 2694.1, "colloc8": 5048.1,
                  "magnus2": 1486.0, "magnus4": 1848.0, "magnus6": 2562.0, "magnus8": 3900.1}
# HBM bytes per launch of the dominant kernels, from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
# (profiles/r01_pmc_det_hex_colloc8.txt; raw counters in KB, no width correction applied).
TRAFFIC = {"det_hex<colloc8>": 10872.8e3 + 156.25e3}
PEAK_FP64_TFLOPS = 78.6  # MI355X vector FP64, FMA-counted (256 CU x 128 flop/clk x 2.4 GHz); = dense FP64 MFMA peak

# longest jobs first (colloc8 costs ~2x magnus6) so their latency-bound tails overlap the other scans
CASES = [(ell, scheme) for scheme in ("colloc8", "magnus6") for ell in (3, 2, 1)]
N_FREQ_PER_GPU = 10000
PRECISION = 1e-8


def cpu_baseline(npz, budget_s=15.0):
    """The CPU oracle (a port of the reference algorithm, -O2 -ffp-contract=off) on a bounded sample
    of the same workload: the first `k` frequencies of each of the six scans, all host cores."""
    from oracle import storm_ref
    storm_ref.build()
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16), 64)
    model = storm_ref.Model.from_gsm_npz(npz)
    grid = model.field("r_coord")
    freqs = storm_ref.linspace(1.0, 25.0, N_FREQ_PER_GPU)
    solvers = [storm_ref.Solver(model, ell, 0, scheme, grid) for ell, scheme in CASES]
    # calibrate on a small sample, then size the timed sample to ~budget_s
    t0 = time.perf_counter()
    for s in solvers:
        s.det(freqs[:cores], threads=cores)
    dt = max(time.perf_counter() - t0, 1e-3)
    k = int(max(cores, min(N_FREQ_PER_GPU, cores * budget_s / dt)))
    k -= k % cores
    sample = freqs[np.linspace(0, N_FREQ_PER_GPU - 1, k).astype(int)]
    t0 = time.perf_counter()
    for s in solvers:
        s.det(sample, threads=cores)
    dt = time.perf_counter() - t0
    evals = len(solvers) * k * solvers[0].n_int
    return {"value": evals / dt, "unit": "freq*interval evals/s", "cores": cores, "kind": "port",
            "sample": f"{k} of {N_FREQ_PER_GPU} scan frequencies (evenly spaced) of each of the 6 scans "
                      f"(tams, ell=1..3 x magnus6/colloc8), discriminant only, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--threads", type=int, default=6, help="host threads driving independent scans (1 = sequential)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--arithmetic", choices=["parity", "fast"], default="parity",
                    help="headline arithmetic: parity = bit-identical to the reference (default); fast = FMA build")
    ap.add_argument("--no-sweep", action="store_true", help="skip the rotation-sweep leg (storm_sweep)")
    ap.add_argument("--single-mode", action="store_true", help="do not also measure the other arithmetic mode")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (CPU rehearsal of N > 1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()  # (rehearsals may oversubscribe one GPU)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    coll_device = device if args.dist_backend == "nccl" else torch.device("cpu")

    from storm_amd import host, distributed as D
    npz = os.path.join(ROOT, "tests", "golden", "gsm_tams.npz")
    model = host.Model.from_gsm_npz(npz)
    grid = model.field("r_coord")
    n_total = N_FREQ_PER_GPU * world
    freqs = host.linspace(1.0, 25.0, n_total)
    ulp = __import__("storm_amd").relative_to_ulp(PRECISION)

    solvers = []
    for ell, scheme in CASES:
        s = host.ErasedSolver(model, ell, 0, scheme, grid)
        s.set_option("profile", 1)
        solvers.append(s)
    n_int = solvers[0].n_int

    class RawScan:
        """adapter: C-ABI scan on the raw handle with bracket indices (for the sharded gather)"""
        def __init__(self, es):
            import ctypes as C
            from storm_amd import ffi
            self.h, self.C, self.ffi = C.c_void_p(es.raw_handle()), C, ffi

        def scan_and_optimize(self, f, ulp):
            C, ffi = self.C, self.ffi
            f = np.ascontiguousarray(f, dtype=np.float64)
            cap = max(1, f.size)
            out = (ffi.Bracket * cap)()
            idx = (C.c_uint64 * cap)()
            n_out, inv = C.c_size_t(0), C.c_int(0)
            rc = ffi.load_library().storm_scan_and_optimize(self.h, f.ctypes.data_as(C.POINTER(C.c_double)), f.size,
                                                            ulp, out, cap, C.byref(n_out), C.byref(inv), None, idx)
            if rc != 0:
                raise ffi.StormError(rc, ffi.load_library().storm_last_error().decode())
            res = [dict(lo_x=b.lo_x, lo_f=b.lo_f, hi_x=b.hi_x, hi_f=b.hi_f, root=b.root, evals=int(b.evals),
                        index=int(idx[i])) for i, b in enumerate(out[:n_out.value])]
            return res, None, bool(inv.value)

    raws = [RawScan(s) for s in solvers]

    # The six scans are independent jobs on independent handles (each with its own HIP stream):
    # run them from six host threads so the latency-bound phases of one scan (bracket refinement
    # rounds, eigenvector elimination) overlap the throughput-bound step-matrix kernels of the others.
    # ctypes releases the GIL inside the library.  Collectives (N > 1) stay on the main thread.
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=len(solvers)) if args.threads > 1 else None

    def scan_one(k):
        torch.cuda.set_device(local_rank)
        lo, mine = D.local_slice(freqs, world, rank) if world > 1 else (0, freqs)
        res, _, inv = raws[k].scan_and_optimize(mine, ulp) if len(mine) else ([], None, False)
        if res:
            solvers[k].eigenvector(np.array([b["root"] for b in res]))
        return lo, res, inv

    def finish_step(parts):
        modes_global = 0
        for lo, res, inv in parts:
            if world > 1:
                rec, _ = D.gather_brackets(D.pack(res, lo), inv, coll_device)
                modes_global += rec.shape[0]
            else:
                modes_global += len(res)
        return modes_global

    def one_step():
        if pool is not None:
            parts = list(pool.map(scan_one, range(len(solvers))))
        else:
            parts = [scan_one(k) for k in range(len(solvers))]
        return finish_step(parts)

    def run_steps(steps):
        """`steps` passes = steps x 6 scans.  Each of the six handles runs its scans back to back on its own host thread
        and stream, so the latency-bound end of one pass (refinement rounds, eigenvectors of the last scans) overlaps
        the step-matrix kernels of the next pass instead of leaving the GPU idle between passes; the main thread
        completes the passes in order (for N > 1: the gathers of pass s while the workers are already in pass s + 1)."""
        if pool is None:
            modes = 0
            for _ in range(steps):
                modes = one_step()
            return modes
        import queue
        done = [queue.Queue() for _ in solvers]

        def chain(k):
            try:
                for _ in range(steps):
                    done[k].put(scan_one(k))
            except BaseException as e:  # noqa: BLE001 -- handed to the main thread
                done[k].put(e)

        futures = [pool.submit(chain, k) for k in range(len(solvers))]
        modes = 0
        for _ in range(steps):
            parts = [q.get() for q in done]
            for part in parts:
                if isinstance(part, BaseException):
                    raise part
            modes = finish_step(parts)
        for f in futures:
            f.result()
        return modes

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(arith, steps, warmup):
        """Times `steps` passes in arithmetic mode `arith` (0 parity, 1 fast); returns a dict."""
        for s in solvers:
            s.set_option("arithmetic", arith)
        for _ in range(warmup):
            one_step()
        for s in solvers:
            s.stats(reset=True)
            s.kernel_times(reset=True)
        sync()
        t0 = time.perf_counter()
        modes = run_steps(steps)
        sync()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        det_evals = sum(s.stats()["det_evals"] for s in solvers)
        refine = sum(s.stats()["refine_evals"] for s in solvers)
        eigs = sum(s.stats()["eigenvectors"] for s in solvers)
        if world > 1:
            t = torch.tensor([det_evals, refine, eigs], dtype=torch.float64, device=coll_device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            det_evals, refine, eigs = (int(x) for x in t.tolist())
        # Per-kernel device time inside the timed region (HIP events on each solver's launch stream).
        # With six scans in flight these durations overlap each other, so they are reported as they
        # are but NOT used for the roofline.
        per_kind = {}
        for (ell, scheme), s in zip(CASES, solvers):
            for kind, (ms, cnt) in s.kernel_times().items():
                if cnt:
                    d = per_kind.setdefault((kind, scheme), {"ms": 0.0, "launches": 0})
                    d["ms"] += ms
                    d["launches"] += cnt
        kernel_summary = {f"{k}/{sch}": {"ms_total": round(v["ms"], 3), "launches": v["launches"],
                                          "avg_ms": round(v["ms"] / v["launches"], 4)} for (k, sch), v in per_kind.items()}
        # dominant kernel = the one doing most of the arithmetic (launch durations inside the region are
        # contended and would make this choice noisy)
        dom_scheme = max(set(sch for _, sch in CASES), key=lambda sch: FLOPS_PER_EVAL[sch])
        # Roofline of the dominant kernel: the same launch (same solver, same frequency slice, same grid)
        # repeated alone on the device, timed with HIP events on the stream it is launched on.
        roofline = None
        if rank == 0:
            k_dom = max((k for k, (ell, sch) in enumerate(CASES) if sch == dom_scheme), key=lambda k: CASES[k][0])
            sd = solvers[k_dom]
            lo, mine = D.local_slice(freqs, world, rank) if world > 1 else (0, freqs)
            # one launch per batch here (in the timed steps a Colloc8 batch goes out in pieces of one wave round, option
            # "hex_rounds"), so that a launch of this leg is one row of the rocprofv3 kernel stats with the same duration
            sd.set_option("hex_rounds", 0)
            sd.det(mine)
            sd.kernel_times(reset=True)
            reps = 5
            for _ in range(reps):
                sd.det(mine)
            kt = {k: v for k, v in sd.kernel_times(reset=True).items() if v[1]}
            sd.set_option("hex_rounds", 1)
            kind = max(kt.items(), key=lambda kv: kv[1][0])[0]
            ms, cnt = kt[kind]
            flops_per_rep = len(mine) * n_int * FLOPS_PER_EVAL[dom_scheme]
            if kind == "build_steps":  # the ordered product (112 flop/eval) runs in the chain kernel
                flops_per_rep = len(mine) * n_int * (FLOPS_PER_EVAL[dom_scheme] - 112.0)
            achieved = flops_per_rep * reps / (ms * 1e-3) / 1e12
            roofline = {"bound": "mfma", "pipe": "fp64-valu", "kernel": f"{kind}<{dom_scheme}>", "achieved": round(achieved, 3),
                        "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP64_TFLOPS, 4),
                        "traffic": TRAFFIC.get(f"{kind}<{dom_scheme}>"), "avg_launch_ms": round(ms / cnt, 4),
                        "launches_timed": cnt, "flops_per_launch": flops_per_rep * reps / cnt,
                        "note": NOTE_FAST if arith else NOTE_PARITY}
        return {"elapsed": elapsed, "modes": modes, "det_evals": det_evals, "refine": refine, "eigs": eigs,
                "kernels": kernel_summary, "roofline": roofline}

    NOTE_PARITY = ("algorithmic un-fused flops (oracle op count x evals per launch) / HIP-event duration of the kernel "
                   "run alone; parity arithmetic = the reference's separate mul and add (bit-exact results), whose "
                   "measured chip-wide issue ceiling is 38.1 TFLOP/s (tools/fp64_microbench.hip) = 0.485 of the "
                   "FMA-counted peak; bound: compute, priced against the dense FP64 peak of MI355X_MICROARCH.md (78.6 TFLOP/s, the "
                   "same figure for MFMA and for the vector pipe this kernel runs on -- `pipe`; the path is "
                   "4x4 products in a strict order and is not reshaped for MFMA); traffic = HBM "
                   "bytes per launch from rocprofv3 PMC passes (profiles/)")
    NOTE_FAST = ("same kernel built with FMA contraction (option arithmetic=1): results agree with the reference to "
                 "rounding (tests/test_hip_fast_mode.py: identical bracket indices, roots within 1e-10); algorithmic "
                 "flops are still the reference's un-fused count")
    main_arith = 1 if args.arithmetic == "fast" else 0
    res = measure(main_arith, args.steps, args.warmup)
    other = measure(1 - main_arith, max(1, min(args.steps, 2)), 1) if not args.single_mode else None
    elapsed, modes, det_evals, refine, eigs = res["elapsed"], res["modes"], res["det_evals"], res["refine"], res["eigs"]

    # Next-row leg (outside the timed region and outside `value`): eigenvectors + device
    # post-processing of one scan's modes with nothing but the mode ids crossing PCIe (storm_modes_batch).
    post = None
    if rank == 0:
        for s in solvers:
            s.set_option("arithmetic", 0)
        sd = solvers[0]
        lo, mine = D.local_slice(freqs, world, rank) if world > 1 else (0, freqs)
        roots = np.array([b["root"] for b in raws[0].scan_and_optimize(mine, ulp)[0]])
        if roots.size:
            pp = host.PostProcessor(model.eval(grid))
            pp.modes(sd, roots)
            reps = 5
            sd.kernel_times(reset=True)
            t0 = time.perf_counter()
            for _ in range(reps):
                ids = pp.modes(sd, roots)["ids"]
            wall = (time.perf_counter() - t0) / reps
            kt = sd.kernel_times(reset=True)
            t0 = time.perf_counter()
            for _ in range(reps):
                sd.eigenvector(roots)
            wall_eig = (time.perf_counter() - t0) / reps
            k_ms = kt["postprocess"][0] / reps
            alg = roots.size * grid.size * 80.0   # terms: 32 B vector + 8 B out; modes: 8 B term + 32 B vector
            post = {"modes": int(roots.size), "n_pts": int(grid.size),
                    "ms_eigenvectors_plus_ids": round(wall * 1e3, 3), "ms_eigenvectors_to_host": round(wall_eig * 1e3, 3),
                    "postprocess_kernels_ms": round(k_ms, 4), "eigen_kernels_ms": round(kt["eigen"][0] / reps, 3),
                    "algorithmic_GB_per_s": round(alg / (k_ms * 1e-3) / 1e9, 1) if k_ms else None, "hbm_peak_GB_per_s": 8000,
                    "radial_orders": [int(ids[0, 2]), int(ids[-1, 2])],
                    "note": "latency-bound at this batch size: one wave per mode sums the norm integrand in index "
                            "order (bit-exact parity), ~0.1 ms regardless of the number of modes"}
    # Next-row leg (outside the timed region and outside `value`): a rotation sweep of many small
    # scans, 8 rotation rates x 8 (ell, m) systems x 300 frequencies on the tams model with Colloc4 (the CLI default), in
    # lock-step through storm_sweep: solver construction + scan + refinement + eigenvectors + mode ids.
    sweep = None
    if rank == 0 and not args.no_sweep:
        combos = [(0, 0), (2, 0), (4, 0), (1, 1), (1, -1), (2, 1), (2, -1), (3, 0)]
        rates = [0.01 + 0.002 * k for k in range(8)]
        rmodels = []
        for rot in rates:
            mdl = host.Model.from_gsm_npz(npz)
            mdl.set_rotation_constant(rot)
            rmodels.append(mdl)
        systems = [(rmodels[i], ell, m) for i in range(len(rates)) for ell, m in combos]
        sgrid = host.linspace(1.0, 25.0, 300)

        def sweep_once():
            t0 = time.perf_counter()
            sw = host.SolverSweep(systems, "colloc4", grid)
            t1 = time.perf_counter()
            r = sw.scan_and_optimize([sgrid] * len(systems), host.Precision.Relative(1e-8))
            t2 = time.perf_counter()
            posts = [host.PostProcessor(mdl.eval(grid)) for mdl in rmodels]
            found = sw.modes([posts[k // len(combos)] for k in range(len(systems))],
                             [np.array([b["root"] for b in br]) for br, _ in r])
            t3 = time.perf_counter()
            return sum(len(x["ids"]) for x in found), (t1 - t0, t2 - t1, t3 - t2), sw.stats()

        sweep_once()
        n_modes, (t_new, t_scan, t_modes), st = sweep_once()
        total = t_new + t_scan + t_modes
        sweep = {"systems": len(systems), "frequencies_per_scan": int(sgrid.size), "scheme": "colloc4", "modes": int(n_modes),
                 "ms_per_scan": round(total / len(systems) * 1e3, 4), "modes_per_s": round(n_modes / total, 1),
                 "ms_create_scan_modes": [round(t_new * 1e3, 2), round(t_scan * 1e3, 2), round(t_modes * 1e3, 2)],
                 "refine_rounds": st["refine_rounds"], "launches": st["launches"],
                 "note": "all systems advance in lock-step: one launch per phase over every (system, frequency) pair "
                         "or mode; bit-identical to one handle per system (tests/test_hip_sweep.py); one handle per "
                         "system on 16 host threads: tools/many_scans.cpp, DESIGN.md section 11"}
    # The same kind of workload driven from the reference's batch-input language by tools/storm_batch (a child process):
    # 64 rotation rates x (8 scans + post-process + deform + perturb-deformed), groups of 32 blocks, two groups in flight.
    batch = None
    tool = os.path.join(ROOT, "tools", "storm_batch.bin")
    if rank == 0 and not args.no_sweep and os.path.exists(tool):
        import re
        import subprocess
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            mb, bt = os.path.join(tmp, "tams_model.bin"), os.path.join(tmp, "batch.txt")
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_model_bin.py"), "tams", mb], check=True,
                           stdout=subprocess.DEVNULL)
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_sweep_batch.py"), bt, "64"], check=True)
            r = subprocess.run([tool, mb, bt, os.path.join(tmp, "out.txt"), "--group", "32", "--summary", "--device", str(local_rank)],
                               capture_output=True, text=True, timeout=300)
            mt = re.search(r"(\d+) blocks, (\d+) scans, (\d+) modes in ([0-9.]+) ms", r.stderr)
            if r.returncode == 0 and mt:
                nb, ns, nm, ms = int(mt.group(1)), int(mt.group(2)), int(mt.group(3)), float(mt.group(4))
                batch = {"rotation_rates": nb, "scans": ns, "modes": nm, "wall_ms": ms, "ms_per_rotation_rate": round(ms / nb, 3),
                         "modes_per_s": round(nm / ms * 1e3, 1),
                         "note": "per rate: set-rotation-constant, 8 scans (300 / 100 frequencies), post-process, deform, "
                                 "perturb-deformed (coupling integrals of ~238 modes), records with one checksum per "
                                 "matrix; includes process start and first-touch allocations; DESIGN.md section 11c"}
    kernel_summary, roofline = res["kernels"], res["roofline"]

    if rank == 0:
        value = det_evals * n_int / elapsed
        out = {
            "metric": "scan throughput: freq x gridpoint evals/sec (whole job: scan + bracket refinement + eigenvectors)",
            "value": value, "unit": "freq*interval evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: test-model-tams (6199 intervals), ell=1..3, m=0, Magnus6 + "
                                   "Colloc8, linspace(1,25,10^4 per GPU), precision 1e-8, + eigenvector per mode",
                       "n_freq_total": n_total, "n_int": n_int, "scans": len(CASES),
                       "parallelism": f"freq-slice x{world}", "concurrent_scans": args.threads if args.threads > 1 else 1,
                       "passes_back_to_back_per_handle": args.threads > 1},
            "modes_per_s": modes * args.steps / elapsed if False else (eigs / elapsed),
            "modes_per_step": modes, "det_evals": det_evals, "refine_evals": refine,
            "arithmetic": args.arithmetic, "kernels": kernel_summary, "roofline": roofline,
            "post_processing": post,
            "rotation_sweep": sweep,
            "batch_pipeline": batch,
        }
        if other is not None:
            oname = "parity" if main_arith else "fast"
            out[f"{oname}_arithmetic"] = {
                "value": other["det_evals"] * n_int / other["elapsed"], "unit": "freq*interval evals/s",
                "ms_per_step": other["elapsed"] / max(1, min(args.steps, 2)) * 1e3,
                "modes_per_s": other["eigs"] / other["elapsed"], "modes_per_step": other["modes"],
                "roofline": other["roofline"]}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(npz, args.cpu_budget)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()  # rank 0 ran the extra legs: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
