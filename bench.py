#!/usr/bin/env python3
"""bench.py -- scan throughput of the HIP path on BASELINE.json's config 3 (the north_star target).

Headline workload: a synthetic 10^4-point stellar model (the n = 3, Gamma_1 = 5/3 polytrope of the reference's
`input-poly`, src/model/polytrope.rs, integrated with dx = xi_1/9999), ell = 2, m = 0, Colloc4 (the CLI default scheme),
scan over linspace(1, 25, 10^5) at relative precision 1e-8, then the eigenvector of every mode found -- exactly what
StormState::scan does (reference src/main.rs:793-838).  A "step" is one such scan, whole job: discriminants on the
grid, sign-swap filter, bracket refinement, eigenvectors.  The coefficient planes are resident in HBM before the
timed region (the solver is created outside it).

value = (frequency x interval) discriminant evaluations per second over the whole job (scan + refinement evaluations in
the numerator, everything incl. eigenvectors in the denominator), summed over all ranks.

N > 1 is STRONG scaling: the same 10^5-point grid is cut into N contiguous slices (+1 halo point), one per rank/GPU;
every rank scans and refines its slice in one library call, then the refined brackets are gathered over RCCL (ONE
all_gather, in flight while the rank computes the eigenvectors of its own modes; storm_amd/distributed.py:
strong_scan_and_optimize(deal=False) -- the two-collective protocol that deals the brackets out cyclically is its
deal=True).

Both arithmetic builds run in the same invocation.  The headline is the FMA build ("fast": v_fma_f64 allowed; results
agree with the reference to rounding, north_star's own contract: bracket indices bit-identical -- asserted below on the
full grid before anything is timed -- and roots within 1e-10); the parity build (separate mul and add, every f64 output
bit-identical to the reference) is reported next to it under "parity_arithmetic".
"""
import argparse
import json
import os
import sys
import time

# The six concurrent scans of the config-2 leg run on six HIP streams; HIP's default of 4 hardware queues per process
# would make two of them wait behind the others' kernels.  Must be set before HIP initialises.
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic flops per (frequency, interval) evaluation, reference form (add/sub/mul/div = 1, un-fused), frozen by the
# oracle's op-counting scalar (tests/test_oracle_golden.py, oracle/storm_ref.hpp Counted) for ell != 0 with uniform
# rotation.  See DESIGN.md section 5.
FLOPS_PER_EVAL = {"colloc2": 610.1, "colloc4": 1396.1, "colloc6": 2694.1, "colloc8": 5048.1,
                  "magnus2": 1486.0, "magnus4": 1848.0, "magnus6": 2562.0, "magnus8": 3900.1}
# What the kernels EXECUTE -- FP64 flops per (frequency, interval) evaluation and HBM bytes per launch -- comes from rocprofv3
# counter passes (SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS,MFMA}_F64, FETCH_SIZE, WRITE_SIZE; scripts/gpu_r3_executed.sh condensed by
# tools/executed_flops.py into profiles/r03_executed_flops.json, which records the commit it was taken at).  roofline.frac is
# priced on THAT count; the reference formulation's count (FLOPS_PER_EVAL above: what the CPU reference would execute for the
# same evaluations) gives the separately named `reference_equivalent` rate, which can pass the peak because the FMA build
# re-formulates the step (LU stage solve, folded scalings, re-associated inverse: DESIGN.md section 5).
with open(os.path.join(ROOT, "profiles", "r03_executed_flops.json")) as _fh:
    EXECUTED = json.load(_fh)


def executed_entry(group, arithmetic, **match):
    """the counter-backed entry of profiles/r03_executed_flops.json for this launch (nearest batch size if not the same)"""
    n_freq = match.pop("n_freq", None)
    cand = [e for e in EXECUTED["entries"] if e["group"] == group and e["arithmetic"] == arithmetic
            and all(e.get(k) == v for k, v in match.items())]
    if not cand:
        raise SystemExit(f"profiles/r03_executed_flops.json has no entry for {group} {arithmetic} {match}")
    if n_freq is not None:
        cand.sort(key=lambda e: abs(e["n_freq"] - n_freq))
    return cand[0]


def pipe_breakdown(e, launch_ms):
    """Where 1 - frac goes, from the same counters: the FP64 pipe's busy share at the clock the profile saw, that clock against
    the 2.4 GHz the peak assumes, and the share of pipe slots that hold one flop (mul, add, rcp) instead of two (fma, mfma).
    Cycles per wave instruction on a SIMD: v_fma/mul/add_f64 4, v_rcp_f64 and friends 16, v_mfma_f64_4x4x4 16,
    v_mfma_f64_16x16x4 64 (tools/fp64_microbench.hip, tools/mfma64_microbench.hip)."""
    k = e["kernels"][e["dominant_kernel"]]
    c = k["counters_per_call"]
    g = lambda n: float(c.get(n, 0.0))
    mfma_cycles, mfma_flops = (64, 2048) if "coupling_matrix" in e["dominant_kernel"] else (16, 512)
    fma, mul, add, trans, mfma = (g("SQ_INSTS_VALU_FMA_F64"), g("SQ_INSTS_VALU_MUL_F64"), g("SQ_INSTS_VALU_ADD_F64"),
                                  g("SQ_INSTS_VALU_TRANS_F64"), g("SQ_INSTS_VALU_MFMA_F64"))
    busy = 4 * (fma + mul + add) + 16 * trans + mfma_cycles * mfma
    flops = 128 * fma + 64 * (mul + add + trans) + mfma_flops * mfma
    if not busy or not g("SQ_BUSY_CYCLES"):
        return None
    clock_ghz = g("SQ_BUSY_CYCLES") / 32 / k["us_per_call"] / 1e3      # (SQ_BUSY_CYCLES sums over 8 XCDs x 4 shader engines)
    simd_cycles = launch_ms * 1e-3 * clock_ghz * 1e9 * 1024
    return {"fp64_pipe_busy": round(busy / simd_cycles, 3), "clock_ghz_in_profile": round(clock_ghz, 3),
            "clock_over_peak_clock": round(clock_ghz / 2.4, 3), "flops_per_pipe_cycle_over_32": round(flops / busy / 32, 3),
            "product": round(busy / simd_cycles * clock_ghz / 2.4 * flops / busy / 32, 3),
            "what": "frac ~= fp64_pipe_busy x clock_over_peak_clock x flops_per_pipe_cycle_over_32 (pipe cycles from the "
                    "instruction counters, the clock from SQ_BUSY_CYCLES of the profiled launch, live launch time)"}


def executed_source(e):
    return {"file": "profiles/r03_executed_flops.json", "commit": EXECUTED.get("commit"), "group": e["group"],
            "kernel": e["dominant_kernel"], "units_per_call": e["units_per_call"],
            "executed_flops_per_unit": round(e["executed_flops_per_unit"], 2),
            "counters_per_call": e["kernels"][e["dominant_kernel"]]["counters_per_call"]}


PEAK_FP64_TFLOPS = 78.6     # MI355X vector FP64, FMA-counted (256 CU x 128 flop/clk x 2.4 GHz) = the dense FP64 MFMA peak
UNFUSED_TFLOPS = 38.1       # measured chip-wide ceiling of separate v_mul_f64 / v_add_f64 (tools/fp64_microbench.hip)
HBM_PEAK_GBS = 8000.0

N_FREQ = 100000
PRECISION = 1e-8
XI1_N3 = 6.89684861937      # first zero of the n = 3 Lane-Emden function


def polytrope_model(host):
    model = host.Model.construct_polytrope(3.0, 5.0 / 3.0, XI1_N3 / 9999.0)
    return model, model.grid(1)


def cpu_baseline(budget_s):
    """The CPU oracle (a port of the reference algorithm, g++ -O2 -ffp-contract=off) on the same job: scan + bracket
    refinement + eigenvectors (what BASELINE.md section 3 asks for), frequencies spread over std::threads the way rayon
    spreads them in the reference.  All host cores, and 1 thread (the reference's own benchmark protocol,
    scripts/benches:32, RAYON_NUM_THREADS=1).  Bounded: the grid is thinned to a coarser linspace over the same range
    (the same modes are found) so that each figure takes about `budget_s` seconds."""
    from oracle import storm_ref
    storm_ref.build()
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16), 64)
    model = storm_ref.Model.polytrope(3.0, 5.0 / 3.0, XI1_N3 / 9999.0)
    grid = model.grid(1)
    solver = storm_ref.Solver(model, 2, 0, "colloc4", grid)
    ulp = storm_ref.relative_to_ulp(PRECISION)

    def job(n_freq, threads):
        f = storm_ref.linspace(1.0, 25.0, n_freq)
        t0 = time.perf_counter()
        res, _, _ = solver.scan_and_optimize(f, ulp, threads=threads)
        t1 = time.perf_counter()
        if res:
            solver.eigenvector(np.array([r["root"] for r in res]), threads=threads)
        t2 = time.perf_counter()
        evals = (n_freq + sum(r["evals"] for r in res)) * solver.n_int
        return evals, t2 - t0, t1 - t0, len(res)

    out = {}
    for label, threads in (("all", cores), ("one", 1)):
        probe = 64 * threads
        evals, dt, _, _ = job(probe, threads)
        rate = evals / dt
        n_freq = int(min(N_FREQ, max(probe, budget_s * rate / solver.n_int)))
        evals, dt, dt_scan, modes = job(n_freq, threads)
        out[label] = {"value": evals / dt, "unit": "freq*interval evals/s", "cores": threads, "kind": "port",
                      "modes_per_s": modes / dt, "seconds": round(dt, 2),
                      "sample": f"whole job (scan + refinement + eigenvectors of {modes} modes) on linspace(1, 25, {n_freq}) "
                                f"instead of 10^5 points, same model/ell/scheme/precision; {dt_scan:.1f} s scan+refine, "
                                f"{dt - dt_scan:.1f} s eigenvectors"}
    res = out["all"]
    res["one_thread"] = out["one"]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=8.0, help="seconds of CPU work per cpu_baseline figure")
    ap.add_argument("--arithmetic", choices=["parity", "fast"], default="fast",
                    help="headline arithmetic: fast = FMA build (default, north_star's tolerance contract); parity = "
                         "bit-identical to the reference")
    ap.add_argument("--legs", choices=["headline", "all"], default="all",
                    help="all (default, N = 1 only): also the config-2 job, Polytrope0, eigenvector/post-processing/sweep legs")
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

    import storm_amd
    from storm_amd import host, ffi, distributed as D
    ulp = storm_amd.relative_to_ulp(PRECISION)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(xs):
        if world == 1:
            return xs
        t = torch.tensor(xs, dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [int(v) for v in t.tolist()]

    # ------------------------------------------------------------------ headline: BASELINE config 3
    model, grid = polytrope_model(host)
    es = host.ErasedSolver(model, 2, 0, "colloc4", grid)
    es.set_option("profile", 1)
    h = ffi.HipSolver.borrow(es.raw_handle(), owner=es)
    n_int = es.n_int
    # (the grid lives in pinned host memory: the upload inside the timed region is then one DMA, not a staged copy)
    freqs = torch.from_numpy(host.linspace(1.0, 25.0, N_FREQ)).pin_memory().numpy()
    lo_idx, mine = D.local_slice(freqs, world, rank) if world > 1 else (0, freqs)

    def result_block(n_doubles):
        """The caller's eigenvector block: pinned host memory, so that the library copies into it by DMA."""
        return torch.empty(n_doubles, dtype=torch.float64, pin_memory=True).numpy()

    vec_buf = result_block(64 * (n_int + 1) * 4)  # reused from scan to scan

    def roots_of(brackets):
        """converged frequencies of a bracket list (dicts) or of a record array (storm_amd.distributed, column 4)"""
        return brackets[:, 4].copy() if isinstance(brackets, np.ndarray) else np.array([b["root"] for b in brackets])

    def indices_of(brackets):
        return brackets[:, 6].astype(np.int64).tolist() if isinstance(brackets, np.ndarray) else [b["index"] for b in brackets]

    def eigenvectors(brackets):
        nonlocal vec_buf
        if len(brackets):
            need = len(brackets) * (n_int + 1) * 4
            if vec_buf.size < need:
                vec_buf = result_block(need)
            h.eigenvector(roots_of(brackets), out=vec_buf)

    phase_s = {}

    def one_scan(timings=None):
        """StormState::scan on this rank's share; returns the global bracket list."""
        if world == 1 and timings is None:
            res, _, _ = h.scan_and_optimize(freqs, ulp, cap=8192, want_dets=False)  # (the reference returns brackets only)
            eigenvectors(res)
            return res
        # N > 1: scan + refinement of the slice in one call, one gather overlapped with the eigenvectors; brackets travel as
        # record arrays (no per-bracket Python inside the step).  (The phase-timing pass at N = 1: the two-call form.)
        res, _, _ = D.strong_scan_and_optimize(h, freqs, ulp, device=coll_device, after_refine=eigenvectors,
                                               timings=timings, records=True, deal=False)
        return res

    # Before anything is timed: the FMA build must find the parity build's brackets, index for index, on the full grid.
    found = {}
    for arith, name in ((0, "parity"), (1, "fast")):
        h.set_option("arithmetic", arith)
        found[name] = one_scan()
    idx_p, idx_f = indices_of(found["parity"]), indices_of(found["fast"])
    if idx_p != idx_f:
        raise SystemExit(f"fast arithmetic changed the bracket index list on the full grid: {idx_p} vs {idx_f}")
    rp, rf = roots_of(found["parity"]), roots_of(found["fast"])
    root_diff = float(np.max(np.abs(rf - rp) / np.abs(rp))) if rp.size else 0.0
    if root_diff > 1e-10:
        raise SystemExit(f"fast arithmetic moved a root by {root_diff:.3e} relative (> 1e-10)")
    identity = {"bracket_indices_identical": True, "brackets": len(idx_p), "max_root_rel_diff": root_diff,
                "grid_points": N_FREQ, "checked": "in this run, before timing, against the parity build (which is "
                "bit-identical to the reference: tests/test_hip_parity.py, tests/test_hip_api.py)"}

    NOTE = {0: "achieved = EXECUTED FP64 flops (rocprofv3 counters: v_fma_f64 x 2, v_mul/v_add/transcendental x 1 per lane) per "
               "launch / HIP-event duration of ONE whole-batch launch run alone; parity arithmetic = the reference's separate mul "
               f"and add (every f64 output bit-exact), whose measured chip-wide issue ceiling is {UNFUSED_TFLOPS} TFLOP/s "
               f"(tools/fp64_microbench.hip), i.e. frac <= {UNFUSED_TFLOPS / PEAK_FP64_TFLOPS:.3f} by construction; the counters do not "
               "see the helper instructions of the IEEE divisions (v_div_scale / v_div_fmas / v_div_fixup), which is why the "
               "executed count is below the reference formulation's 1396.1 although the arithmetic is the reference's",
            1: "achieved = EXECUTED FP64 flops (rocprofv3 counters: v_fma_f64 x 2, v_mul/v_add/v_rcp x 1 per lane, "
               "v_mfma_f64_4x4x4 x 512 per wave) per launch / HIP-event duration of ONE whole-batch launch run alone.  The FMA "
               "build re-formulates the step (LU stage solve, folded scalings, re-associated inverse, ordered product on "
               "v_mfma_f64_4x4x4: DESIGN.md section 5) and executes ~770 flops per evaluation where the reference's formulation "
               "has 1396.1; `reference_equivalent` is the rate in the reference's count (comparable between builds and rounds; "
               "not a hardware fraction, it can pass the peak).  FP64 MFMA shares the vector FP64 pipe and its peak on gfx950 "
               "(tools/mfma64_microbench.hip)"}

    def roofline_of(arith):
        """The dominant kernel (det_hex<colloc4>, all of this rank's grid points in ONE launch) run alone, timed with
        HIP events on the stream it is launched on (storm_solver_kernel_times)."""
        h.set_option("arithmetic", arith)
        h.set_option("hex_rounds", 0)
        h.det(mine)
        h.kernel_times(reset=True)
        reps = 5
        for _ in range(reps):
            h.det(mine)
        kt = {k: v for k, v in h.kernel_times(reset=True).items() if v[1]}
        h.set_option("hex_rounds", 1)
        kind = max(kt.items(), key=lambda kv: kv[1][0])[0]
        ms, cnt = kt[kind]
        evals_per_launch = len(mine) * n_int * reps / cnt
        ex = executed_entry("headline", "fast" if arith else "parity", scheme="colloc4", n_freq=len(mine))
        flops_per_launch = evals_per_launch * ex["executed_flops_per_unit"]
        achieved = flops_per_launch / (ms / cnt * 1e-3) / 1e12
        ref_eq = evals_per_launch * FLOPS_PER_EVAL["colloc4"] / (ms / cnt * 1e-3) / 1e12
        kernel = f"{kind}<colloc4>"
        pipe = ("fp64-valu (no MFMA: 4x4 products in the reference's order with separate mul and add, DESIGN.md section 2)"
                if arith == 0 else
                "fp64-valu + fp64-mfma, one pipe and one peak on gfx950 (the step matrices in v_fma_f64, the ordered "
                "product in v_mfma_f64_4x4x4: DESIGN.md section 5)")
        # HBM bytes per launch from the same counter file ((2 x FETCH_SIZE + WRITE_SIZE) x 1024, the guide's gfx950 correction);
        # scaled by the batch size if this launch is not one of the profiled sizes
        traffic = ex["kernels"][ex["dominant_kernel"]]["hbm_bytes_per_call"]
        r = {"bound": "compute", "pipe": pipe,
             "kernel": kernel, "achieved": round(achieved, 3), "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
             "frac": round(achieved / PEAK_FP64_TFLOPS, 4), "traffic": traffic,
             "traffic_note": f"HBM bytes per launch, rocprofv3 FETCH_SIZE/WRITE_SIZE passes on {ex['n_freq']} frequencies per launch; "
                             "algorithmic bytes: coefficient planes once (2.7 MB) + 16 B per frequency",
             "avg_launch_ms": round(ms / cnt, 4), "launches_timed": cnt, "flops_per_launch": flops_per_launch,
             "executed_flops_per_eval": round(ex["executed_flops_per_unit"], 2), "executed_from": executed_source(ex),
             "profiled_batch_matches": ex["n_freq"] == len(mine), "breakdown": pipe_breakdown(ex, ms / cnt),
             "reference_equivalent": {"flops_per_eval": FLOPS_PER_EVAL["colloc4"], "achieved": round(ref_eq, 3), "unit": "TFLOP/s",
                                      "ratio_to_peak": round(ref_eq / PEAK_FP64_TFLOPS, 4),
                                      "what": "the reference formulation's un-fused flops for the same evaluations / time"},
             "frequencies_per_launch": len(mine), "note": NOTE[arith],
             "rocprof_row": "det_hex_kernel<colloc4,...> with 16 threads per frequency; in the FMA build, batches of >= 30 x "
                            "wave slots (~9e4) frequencies are ONE launch of det_mix_kernel<colloc4,...> (det_hex's 16-lane "
                            "groups and 4-lane column-owner groups side by side: DESIGN.md section 5) and a batch with a "
                            "remainder below a quarter wave round adds fast:seg_product + fast:chain_kernel for it"}
        if arith == 0:
            r["frac_of_unfused_ceiling"] = round(ref_eq / UNFUSED_TFLOPS, 4)
        return r

    def measure(arith, steps, warmup):
        h.set_option("arithmetic", arith)
        for _ in range(warmup):
            one_scan()
        h.stats(reset=True)
        h.kernel_times(reset=True)
        sync()
        t0 = time.perf_counter()
        marks = []
        for _ in range(steps):
            res = one_scan()
            marks.append(time.perf_counter())  # (a scan ends synchronised: its results are on the host)
        sync()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        step_ms = [round((b - a) * 1e3, 3) for a, b in zip([t0] + marks[:-1], marks)]
        st = h.stats()
        det_evals, refine, eigs, rounds, launches = sum_over_ranks(
            [st["det_evals"], st["refine_evals"], st["eigenvectors"], st["refine_rounds"], st["launches"]])
        kt = {k: {"ms_total": round(v[0], 3), "launches": v[1], "avg_ms": round(v[0] / v[1], 4)}
              for k, v in h.kernel_times(reset=True).items() if v[1]}
        return {"elapsed": elapsed, "modes": len(res), "det_evals": det_evals, "refine": refine, "eigs": eigs,
                "rounds_per_scan": rounds / steps / world, "launches_per_scan": launches / steps / world, "kernels": kt,
                "step_ms": step_ms}

    # The interpreter's full garbage collection walks every object `import torch` left behind (~40 ms) and its turn comes
    # after a number of allocations, i.e. inside some step or other (seen as one 52 ms step among 16 ms ones under `step_ms`,
    # in the first process of a fresh tree, whose byte-compilation shifts the count).  Collect now and move what exists to the
    # permanent generation: the collector stays on, its later passes see the benchmark's own few objects only.
    import gc
    gc.collect()
    gc.freeze()
    # Order: the two kernel-alone timings and the other build's leg first, the headline region last.
    main_arith = 1 if args.arithmetic == "fast" else 0
    other_steps = max(1, min(args.steps, 5))
    roof = {a: (roofline_of(a) if rank == 0 else None) for a in (0, 1)}  # (HIP-event timed: a host stall does not show)
    other = measure(1 - main_arith, other_steps, 1)
    res = measure(main_arith, args.steps, args.warmup)

    def phase_times(steps=5):
        """Where a step goes, per rank (host clock around each phase inside strong_scan_and_optimize).  N > 1: slice_scan =
        scan + refinement of the rank's slice (one library call), eigenvectors = this rank's modes, computed while the one
        all_gather is in flight, gather_refined = what is left to wait for (it contains the wait for the slowest rank);
        gather_found and refine are 0 in this protocol.  N = 1: no gather, and the step is the two-call form of the scan
        (storm_scan_brackets, storm_optimize_brackets), not the one-call form `value` is measured on.  Returns rank 0's
        figures and the maximum over the ranks, ms per step."""
        h.set_option("arithmetic", main_arith)
        one_scan(timings={})
        t = {}
        sync()
        for _ in range(steps):
            one_scan(timings=t)
        sync()
        mine_ms = [t.get(k, 0.0) / steps * 1e3 for k in D.PHASES]
        if world > 1:
            tt = torch.tensor(mine_ms, dtype=torch.float64, device=coll_device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            worst = tt.tolist()
        else:
            worst = mine_ms
        names = ["slice_scan", "gather_found", "refine", "eigenvectors", "gather_refined"]
        return {"steps": steps, "arithmetic": "fast" if main_arith else "parity", "unit": "ms per step",
                "frequencies_in_this_ranks_slice": int(len(mine)),
                "rank0": {k: round(v, 4) for k, v in zip(names, mine_ms)},
                "max_over_ranks": {k: round(v, 4) for k, v in zip(names, worst)},
                "sum_of_max": round(sum(worst), 4)}

    def leg(fn, *a):
        """A leg that is not `value`: an exception in it is reported in its place instead of costing the whole line."""
        try:
            return fn(*a)
        except Exception as e:  # noqa: BLE001 (reported, not hidden: the key then holds the error text)
            return {"error": f"{type(e).__name__}: {e}"}

    phases = phase_times() if world > 1 else leg(phase_times)  # (N > 1: every rank is inside its collectives, no lone catch)

    def multi_in_library(steps=5):
        """The same strong-scaled scan through storm_multi_* (include/storm_hip.h): ONE process, one solver and one host
        thread per device, gathers in host memory, no collective library -- the form the Rust host would call.  Run by rank
        0 over the N devices of the job while the other ranks wait (their processes hold no kernels then)."""
        if torch.cuda.device_count() < world:
            return {"skipped": f"{torch.cuda.device_count()} device(s) visible to rank 0, {world} needed"}
        devices = list(range(world)) if world > 1 else [local_rank]
        ms = host.MultiSolver(model, 2, 0, "colloc4", grid, devices)
        ms.hip.set_option("arithmetic", main_arith)

        def step():
            res, _, _ = ms.hip.scan_and_optimize(freqs, ulp, cap=8192, want_dets=False)
            if res:
                ms.hip.eigenvector(roots_of(res), out=vec_buf)
            return res

        step()
        ms.hip.stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
        dt = time.perf_counter() - t0
        st = ms.hip.stats()
        torch.cuda.set_device(local_rank)
        return {"devices": devices, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4), "modes_per_step": len(res),
                "value": st["det_evals"] * n_int / dt, "unit": "freq*interval evals/s",
                "bracket_indices_equal_headline": indices_of(res) == idx_f}

    # While rank 0 drives all N devices the other ranks wait on the HOST (a key in the rendezvous store): a barrier of the
    # GPU backend would keep a spinning kernel on every device the leg is about to time.
    multi = leg(multi_in_library) if rank == 0 else None
    if world > 1:
        try:
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("storm_multi_leg_done", "1")
            else:
                store.wait(["storm_multi_leg_done"])
        except Exception:  # noqa: BLE001 (no store to wait on: the backend's own barrier, kernels and all)
            pass
        dist.barrier()
    h.set_option("arithmetic", 0)

    def eigen_roofline():
        """The eigenvector kernels of the headline's modes against their HBM roofline (per interval and
        mode 128 B step matrix in, 176 B factors out and in again, 32 B eigenvector out)."""
        roots = roots_of(found["parity"])
        if not roots.size:
            return None
        h.eigenvector(roots)
        h.kernel_times(reset=True)
        reps = 3
        for _ in range(reps):
            h.eigenvector(roots)
        kt = h.kernel_times(reset=True)
        ms = kt["eigen"][0] / reps
        per_unit = 128.0 + 2 * 176.0 + 32.0
        gbs = roots.size * n_int * per_unit / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "eigen_forward + eigen_backsub", "modes": int(roots.size), "ms": round(ms, 3),
                "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                "bytes_per_interval_and_mode": per_unit,
                "note": "latency-bound, not bandwidth-bound, at a scan's mode count: one serial pivoted elimination per mode"}

    out = None
    if rank == 0:
        name_main, name_other = ("fast", "parity") if main_arith else ("parity", "fast")
        value = res["det_evals"] * n_int / res["elapsed"]
        out = {
            "metric": "scan throughput: freq x gridpoint evals/sec (whole job: scan + bracket refinement + eigenvectors)",
            "value": value, "unit": "freq*interval evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": res["elapsed"] / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (n = 3 polytrope integrated from the Lane-Emden equation, 10^4 points; deterministic, no RNG)",
            "config": {"workload": "BASELINE config 3: synthetic 10^4-point model (n=3 polytrope, Gamma_1=5/3), ell=2, m=0, "
                                   "Colloc4, scan linspace(1,25,10^5) at precision 1e-8 + eigenvector per mode "
                                   "(= StormState::scan, src/main.rs:793-838)",
                       "n_freq_total": N_FREQ, "n_int": n_int, "scheme": "colloc4", "ell": 2,
                       "parallelism": f"one 10^5-point grid cut into {world} contiguous slice(s), brackets refined where found, one all_gather"},
            "arithmetic": name_main,
            "modes_per_s": res["eigs"] / res["elapsed"], "modes_per_step": res["modes"],
            "det_evals": res["det_evals"], "refine_evals": res["refine"],
            "refine_rounds_per_scan": res["rounds_per_scan"], "launches_per_scan": res["launches_per_scan"],
            "fast_vs_parity": identity, "kernels": res["kernels"], "step_ms": res["step_ms"],
            "ms_per_step_median": sorted(res["step_ms"])[len(res["step_ms"]) // 2],
            "roofline": roof[main_arith],
            "phases": phases, "multi_in_library": multi,
            f"{name_other}_arithmetic": {
                "value": other["det_evals"] * n_int / other["elapsed"], "unit": "freq*interval evals/s",
                "steps": other_steps, "ms_per_step": other["elapsed"] / other_steps * 1e3,
                "modes_per_s": other["eigs"] / other["elapsed"], "modes_per_step": other["modes"],
                "kernels": other["kernels"], "step_ms": other["step_ms"], "roofline": roof[1 - main_arith]},
            "eigenvector_roofline": eigen_roofline(),
        }
    if rank == 0 and world == 1 and args.legs == "all":
        out["secondary"] = secondary_legs(host, ffi, D, torch, ulp, local_rank)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_budget)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()  # rank 0 ran the extra legs: leave together
        dist.destroy_process_group()


def secondary_legs(host, ffi, D, torch, ulp, local_rank):
    """Outside `value`: the other BASELINE configs and the SURVEY 8(f) rows, each measured briefly (N = 1 only)."""
    legs = {}
    npz = os.path.join(ROOT, "tests", "golden", "gsm_tams.npz")

    # ---- Polytrope0 on linspace(0, 1, 10^4) (SURVEY 8(d) config 3's second model), Colloc4, 10^5 frequencies
    m0 = host.Model.Polytrope0(5.0 / 3.0)
    g0 = host.linspace(0.0, 1.0, 10000)
    e0 = host.ErasedSolver(m0, 2, 0, "colloc4", g0)
    h0 = ffi.HipSolver.borrow(e0.raw_handle(), owner=e0)
    f0 = host.linspace(1.0, 25.0, N_FREQ)
    for arith in (0, 1):
        h0.set_option("arithmetic", arith)
        h0.scan_and_optimize(f0, ulp, cap=8192)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            r0, _, _ = h0.scan_and_optimize(f0, ulp, cap=8192)
            h0.eigenvector(np.array([b["root"] for b in r0]))
        dt = (time.perf_counter() - t0) / reps
        legs.setdefault("polytrope0", {})["fast" if arith else "parity"] = {
            "ms_per_scan": round(dt * 1e3, 3), "modes": len(r0),
            "evals_per_s": (N_FREQ + sum(b["evals"] for b in r0)) * e0.n_int / dt}
    legs["polytrope0"]["workload"] = "Polytrope0{5/3} on linspace(0,1,10^4), ell=2, Colloc4, linspace(1,25,10^5), 1e-8, + eigenvectors"
    # ---- the same scan bracketing on the scaled discriminant (option scaled_scan: mantissa x 2^exponent), FMA build
    h0.set_option("arithmetic", 1)
    raw0, _, _ = h0.scan_and_optimize(f0, ulp, cap=8192, want_dets=False)
    t0 = time.perf_counter()
    h0.scan_and_optimize(f0, ulp, cap=8192, want_dets=False)
    t_raw = time.perf_counter() - t0
    h0.set_option("scaled_scan", 1)
    sc0, _, _ = h0.scan_and_optimize(f0, ulp, cap=8192, want_dets=False)
    t0 = time.perf_counter()
    h0.scan_and_optimize(f0, ulp, cap=8192, want_dets=False)
    t_sc = time.perf_counter() - t0
    h0.set_option("scaled_scan", 0)
    same = [b["index"] for b in raw0] == [b["index"] for b in sc0]
    legs["scaled_scan"] = {"workload": "the Polytrope0 scan above without eigenvectors, FMA build",
                           "raw_ms_per_scan": round(t_raw * 1e3, 3), "scaled_ms_per_scan": round(t_sc * 1e3, 3),
                           "bracket_indices_equal": same,
                           "max_root_rel_diff": (max(abs(a["root"] - b["root"]) / abs(a["root"]) for a, b in zip(raw0, sc0))
                                                 if same and raw0 else None)}
    del h0, e0

    # ---- BASELINE config 2: tams, ell = 1..3 x {Magnus6, Colloc8}, 10^4 frequencies, six handles on six threads
    model = host.Model.from_gsm_npz(npz)
    grid = model.field("r_coord")
    cases = [(ell, scheme) for scheme in ("colloc8", "magnus6") for ell in (3, 2, 1)]
    freqs = host.linspace(1.0, 25.0, 10000)
    solvers = []
    for ell, scheme in cases:
        s = host.ErasedSolver(model, ell, 0, scheme, grid)
        s.set_option("profile", 1)
        solvers.append(s)
    views = [ffi.HipSolver.borrow(s.raw_handle(), owner=s) for s in solvers]
    n_int = solvers[0].n_int
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=len(solvers))

    def chain(k, steps):
        torch.cuda.set_device(local_rank)
        modes = 0
        for _ in range(steps):
            res, _, _ = views[k].scan_and_optimize(freqs, ulp, cap=4096)
            if res:
                views[k].eigenvector(np.array([b["root"] for b in res]))
            modes = len(res)
        return modes

    cfg2 = {"workload": "tams (6199 intervals), ell=1..3, m=0, Magnus6 + Colloc8, linspace(1,25,10^4), 1e-8, + eigenvectors; "
                        "six scans per step on six handles/streams, each handle's steps back to back"}
    for arith in (0, 1):
        for v in views:
            v.set_option("arithmetic", arith)
        list(pool.map(lambda k: chain(k, 1), range(len(solvers))))
        for v in views:
            v.stats(reset=True)
        steps = 4
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        modes = list(pool.map(lambda k: chain(k, steps), range(len(solvers))))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        evals = sum(v.stats()["det_evals"] for v in views)
        # the dominant kernel alone, one launch
        vd = views[0]
        vd.set_option("hex_rounds", 0)
        vd.det(freqs)
        vd.kernel_times(reset=True)
        for _ in range(3):
            vd.det(freqs)
        ms, cnt = vd.kernel_times(reset=True)["det_hex"]
        vd.set_option("hex_rounds", 1)
        ex = executed_entry("config2", "fast" if arith else "parity", scheme="colloc8", n_freq=freqs.size)
        ach = freqs.size * n_int * ex["executed_flops_per_unit"] / (ms / cnt * 1e-3) / 1e12
        ref_eq = freqs.size * n_int * FLOPS_PER_EVAL["colloc8"] / (ms / cnt * 1e-3) / 1e12
        cfg2["fast" if arith else "parity"] = {
            "evals_per_s": evals * n_int / dt, "ms_per_step": round(dt / steps * 1e3, 3), "modes_per_step": sum(modes),
            "roofline": {"bound": "compute", "pipe": "fp64-valu", "kernel": "det_hex<colloc8>", "achieved": round(ach, 3),
                         "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP64_TFLOPS, 4),
                         "avg_launch_ms": round(ms / cnt, 4),
                         "traffic": ex["kernels"][ex["dominant_kernel"]]["hbm_bytes_per_call"],
                         "executed_flops_per_eval": round(ex["executed_flops_per_unit"], 2), "executed_from": executed_source(ex),
                         "reference_equivalent": {"flops_per_eval": FLOPS_PER_EVAL["colloc8"], "achieved": round(ref_eq, 3),
                                                  "unit": "TFLOP/s", "ratio_to_peak": round(ref_eq / PEAK_FP64_TFLOPS, 4)}}}
    for v in views:
        v.set_option("arithmetic", 0)
    legs["config2"] = cfg2

    # ---- eigenvectors + device post-processing, only the mode ids cross PCIe
    sd, vd = solvers[0], views[0]
    roots = np.array([b["root"] for b in vd.scan_and_optimize(freqs, ulp, cap=4096)[0]])
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
        k_ms = kt["postprocess"][0] / reps
        alg = roots.size * grid.size * 80.0   # terms: 32 B vector + 8 B out; modes: 8 B term + 32 B vector
        legs["post_processing"] = {
            "modes": int(roots.size), "n_pts": int(grid.size), "ms_eigenvectors_plus_ids": round(wall * 1e3, 3),
            "postprocess_kernels_ms": round(k_ms, 4), "eigen_kernels_ms": round(kt["eigen"][0] / reps, 3),
            "algorithmic_GB_per_s": round(alg / (k_ms * 1e-3) / 1e9, 1) if k_ms else None, "hbm_peak_GB_per_s": HBM_PEAK_GBS,
            "radial_orders": [int(ids[0, 2]), int(ids[-1, 2])]}
        # ---- the coupling integrals of the deformed star on M = 512 modes (the found ones, repeated):
        # 148 un-fused flops per (left, right, grid point), FP64 VALU like the discriminant kernels
        got = pp.modes(sd, roots, want_fields=True)
        M = 512
        sel = np.arange(M) % roots.size
        c_f, c_fl = roots[sel], np.ascontiguousarray(got["fields"][sel])
        c_ell = np.full(M, sd.ell if hasattr(sd, "ell") else 1, dtype=np.uint64)
        x = grid
        metric = {"alpha": 1e-3 * x ** 2, "dalpha": 2e-3 * x ** 2, "ddalpha": 4e-3 * x ** 2,
                  "beta": -1.2e-3 * x ** 2, "dbeta": -2.4e-3 * x ** 2, "ddbeta": -4.8e-3 * x ** 2}
        on_grid = model.eval(grid)
        raw = ffi.HipPostproc({k: on_grid.field(k) for k in ffi.MODEL_FIELDS})
        cp = {"modes": M, "n_pts": int(grid.size), "flops_per_pair_and_point": 148,
              "kernel": "coupling_kernel<16> (+ its two prepasses; in the FMA build + coupling_matrix_kernel, the tiles of like "
                        "degrees on v_mfma_f64_16x16x4: HIP events around all launches)"}
        for arith in (0, 1):
            raw.set_option("arithmetic", arith)
            raw.coupling_matrices(c_ell, c_f, c_fl, 0, metric)
            raw.coupling_matrices(c_ell, c_f, c_fl, 0, metric)
            k_ms = raw.last_kernel_ms()
            ex = executed_entry("coupling", "fast" if arith else "parity", modes=M)
            terms = M * M * (grid.size - 1)
            ach = terms * ex["executed_flops_per_unit"] / (k_ms * 1e-3) / 1e12
            ref_eq = terms * 148.0 / (k_ms * 1e-3) / 1e12
            cp["fast" if arith else "parity"] = {
                "bound": "compute", "pipe": "fp64-valu + fp64-mfma (one pipe)" if arith else "fp64-valu",
                "avg_launch_ms": round(k_ms, 3), "achieved": round(ach, 3), "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_FP64_TFLOPS, 4), "traffic": ex["hbm_bytes_per_call"],
                "executed_flops_per_pair_and_point": round(ex["executed_flops_per_unit"], 2),
                "executed_from": {"file": "profiles/r03_executed_flops.json", "commit": EXECUTED.get("commit"),
                                  "label": ex["label"], "kernels": ex["priced_kernels"]},
                "reference_equivalent": {"flops_per_pair_and_point": 148.0, "achieved": round(ref_eq, 3), "unit": "TFLOP/s",
                                         "ratio_to_peak": round(ref_eq / PEAK_FP64_TFLOPS, 4)}}
        cp["note"] = ("achieved = EXECUTED flops (rocprofv3 FP64 instruction counters over all launches of the call) / HIP-event time "
                      "around them; reference_equivalent = the reference formulation's 148 flops per (pair, point) / time.  The FMA "
                      "build's matrix walk executes 21 v_mfma_f64_16x16x4 per four points and tile = 42 flops per (pair, point) "
                      "plus the weighted left operands, and is bound by staging the fields from L2 / HBM, not by the pipe: "
                      "DESIGN.md section 10")
        legs["coupling_integrals"] = cp
        del raw, c_fl, got
    del views, solvers
    pool.shutdown()

    # ---- a rotation sweep of many small scans in lock-step (storm_sweep)
    combos = [(0, 0), (2, 0), (4, 0), (1, 1), (1, -1), (2, 1), (2, -1), (3, 0)]
    rmodels = []
    for k in range(8):
        mdl = host.Model.from_gsm_npz(npz)
        mdl.set_rotation_constant(0.01 + 0.002 * k)
        rmodels.append(mdl)
    systems = [(rmodels[i], ell, m) for i in range(len(rmodels)) for ell, m in combos]
    sgrid = host.linspace(1.0, 25.0, 300)

    def sweep_once():
        t0 = time.perf_counter()
        sw = host.SolverSweep(systems, "colloc4", grid)
        t1 = time.perf_counter()
        r = sw.scan_and_optimize([sgrid] * len(systems), host.Precision.Relative(1e-8))
        t2 = time.perf_counter()
        posts = [host.PostProcessor(mdl.eval(grid)) for mdl in rmodels]
        got = sw.modes([posts[k // len(combos)] for k in range(len(systems))],
                       [np.array([b["root"] for b in br]) for br, _ in r])
        t3 = time.perf_counter()
        return sum(len(x["ids"]) for x in got), (t1 - t0, t2 - t1, t3 - t2), sw.stats()

    sweep_once()
    n_modes, (t_new, t_scan, t_modes), st = sweep_once()
    total = t_new + t_scan + t_modes
    legs["rotation_sweep"] = {
        "systems": len(systems), "frequencies_per_scan": int(sgrid.size), "scheme": "colloc4", "modes": int(n_modes),
        "ms_per_scan": round(total / len(systems) * 1e3, 4), "modes_per_s": round(n_modes / total, 1),
        "ms_create_scan_modes": [round(t_new * 1e3, 2), round(t_scan * 1e3, 2), round(t_modes * 1e3, 2)],
        "refine_rounds": st["refine_rounds"], "launches": st["launches"]}

    # ---- the reference's batch-input language end to end (tools/storm_batch, a child process)
    tool = os.path.join(ROOT, "tools", "storm_batch.bin")
    if os.path.exists(tool):
        import re
        import subprocess
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            mb, bt = os.path.join(tmp, "tams_model.bin"), os.path.join(tmp, "batch.txt")
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_model_bin.py"), "tams", mb], check=True,
                           stdout=subprocess.DEVNULL)
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_sweep_batch.py"), bt, "64"], check=True)
            for arith, name in ((0, "parity"), (1, "fast")):
                r = subprocess.run([tool, mb, bt, os.path.join(tmp, "out.txt"), "--group", "32", "--summary", "--device",
                                    str(local_rank), "--arithmetic", str(arith)], capture_output=True, text=True, timeout=300)
                mt = re.search(r"(\d+) blocks, (\d+) scans, (\d+) modes in ([0-9.]+) ms", r.stderr)
                if r.returncode == 0 and mt:
                    nb, ns, nm, ms = int(mt.group(1)), int(mt.group(2)), int(mt.group(3)), float(mt.group(4))
                    rec = {"rotation_rates": nb, "scans": ns, "modes": nm, "wall_ms": ms,
                           "ms_per_rotation_rate": round(ms / nb, 3), "modes_per_s": round(nm / ms * 1e3, 1)}
                    if arith == 0:
                        legs["batch_pipeline"] = rec
                    else:
                        legs.setdefault("batch_pipeline", {})["fast"] = rec
    return legs


if __name__ == "__main__":
    main()
