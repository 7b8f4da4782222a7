#!/usr/bin/env python3
"""Executed FP64 work per launch from rocprofv3 passes over scripts/profile_executed.py -> profiles/r03_executed_flops.json
(the file bench.py prices its roofline entries on).

usage: executed_flops.py <dir> <out.json> [commit]
<dir> holds, per group g of scripts/gpu_r3_executed.sh: g_manifest.json (the calls, in order), g_trace/ (--kernel-trace:
durations), g_flops/ (--pmc SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS,MFMA}_F64 ...), g_fetch/ (--pmc FETCH_SIZE), g_write/ (WRITE_SIZE).

Flops per wave-instruction (64 lanes; the SQ counters count wave-instructions whatever the exec mask, which is what occupies the
pipe): v_fma_f64 128, v_mul_f64 / v_add_f64 64, transcendental (v_rcp_f64, v_rsq_f64, v_sqrt_f64) 64, v_mfma_f64_4x4x4 512
(four 4x4x4 blocks x 2), v_mfma_f64_16x16x4 2048.  The division helpers v_div_scale / v_div_fmas / v_div_fixup are in none of
these counters (the parity build's IEEE divisions are therefore under-counted by ~3 of their ~11 instructions).
HBM bytes: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE tallies 128-B requests
at 64 B: calibrated on 16-byte-per-lane streams, so for these kernels' 8-byte and LDS-DMA reads it is an upper bound)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocprof_summary import short  # noqa: E402

PEAK_TFLOPS = 78.6
WEIGHT = {"SQ_INSTS_VALU_FMA_F64": 128.0, "SQ_INSTS_VALU_MUL_F64": 64.0, "SQ_INSTS_VALU_ADD_F64": 64.0,
          "SQ_INSTS_VALU_TRANS_F64": 64.0}
# reference formulation (oracle op count), per (frequency, interval) evaluation / per (pair, point)
REF_FLOPS = {"colloc2": 610.1, "colloc4": 1396.1, "colloc6": 2694.1, "colloc8": 5048.1,
             "magnus2": 1486.0, "magnus4": 1848.0, "magnus6": 2562.0, "magnus8": 3900.1, "coupling": 148.0}


def is_sentinel(name):
    return "at::native" in name or "at_cuda" in name


def dispatches(d, counters):
    """[(dispatch id, kernel name, grid, {counter: value}, duration us)] in launch order"""
    rows = {}
    pat = "*counter_collection.csv" if counters else "*kernel_trace.csv"
    # (one process per pass; if the directory holds the files of several runs -- gpurun merges into what is there -- the newest)
    files = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            e = rows.setdefault(k, [k, r["Kernel_Name"], 0, {}, 0.0, 0, 0])
            if counters:
                e[2] = int(r["Grid_Size"])
                e[3][r["Counter_Name"]] = e[3].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            else:
                e[2] = int(r["Grid_Size_X"]) * max(1, int(r["Grid_Size_Y"])) * max(1, int(r["Grid_Size_Z"]))
                e[4] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                e[5], e[6] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    return [rows[k] for k in sorted(rows)]


def segments(rows, n_calls, what):
    """cut the dispatch stream at the sentinels: segment i+1 = call i of the manifest"""
    segs, cur = [], []
    for r in rows:
        if is_sentinel(r[1]):
            segs.append(cur)
            cur = []
        else:
            cur.append(r)
    # (the sentinel tensor's own creation is one more element-wise dispatch, ahead of everything)
    if len(segs) not in (n_calls + 1, n_calls + 2):
        raise SystemExit(f"{what}: {len(segs) - 1} sentinel-delimited calls in the profile, {n_calls} in the manifest")
    return segs[-n_calls:]


def mfma_flops(name):
    return 2048.0 if "coupling" in name else 512.0   # v_mfma_f64_16x16x4 (coupling_matrix_kernel) / v_mfma_f64_4x4x4 (det_*)


def main():
    d, out_path = sys.argv[1], sys.argv[2]
    commit = sys.argv[3] if len(sys.argv) > 3 else None
    entries = []
    for mf in sorted(glob.glob(os.path.join(d, "*_manifest.json"))):
        g = os.path.basename(mf)[:-len("_manifest.json")]
        manifest = json.load(open(mf))
        n = len(manifest)
        trace = segments(dispatches(os.path.join(d, g + "_trace"), False), n, g + "_trace")
        flops = segments(dispatches(os.path.join(d, g + "_flops"), True), n, g + "_flops")
        fetch = segments(dispatches(os.path.join(d, g + "_fetch"), True), n, g + "_fetch")
        write = segments(dispatches(os.path.join(d, g + "_write"), True), n, g + "_write")
        # group the timed calls of one item
        items = defaultdict(list)
        for i, m in enumerate(manifest):
            if not m["warm"]:
                key = tuple(sorted((k, v) for k, v in m.items() if k != "warm"))
                items[key].append(i)
        for key, calls in items.items():
            m = dict(key)
            per_kernel = defaultdict(lambda: {"us": [], "cnt": defaultdict(list), "grid": 0})
            spans = []   # per call: first start to last end over the coupling launches (two walks run side by side)
            for i in calls:
                cp = [r for r in trace[i] if "coupling" in r[1]]
                if cp:
                    spans.append((max(r[6] for r in cp) - min(r[5] for r in cp)) / 1e3)
                for _, name, grid, _, us, _, _ in trace[i]:
                    k = per_kernel[short(name)]
                    k["us"].append(us); k["grid"] = grid
                for seg in (flops[i], fetch[i], write[i]):
                    for _, name, grid, cnt, _, _, _ in seg:
                        for c, v in cnt.items():
                            per_kernel[short(name)]["cnt"][c].append(v)
            n_calls = len(calls)
            kernels = {}
            for name, k in per_kernel.items():
                if name.startswith("__amd_rocclr"):
                    continue
                c = {cn: sum(v) / n_calls for cn, v in k["cnt"].items()}   # per call
                ex = sum(WEIGHT[cn] * c.get(cn, 0.0) for cn in WEIGHT) + mfma_flops(name) * c.get("SQ_INSTS_VALU_MFMA_F64", 0.0)
                kernels[name] = {"grid_threads": k["grid"], "launches_per_call": len(k["us"]) / n_calls,
                                 "us_per_call": sum(k["us"]) / n_calls, "counters_per_call": c, "executed_flops_per_call": ex,
                                 "hbm_bytes_per_call": (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0}
            dom = max(kernels, key=lambda kn: kernels[kn]["us_per_call"])
            if m["family"] == "det":
                units = m["n_freq"] * m["n_int"]
                ref = REF_FLOPS[m["scheme"]]
                sel = [dom]                                            # the roofline entry is the dominant kernel's
            else:
                units = m["modes"] * m["modes"] * m["n_int"]
                ref = REF_FLOPS["coupling"]
                sel = [kn for kn in kernels if "coupling" in kn]       # all launches of the call (two walks side by side)
            ex = sum(kernels[kn]["executed_flops_per_call"] for kn in sel)
            us = sum(kernels[kn]["us_per_call"] for kn in sel) if m["family"] == "det" else sum(spans) / len(spans)
            hbm = sum(kernels[kn]["hbm_bytes_per_call"] for kn in sel)
            e = dict(m)
            e.update({"dominant_kernel": dom, "priced_kernels": sel, "units_per_call": units,
                      "unit": "(frequency, interval) evaluations" if m["family"] == "det" else "(pair, point) terms",
                      "kernel_us": us, "executed_flops_per_unit": ex / units, "executed_flops_per_call": ex,
                      "executed_tflops": ex / (us * 1e-6) / 1e12, "frac_of_peak": ex / (us * 1e-6) / 1e12 / PEAK_TFLOPS,
                      "reference_flops_per_unit": ref, "reference_equivalent_tflops": ref * units / (us * 1e-6) / 1e12,
                      "hbm_bytes_per_call": hbm, "kernels": kernels})
            entries.append(e)
    out = {"what": "executed FP64 flops and HBM bytes per launch from rocprofv3 counter passes (scripts/gpu_r3_executed.sh, "
                   "tools/executed_flops.py); bench.py prices roofline.frac on executed_flops_per_unit",
           "commit": commit, "peak_tflops": PEAK_TFLOPS,
           "flops_per_wave_instruction": dict(WEIGHT, SQ_INSTS_VALU_MFMA_F64_4x4x4=512.0, SQ_INSTS_VALU_MFMA_F64_16x16x4=2048.0),
           "entries": entries}
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"{'group':9s} {'item':34s} {'build':6s} {'dominant kernel':36s} {'us':>9s} {'exec/unit':>9s} {'ref/unit':>8s} {'TFLOP/s':>8s} {'frac':>6s} {'ref-eq':>7s} {'HBM MB':>8s}")
    for e in entries:
        item = (e.get("scheme", "") + f" n={e['n_freq']}") if e["family"] == "det" else f"M={e['modes']}"
        print(f"{e['group']:9s} {item:34s} {e['arithmetic']:6s} {e['dominant_kernel'][:36]:36s} {e['kernel_us']:9.1f} "
              f"{e['executed_flops_per_unit']:9.1f} {e['reference_flops_per_unit']:8.1f} {e['executed_tflops']:8.2f} "
              f"{e['frac_of_peak']:6.3f} {e['reference_equivalent_tflops']:7.2f} {e['hbm_bytes_per_call'] / 1e6:8.1f}")


if __name__ == "__main__":
    main()
