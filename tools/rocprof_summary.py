#!/usr/bin/env python3
"""Condense rocprofv3 output (rocpd sqlite db or *_kernel_trace.csv / *_counter_collection.csv) into a
small text summary for profiles/.  usage: rocprof_summary.py <dir> [> profiles/xxx.txt]"""
import csv
import glob
import os
import re
import sqlite3
import sys
from collections import defaultdict


NAMES = ["colloc2", "colloc4", "colloc6", "colloc8", "magnus2", "magnus4", "magnus6", "magnus8"]


def short(name):
    name = re.sub(r"\.kd$", "", name)
    m = re.match(r"(?:void )?storm(_fast)?::(?:\(anonymous namespace\)::)?(\w+)(?:<([^>]*)>)?\(", name)
    if m:  # demangled
        targs = [a.strip() for a in (m.group(3) or "").split(",") if a.strip()]
        if targs and targs[0].isdigit() and not m.group(2).startswith(("eigen", "coupling")):  # (eigen kernels: an ell class, not a scheme)
            targs[0] = NAMES[int(targs[0])]
        targs = [{"true": "1", "false": "0"}.get(a, a) for a in targs]
        return ("fast:" if m.group(1) else "") + m.group(2) + (f"<{','.join(targs)}>" if targs else "")
    m = re.match(r"_ZN5storm(\d+)(\w+)", name)
    if m:
        n = int(m.group(1))
        rest = m.group(2)
        base = rest[:n]
        targs = re.match(r"I(?:Li(\d+)E)?((?:Lb[01]E)*)E", rest[n:])
        if targs:
            scheme = targs.group(1)
            flags = re.findall(r"Lb([01])E", targs.group(2))
            names = ["colloc2", "colloc4", "colloc6", "colloc8", "magnus2", "magnus4", "magnus6", "magnus8"]
            parts = ([scheme if base.startswith("eigen") else names[int(scheme)]] if scheme is not None else []) + flags
            return f"{base}<{','.join(parts)}>"
        return base
    return name[:60]


def from_db(path):
    db = sqlite3.connect(path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute(f"select s.kernel_name, d.grid_size_x, d.grid_size_y, d.start, d.end, s.arch_vgpr_count, "
                     f"s.accum_vgpr_count, s.sgpr_count from {disp} d join {sym} s on d.kernel_id = s.id").fetchall()
    return [(short(n), gx * max(gy, 1), (e - s) / 1e3, f"v{v}+a{a} s{sg}") for n, gx, gy, s, e, v, a, sg in rows]


def from_csv(d):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            grid = int(r.get("Grid_Size_X", 0) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1))
            out.append((short(r["Kernel_Name"]), grid, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                        f"v{r.get('VGPR_Count', '?')}+a{r.get('Accum_VGPR_Count', '?')} s{r.get('SGPR_Count', '?')}"))
    return out


def counters(d):
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    d = sys.argv[1]
    rows = []
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        rows += from_db(db)
    rows += from_csv(d)
    if rows:
        groups = defaultdict(list)
        for name, grid, us, regs in rows:
            groups[(name, grid, regs)].append(us)
        tot = sum(sum(v) for v in groups.values())
        print(f"# kernel dispatches: {len(rows)}, total device time {tot / 1e3:.3f} ms")
        print(f"{'kernel':48s} {'grid(threads)':>14s} {'regs':>14s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
        for (name, grid, regs), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            print(f"{name:48s} {grid:14d} {regs:>14s} {len(v):6d} {sum(v) / 1e3:10.3f} {sum(v) / len(v):10.1f} {min(v):10.1f} {max(v):10.1f} {100 * sum(v) / tot:6.1f}")
    cnt = counters(d)
    if cnt:
        print("# PMC counters (per dispatch average)")
        for k, cs in cnt.items():
            for cname, vals in cs.items():
                print(f"{k:48s} {cname:24s} n={len(vals):4d} avg={sum(vals) / len(vals):.6g}")


if __name__ == "__main__":
    main()
