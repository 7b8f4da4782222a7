#!/usr/bin/env python3
"""Device timeline of the LAST sentinel-delimited segment of a rocprofv3 --kernel-trace [--memory-copy-trace] csv run:
one line per kernel / copy with its start relative to the segment's first event, duration and the gap before it.
usage: timeline_summary.py <dir>"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocprof_summary import short
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], f"q{r['Queue_Id']}"))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), ""))
ev.sort()
marks = [i for i, e in enumerate(ev) if "at::native" in e[2]]
seg = ev[marks[-2] + 1:marks[-1]]
t0 = seg[0][0]
prev_end = t0
print(f"{'start_us':>10s} {'dur_us':>10s} {'gap_us':>8s}  what")
for s, e, name, q in seg:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:10.1f} {(s - prev_end) / 1e3:8.1f}  {short(name) if not name.startswith('COPY') else name} {q}")
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {sum(e - s for s, e, _, _ in seg) / 1e3:.1f} us")
