#!/bin/bash
# Sizes of the hipMalloc / hipHostMalloc calls of one 1001-rate batch run (rocprofv3 runtime trace).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/make_sweep_batch.py /tmp/sweep1001.txt 1001 > /dev/null || exit 1
rm -rf /tmp/mtrace
timeout -k 10 280 rocprofv3 --hip-runtime-trace --output-format csv -d /tmp/mtrace -- $R/tools/storm_batch.bin $R/tools/tams_model.bin /tmp/sweep1001.txt /tmp/sweep1001.out --group 32 --summary --arithmetic 1 > /tmp/mtrace.log 2>&1 || { tail -5 /tmp/mtrace.log; exit 1; }
grep 'blocks,' /tmp/mtrace.log | cut -c1-110
F=$(ls /tmp/mtrace/*/*hip_api_trace.csv | head -1)
head -1 $F
grep -m3 "hipMalloc" $F
grep -c "hipMalloc\"" $F
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
c = collections.Counter()
for r in rows:
    if r.get("Function") in ("hipMalloc", "hipHostMalloc"):
        c[(r["Function"], r.get("Thread_Id"))] += 1
print(c.most_common(12))
PY
