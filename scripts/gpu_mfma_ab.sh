#!/bin/bash
# MFMA phase B of det_hex (FMA build): fast-mode tests, then the per-scheme launch times (automatic choice and 16 lanes forced).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread > $O/t_mfma.log 2>&1; rc=$?
if [ $rc -gt 1 ]; then tail -30 $O/t_mfma.log; exit $rc; fi
grep -E "^FAILED|^ERROR" $O/t_mfma.log
tail -2 $O/t_mfma.log
ALL=colloc2,colloc4,colloc6,colloc8,magnus2,magnus4,magnus6,magnus8
for n in 10000 100000; do
  for lanes in 0 16; do
    timeout -k 10 300 python scripts/time_scheme.py $ALL $n $lanes 2>&1 | grep "arith=1" || exit 1
  done
done > $O/mfma_time_scheme.txt
cat $O/mfma_time_scheme.txt
timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --legs headline > $O/bench_mfma.log 2>&1 || { tail $O/bench_mfma.log; exit 1; }
python - <<'P'
import json
d=json.loads(open('gpurun_out/bench_mfma.log').read().strip().splitlines()[-1])
print("headline ms", d['ms_per_step'], "launch", d['roofline']['avg_launch_ms'], d['roofline']['kernel'], d['fast_vs_parity'])
P
