#!/bin/bash
# End-of-round run: GPU tests, default bench, all-scheme launch times (both builds), per-rank step of the strong-scaled scan.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/t_full.log 2>&1 || { tail -30 $O/t_full.log; exit 1; }
tail -2 $O/t_full.log
timeout -k 10 400 python bench.py > $O/bench_default.log 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
tail -c 300 $O/bench_default.log; echo
ALL=colloc2,colloc4,colloc6,colloc8,magnus2,magnus4,magnus6,magnus8
for n in 10000 100000; do
  timeout -k 10 300 python scripts/time_scheme.py $ALL $n 0 2>&1 | grep "arith=" || exit 1
done > $O/r2_all_schemes_time_scheme.txt
cat $O/r2_all_schemes_time_scheme.txt
timeout -k 10 300 python scripts/strong_slice_timing.py > $O/strong_slice.txt 2>&1 || { tail $O/strong_slice.txt; exit 1; }
cat $O/strong_slice.txt
