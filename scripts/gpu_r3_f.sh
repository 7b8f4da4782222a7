#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python3 -u -m pytest tests/test_hip_coupling.py tests/test_hip_batch.py -x -q --timeout 300 --timeout-method thread > $O/r3_f_tests.log 2>&1 || { tail -60 $O/r3_f_tests.log; exit 1; }
tail -3 $O/r3_f_tests.log
timeout -k 10 600 python3 scripts/coupling_bench.py 80 240 512 1024 2048 > $O/r3_coupling_bench.txt 2>&1 || { tail -20 $O/r3_coupling_bench.txt; exit 1; }
grep "arith=1" $O/r3_coupling_bench.txt
