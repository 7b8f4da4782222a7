#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python3 -u -m pytest tests/test_hip_coupling.py tests/test_hip_batch.py tests/test_hip_sweep.py -x -q --timeout 300 --timeout-method thread > $O/r3_e_tests.log 2>&1 || { tail -60 $O/r3_e_tests.log; exit 1; }
tail -3 $O/r3_e_tests.log
timeout -k 10 600 python3 scripts/coupling_bench.py 32 80 240 512 1024 > $O/r3_coupling_bench.txt 2>&1 || { tail -20 $O/r3_coupling_bench.txt; exit 1; }
cat $O/r3_coupling_bench.txt
python3 tools/make_model_bin.py tams /tmp/tams_model.bin > /dev/null && python3 tools/make_sweep_batch.py /tmp/batch64.txt 64 && for A in 0 1 1; do timeout -k 10 300 tools/storm_batch.bin /tmp/tams_model.bin /tmp/batch64.txt /tmp/out.txt --group 32 --summary --arithmetic $A 2>&1 | grep "blocks,"; done
