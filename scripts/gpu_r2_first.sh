#!/bin/bash
# Round-2 first GPU check: new full-size tests first (they are the riskiest), then the whole GPU suite, then the bench.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R || exit 1
timeout -k 10 900 python -u -m pytest tests/test_hip_full_size.py -m gpu -x -v --timeout 400 --timeout-method thread > $O/r2_fullsize.log 2>&1 || { tail -40 $O/r2_fullsize.log; exit 1; }
tail -3 $O/r2_fullsize.log
timeout -k 10 400 python -u bench.py > $O/r2_bench.log 2>&1 || { tail -30 $O/r2_bench.log; exit 1; }
tail -1 $O/r2_bench.log | cut -c1-1500
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread --deselect tests/test_hip_full_size.py > $O/r2_tests.log 2>&1 || { tail -30 $O/r2_tests.log; exit 1; }
tail -2 $O/r2_tests.log
echo done
