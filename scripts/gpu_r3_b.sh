#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python3 -u -m pytest tests/test_hip_full_size.py tests/test_hip_fast_mode.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread > $O/r3_b_tests.log 2>&1 || { tail -40 $O/r3_b_tests.log; exit 1; }
tail -3 $O/r3_b_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_slice8
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/r3_slice8 -- python3 $R/scripts/slice_timeline.py 8 5 > $O/r3_slice8.log 2>&1 || { tail -20 $O/r3_slice8.log; exit 1; }
grep "^rep" $O/r3_slice8.log
cd $R && python3 tools/timeline_summary.py $O/r3_slice8 > $O/r3_slice8_timeline.txt && cat $O/r3_slice8_timeline.txt
timeout -k 10 300 python3 scripts/slice_timeline.py 8 5 | grep "^rep"
timeout -k 10 300 python3 scripts/slice_timeline.py 1 4 | grep "^rep"
