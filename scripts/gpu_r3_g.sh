#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 1100 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/r3_g_tests.log 2>&1 || { tail -40 $O/r3_g_tests.log; exit 1; }
tail -2 $O/r3_g_tests.log
timeout -k 10 300 python3 scripts/strong_slice_timing.py 2>/dev/null > $O/r3_strong_slice_timing.txt || exit 1
cat $O/r3_strong_slice_timing.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_slice8
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/r3_slice8 -- python3 $R/scripts/slice_timeline.py 8 5 > $O/r3_slice8.log 2>&1 || { tail -20 $O/r3_slice8.log; exit 1; }
cd $R && python3 tools/timeline_summary.py $O/r3_slice8 > $O/r3_slice8_timeline.txt && head -24 $O/r3_slice8_timeline.txt
