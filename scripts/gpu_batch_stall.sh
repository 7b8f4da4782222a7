#!/bin/bash
# Which HIP call holds a slow run of the 1001-rate batch?  rocprofv3 runtime trace, per-API statistics of each run.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/make_sweep_batch.py /tmp/sweep1001.txt 1001 > /dev/null || exit 1
for K in 1 2 3 4; do
  rm -rf /tmp/stall_$K
  timeout -k 10 280 rocprofv3 --hip-runtime-trace --stats --output-format csv -d /tmp/stall_$K -- $R/tools/storm_batch.bin $R/tools/tams_model.bin /tmp/sweep1001.txt /tmp/sweep1001.out --group 32 --summary --arithmetic 1 > /tmp/stall_$K.log 2>&1 || { tail -5 /tmp/stall_$K.log; exit 1; }
  echo "run $K: $(grep 'blocks,' /tmp/stall_$K.log | cut -c1-110)"
  F=$(ls /tmp/stall_$K/*/*hip_api_stats.csv 2>/dev/null | head -1)
  [ -n "$F" ] && { head -1 $F; sort -t, -k4 -nr $F | head -8; cp $F $O/stall_${K}_hip_api_stats.csv; }
done
