#!/bin/bash
# All kernels of the 64-rate batch pipeline in the FMA build (one group at a time): rocprofv3 kernel stats.  Round 3.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/make_sweep_batch.py $O/sweep64.txt 64 > /dev/null || exit 1
python3 $R/tools/make_model_bin.py tams /tmp/tams_model.bin > /dev/null || exit 1
rm -rf $O/b64f
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b64f -- $R/tools/storm_batch.bin /tmp/tams_model.bin $O/sweep64.txt $O/sweep64_f.out --group 32 --summary --pipeline 1 --arithmetic 1 > $O/b64f.log 2>&1 || { tail -20 $O/b64f.log; exit 1; }
{
  echo "# rocprofv3 --kernel-trace --stats -- tools/storm_batch.bin tams_model.bin sweep64.txt out --group 32 --summary --pipeline 1 --arithmetic 1  (round 3)"
  python3 $R/tools/rocprof_summary.py $O/b64f
} > $O/r3_storm_batch64_fast_kernel_stats.txt
head -16 $O/r3_storm_batch64_fast_kernel_stats.txt | cut -c1-160
grep "storm_batch:" $O/b64f.log | tail -1 | cut -c1-300
