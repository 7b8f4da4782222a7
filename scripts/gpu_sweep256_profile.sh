#!/bin/bash
# rocprofv3 kernel stats of a 256-system sweep (tools/many_scans.bin, twice) next to its wall-clock phases
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/sweep256_prof
timeout -k 10 200 $R/tools/many_scans.bin $R/tools/tams_model.bin 256 300 1 -256 -256 > $O/sweep256_plain.log 2>&1 || { tail $O/sweep256_plain.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/sweep256_prof -- $R/tools/many_scans.bin $R/tools/tams_model.bin 256 300 1 -256 -256 > $O/sweep256_prof.log 2>&1 || { tail $O/sweep256_prof.log; exit 1; }
cd $R
cat $O/sweep256_plain.log
python3 tools/rocprof_summary.py $O/sweep256_prof > $O/sweep256_kernel_stats.txt
head -30 $O/sweep256_kernel_stats.txt
