#!/bin/bash
# tools/storm_batch on N rotation rates of the reference's deformed-star batch (ell = 1, 3, 5, 7): wall time and md5
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
N=${1:-256}
mkdir -p $O
python3 $R/tools/make_sweep_batch.py $O/sweep$N.txt $N > /dev/null
for rep in 1 2; do
  timeout -k 10 400 $R/tools/storm_batch.bin $R/tools/tams_model.bin $O/sweep$N.txt $O/sweep$N.out --summary 2> $O/storm_batch$N.log
  tail -4 $O/storm_batch$N.log
  md5sum $O/sweep$N.out
done
