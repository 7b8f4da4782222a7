#!/bin/bash
# HBM/L2 fetch of the batch coupling kernel with 8 x 8 and 16 x 16 tiles (one --pmc pass per run, program directly after --).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/make_sweep_batch.py $O/sweep64.txt 64 > /dev/null || exit 1
for T in 8 16; do
  rm -rf $O/cpf_$T
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cpf_$T -- $R/tools/storm_batch.bin $R/tools/tams_model.bin $O/sweep64.txt $O/sweep64_f$T.out --group 32 --summary --pipeline 1 --coupling-tile $T > $O/cpf_$T.log 2>&1 || { tail -20 $O/cpf_$T.log; exit 1; }
  echo "tile $T"
  python3 $R/tools/rocprof_summary.py $O/cpf_$T | grep "coupling_batch" | cut -c1-200
done
