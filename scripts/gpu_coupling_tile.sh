#!/bin/bash
# Coupling kernel of the 64-rate batch pipeline with 8 x 8 and 16 x 16 tiles, both arithmetic builds: kernel stats per run.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/make_sweep_batch.py $O/sweep64.txt 64 > /dev/null || exit 1
for T in ${TILES:-8 16}; do
  for A in 0 1; do
    rm -rf $O/cpt_${T}_$A
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cpt_${T}_$A -- $R/tools/storm_batch.bin $R/tools/tams_model.bin $O/sweep64.txt $O/sweep64_${T}_$A.out --group 32 --summary --pipeline 1 --coupling-tile $T --arithmetic $A > $O/cpt_${T}_$A.log 2>&1 || { tail -20 $O/cpt_${T}_$A.log; exit 1; }
    echo "tile $T arithmetic $A"
    python3 $R/tools/rocprof_summary.py $O/cpt_${T}_$A | grep "coupling\|total device" | cut -c1-200
  done
done
cmp $O/sweep64_8_0.out $O/sweep64_16_0.out && echo "parity outputs identical for both tiles"
