#!/bin/bash
# rocprofv3 --pmc passes (one counter group per run) over the 64-rate batch pipeline, one group at a time
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
python3 $R/tools/make_sweep_batch.py $O/sweep64.txt 64 > /dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmcb_*
for C in FETCH_SIZE "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VMEM SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" WRITE_SIZE; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/pmcb_$N -- $R/tools/storm_batch.bin $R/tools/tams_model.bin $O/sweep64.txt $O/sweep64.out --group 32 --summary --pipeline 1 > $O/pmcb_$N.log 2>&1 || { tail -20 $O/pmcb_$N.log; exit 1; }
done
echo done
