#!/bin/bash
# SQ counters of the magnus6 discriminant launch of the bench (tams, ell=3, 10^4 frequencies)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmcm_*
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/pmcm_$N -- python3 $R/scripts/profile_driver.py 0 3 magnus6 > $O/pmcm_$N.log 2>&1 || { tail -20 $O/pmcm_$N.log; exit 1; }
done
echo done
