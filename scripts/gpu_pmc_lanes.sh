#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for L in 4 16; do
  rm -rf $O/pmcl_$L*
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA"; do
    N=$(echo $C | cut -d' ' -f1)
    timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/pmcl_${L}_$N -- python3 $R/scripts/profile_lanes.py $L 1 98304 > $O/pmcl_${L}_$N.log 2>&1 || { tail -20 $O/pmcl_${L}_$N.log; exit 1; }
    python3 $R/tools/rocprof_summary.py $O/pmcl_${L}_$N | grep -E "det_(col|hex)"
  done
done
