#!/bin/bash
# SQ counter passes (own runs) over the coupling kernel of the FMA build and of the parity build, M = 1024
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/cpf_*
for A in 1 0; do
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"; do
    N=$(echo $C | cut -d' ' -f1)
    timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $O/cpf_a${A}_$N -- python3 $R/scripts/profile_coupling.py 1024 2 $A > $O/cpf_a${A}_$N.log 2>&1 || { tail -20 $O/cpf_a${A}_$N.log; exit 1; }
  done
done
cd $R
for A in 1 0; do
  echo "# arithmetic $A (1 = FMA build), storm_coupling_matrices, M = 1024 modes of ell = 0, 2 on tams"
  for d in $O/cpf_a${A}_*/; do python3 tools/rocprof_summary.py $d; done | grep "coupling_kernel\|^#"
  grep "^ok" $O/cpf_a${A}_SQ_WAVES.log
done > $O/r2_pmc_coupling_m1024.txt
cat $O/r2_pmc_coupling_m1024.txt
