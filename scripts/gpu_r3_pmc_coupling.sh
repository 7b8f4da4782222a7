#!/bin/bash
# Round 3: the matrix-pipe walk of the coupling integrals on the ordered mode list (FMA build), M = 1024 modes of ell = 0, 2 in
# caller order (NOT grouped: the library orders them), 16 x 16 and 32 x 32 tiles: kernel stats and counter passes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
M=${1:-1024}
rm -rf $O/cp3_*
for T in 16 32; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cp3_trace_$T -- python3 $R/scripts/profile_coupling.py $M 3 1 0 $T > $O/cp3_trace_$T.log 2>&1 || { tail -20 $O/cp3_trace_$T.log; exit 1; }
  for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" FETCH_SIZE; do
    N=$(echo $C | cut -d' ' -f1)
    timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $O/cp3_${T}_$N -- python3 $R/scripts/profile_coupling.py $M 2 1 0 $T > $O/cp3_${T}_$N.log 2>&1 || { tail -20 $O/cp3_${T}_$N.log; exit 1; }
  done
done
cd $R
{
  echo "# storm_coupling_matrices, FMA build, M = $M modes of ell = 0, 2 on tams in caller order (scripts/gpu_r3_pmc_coupling.sh): the matrix-pipe walk"
  echo "# on the ordered mode list takes every tile.  Per tile of the walk: kernel stats, then counters per dispatch (separate passes)."
  for T in 16 32; do
    echo "## tile $T"
    python3 tools/rocprof_summary.py $O/cp3_trace_$T | grep "coupling\|^#\|^kernel"
    for d in $O/cp3_${T}_SQ_WAVES/ $O/cp3_${T}_SQ_INSTS_LDS/ $O/cp3_${T}_FETCH_SIZE/; do python3 tools/rocprof_summary.py $d; done | grep "coupling_matrix"
    grep "^ok" $O/cp3_trace_$T.log
  done
} > $O/r3_pmc_coupling_matrix.txt
cat $O/r3_pmc_coupling_matrix.txt
