#!/bin/bash
# The matrix-pipe walk of the coupling integrals (FMA build, M = 1024 modes of ell = 0, 2 grouped by degree): rocprofv3 kernel
# stats and counter passes (own runs): FP64 MFMA / VALU instructions and busy cycles, HBM fetch.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/cpm_*
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cpm_trace -- python3 $R/scripts/profile_coupling.py 1024 3 1 1 > $O/cpm_trace.log 2>&1 || { tail -20 $O/cpm_trace.log; exit 1; }
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" FETCH_SIZE; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $O/cpm_$N -- python3 $R/scripts/profile_coupling.py 1024 2 1 1 > $O/cpm_$N.log 2>&1 || { tail -20 $O/cpm_$N.log; exit 1; }
done
cd $R
{
  echo "# storm_coupling_matrices, FMA build, M = 1024 modes of ell = 0, 2 on tams grouped by degree (scripts/gpu_pmc_coupling_matrix.sh)"
  python3 tools/rocprof_summary.py $O/cpm_trace | grep "coupling\|^#\|^kernel"
  for d in $O/cpm_SQ_WAVES/ $O/cpm_FETCH_SIZE/; do python3 tools/rocprof_summary.py $d; done | grep "coupling_matrix\|coupling_kernel\|^#"
  grep "^ok" $O/cpm_trace.log
} > $O/r2_pmc_coupling_matrix.txt
cat $O/r2_pmc_coupling_matrix.txt
