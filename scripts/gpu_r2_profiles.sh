#!/bin/bash
# Round-2 profiles: rocprofv3 kernel stats of the bench headline, PMC passes (own runs) on its dominant launch in both builds.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r2prof_* $O/r2pmc_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r2prof_bench -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --legs headline > $O/r2prof_bench.log 2>&1 || { tail -20 $O/r2prof_bench.log; exit 1; }
tail -1 $O/r2prof_bench.log | cut -c1-300
for A in 1 0; do
  for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES"; do
    N=$(echo $C | cut -d' ' -f1)
    timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/r2pmc_a${A}_$N -- python3 $R/scripts/profile_config3.py $A 3 > $O/r2pmc_a${A}_$N.log 2>&1 || { tail -20 $O/r2pmc_a${A}_$N.log; exit 1; }
  done
done
cd $R
python3 tools/rocprof_summary.py $O/r2prof_bench > $O/r2_bench_headline_kernel_stats.txt
for A in 1 0; do
  for d in $O/r2pmc_a${A}_*/; do python3 tools/rocprof_summary.py $d; done > $O/r2_pmc_det_hex_colloc4_arith$A.txt
done
head -12 $O/r2_bench_headline_kernel_stats.txt
cat $O/r2_pmc_det_hex_colloc4_arith1.txt | grep -v "^#" | head -40
echo done
