#!/bin/bash
# Round 3: executed FP64 flops and HBM bytes of every launch bench.py prices a roofline entry on.
# Per group of scripts/profile_executed.py: one --kernel-trace run (durations) and three --pmc runs of their own
# (FP64 instruction counters; FETCH_SIZE; WRITE_SIZE).  tools/executed_flops.py condenses them into
# gpurun_out/r03_executed_flops.json (copied to profiles/ by hand, with the commit it was taken at).
# usage: gpu_r3_executed.sh [group ...]   (default: headline schemes config2 coupling)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3exec
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
GROUPS_="${@:-headline schemes config2 coupling}"
for G in $GROUPS_; do
  rm -rf $O/${G}_trace $O/${G}_flops $O/${G}_fetch $O/${G}_write
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/${G}_trace -- python3 $R/scripts/profile_executed.py $G $O/${G}_manifest.json 3 > $O/${G}_trace.log 2>&1 || { tail -20 $O/${G}_trace.log; exit 1; }
  echo "$G trace: $(tail -1 $O/${G}_trace.log)"
  timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/${G}_flops -- python3 $R/scripts/profile_executed.py $G $O/${G}_manifest.json 3 > $O/${G}_flops.log 2>&1 || { tail -20 $O/${G}_flops.log; exit 1; }
  echo "$G flops: $(tail -1 $O/${G}_flops.log)"
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${G}_fetch -- python3 $R/scripts/profile_executed.py $G $O/${G}_manifest.json 3 > $O/${G}_fetch.log 2>&1 || { tail -20 $O/${G}_fetch.log; exit 1; }
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${G}_write -- python3 $R/scripts/profile_executed.py $G $O/${G}_manifest.json 3 > $O/${G}_write.log 2>&1 || { tail -20 $O/${G}_write.log; exit 1; }
  echo "$G fetch/write done"
done
cd $R
python3 tools/executed_flops.py $O $R/gpurun_out/r03_executed_flops.json > $R/gpurun_out/r03_executed_flops.txt || exit 1
cat $R/gpurun_out/r03_executed_flops.txt
