#!/bin/bash
# Round check on the GPU box: tests, bench, rocprofv3 kernel stats of the bench, PMC passes on the dominant launch.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R || exit 1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/round_tests.log 2>&1 || { tail -30 $O/round_tests.log; exit 1; }
tail -2 $O/round_tests.log
timeout -k 10 400 python -u bench.py > $O/round_bench.log 2>&1 || { tail -30 $O/round_bench.log; exit 1; }
tail -1 $O/round_bench.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_bench $O/prof_t1 $O/prof_sweep $O/pmc_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 3 --no-cpu-baseline --single-mode --no-sweep > $O/prof_bench.log 2>&1 || { tail -20 $O/prof_bench.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t1 -- python3 $R/bench.py --steps 3 --no-cpu-baseline --single-mode --no-sweep --threads 1 > $O/prof_t1.log 2>&1 || { tail -20 $O/prof_t1.log; exit 1; }
# the rotation sweep (storm_sweep): 256 systems x 300 frequencies, twice (the second one with warm buffer caches)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sweep -- $R/tools/many_scans.bin $R/tools/tams_model.bin 256 300 1 -256 -256 > $O/prof_sweep.log 2>&1 || { tail -20 $O/prof_sweep.log; exit 1; }
grep -a '^sweep' $O/prof_sweep.log
# the batch-input pipeline (tools/storm_batch): 64 rotation rates end to end
python3 $R/tools/make_sweep_batch.py $O/sweep64.txt 64 > /dev/null && timeout -k 10 300 $R/tools/storm_batch.bin $R/tools/tams_model.bin $O/sweep64.txt $O/sweep64.out --group 32 --summary 2> $O/storm_batch.log || { tail -5 $O/storm_batch.log; exit 1; }
cat $O/storm_batch.log
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$N -- python3 $R/scripts/profile_driver.py 0 3 > $O/pmc_$N.log 2>&1 || { tail -20 $O/pmc_$N.log; exit 1; }
done
echo done
