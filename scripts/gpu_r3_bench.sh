#!/bin/bash
# Round 3: the default bench.py run, then the same command's headline legs under rocprofv3 --kernel-trace --stats.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python3 bench.py > $O/r3_bench_default.json 2> $O/r3_bench_default.err || { tail -20 $O/r3_bench_default.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench_default.json').read().strip().splitlines()[-1])
r=d['roofline']
print("step_ms", d["step_ms"], "parity", d["parity_arithmetic"]["step_ms"]); print("value %.4g ms/step %.3f frac %.4f ref-eq %.3f launch %.3f ms | parity %.4g frac %.4f" % (d['value'], d['ms_per_step'], r['frac'], r['reference_equivalent']['ratio_to_peak'], r['avg_launch_ms'], d['parity_arithmetic']['value'], d['parity_arithmetic']['roofline']['frac']))
print("cpu", d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['one_thread']['value'], "gpu/cpu", d['gpu_over_cpu'])
s=d['secondary']
print("cfg2", {k:(v['ms_per_step'], v['roofline']['frac']) for k,v in s['config2'].items() if isinstance(v,dict)})
print("coupling", {k:(v['avg_launch_ms'], v['frac'], v['reference_equivalent']['ratio_to_peak']) for k,v in s['coupling_integrals'].items() if isinstance(v,dict) and 'frac' in v})
print("sweep", s["rotation_sweep"]["ms_per_scan"], "scaled", s.get("scaled_scan"), "batch", s.get("batch_pipeline"))
print("phases", d['phases']['max_over_ranks'], "multi", d['multi_in_library'])
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3prof_bench
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3prof_bench -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --legs headline > $O/r3prof_bench.log 2>&1 || { tail -20 $O/r3prof_bench.log; exit 1; }
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --legs headline  (round 3)"; python3 tools/rocprof_summary.py $O/r3prof_bench; } > $O/r3_bench_headline_kernel_stats.txt
head -14 $O/r3_bench_headline_kernel_stats.txt | cut -c1-170
# two-rank rehearsal of --gpus 2 on the one GPU (gloo for the collectives, both ranks' kernels share device 0: the phase
# times show the exchange path's host cost, not a scaling figure)
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 --dist-backend gloo --no-cpu-baseline --legs headline > $O/r3_bench_gloo2.json 2> $O/r3_bench_gloo2.err || { tail -20 $O/r3_bench_gloo2.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench_gloo2.json').read().strip().splitlines()[-1])
print("gloo2: ms/step", d['ms_per_step'], "phases", d['phases'], "multi", d.get('multi_in_library'))
PY
