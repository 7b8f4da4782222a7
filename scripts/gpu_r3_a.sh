#!/bin/bash
# round 3, first batch: attribution experiment, the new full-size tests, bench headline
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python3 tests/attribution_config3.py > $O/attribution_config3.txt 2>&1 || { tail -30 $O/attribution_config3.txt; exit 1; }
cat $O/attribution_config3.txt
timeout -k 10 600 python3 -u -m pytest tests/test_hip_full_size.py -x -q -k "config4 or config5" --timeout 300 --timeout-method thread > $O/r3_cfg45.log 2>&1 || { tail -40 $O/r3_cfg45.log; exit 1; }
tail -5 $O/r3_cfg45.log
timeout -k 10 600 python3 bench.py --steps 10 --warmup 2 > $O/r3_bench_a.json 2> $O/r3_bench_a.err || { tail -20 $O/r3_bench_a.err; exit 1; }
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench_a.json'))
print(d['value'], d['ms_per_step'], json.dumps({k:v for k,v in d['roofline'].items() if k not in ('note','rocprof_row','executed_from','pipe')}))
print(json.dumps(d['secondary']['coupling_integrals'])[:1500])
PY
