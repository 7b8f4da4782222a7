#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 900 python3 -u -m pytest tests/test_hip_full_size.py tests/test_hip_fast_mode.py tests/test_hip_segments.py tests/test_hip_multi.py -x -q --timeout 300 --timeout-method thread > $O/r3_c_tests.log 2>&1 || { tail -40 $O/r3_c_tests.log; exit 1; }
tail -3 $O/r3_c_tests.log
timeout -k 10 300 python3 scripts/slice_timeline.py 8 5 | grep "^rep"
timeout -k 10 300 python3 scripts/slice_timeline.py 1 4 | grep "^rep"
timeout -k 10 600 python3 bench.py --steps 10 --warmup 2 --legs headline --no-cpu-baseline > $O/r3_bench_c.json 2> $O/r3_bench_c.err || { tail -20 $O/r3_bench_c.err; exit 1; }
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench_c.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], json.dumps(d['phases']), json.dumps(d['multi_in_library']))
PY
