#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout -k 10 1100 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/r3_d_tests.log 2>&1 || { tail -40 $O/r3_d_tests.log; exit 1; }
tail -3 $O/r3_d_tests.log
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 --dist-backend gloo > $O/r3_bench_2rank_gloo.json 2> $O/r3_bench_2rank_gloo.err || { tail -20 $O/r3_bench_2rank_gloo.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench_2rank_gloo.json').read().strip().splitlines()[-1])
print(d['n_gpus'], d['value'], d['ms_per_step'], json.dumps(d['phases']), json.dumps(d['multi_in_library']))
PY
