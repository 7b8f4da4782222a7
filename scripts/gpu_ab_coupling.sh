#!/bin/bash
# Same-box A/B (tmp_old = baseline, . = candidate): coupling kernel times and the bench's coupling / batch-pipeline legs.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
: > $O/ab_coupling.txt
for rep in 1 2; do for tree in tmp_old .; do
  echo "== tree $tree rep $rep" >> $O/ab_coupling.txt
  (cd $R/$tree && timeout -k 10 200 python scripts/coupling_bench.py 512 1024 2>&1 | grep "tile=16 arith=1") >> $O/ab_coupling.txt || exit 1
  (cd $R/$tree && timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['secondary']
print('coupling leg fast ms', s['coupling_integrals']['fast']['avg_launch_ms'], 'batch wall', s['batch_pipeline']['wall_ms'], 'fast', s['batch_pipeline']['fast']['wall_ms'])") >> $O/ab_coupling.txt || exit 1
done; done
cat $O/ab_coupling.txt
