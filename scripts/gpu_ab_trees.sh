#!/bin/bash
# Same-box A/B of two trees (tmp_old = baseline, . = candidate), alternating: per-scheme whole-batch launch times in the FMA build.
# usage: gpu_ab_trees.sh "schemes" "n list" "lanes list" [reps]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
SCHEMES=${1:-colloc4}; NS=${2:-100000}; LANES=${3:-0}; REPS=${4:-2}
: > $O/ab_trees.txt
for rep in $(seq $REPS); do
  for tree in tmp_old .; do
    echo "== tree $tree rep $rep" >> $O/ab_trees.txt
    for n in $NS; do for l in $LANES; do
      (cd $R/$tree && timeout -k 10 300 python scripts/time_scheme.py $SCHEMES $n $l 2>&1 | grep "arith=") >> $O/ab_trees.txt || exit 1
    done; done
  done
done
cat $O/ab_trees.txt
