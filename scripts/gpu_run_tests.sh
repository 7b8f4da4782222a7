#!/bin/bash
# usage: gpu_run_tests.sh <log name> <pytest args...>   -- one pytest process under a time limit, log under gpurun_out/
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R || exit 1
LOG=$O/$1.log; shift
timeout -k 10 900 python -u -m pytest "$@" -m gpu -x -q --timeout 400 --timeout-method thread > $LOG 2>&1 || { tail -60 $LOG; exit 1; }
tail -3 $LOG
