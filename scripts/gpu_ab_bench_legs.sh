set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for tree in tmp_old .; do
  (cd $R/$tree && timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bs_$(basename $tree | tr . n)_$rep.log 2>/dev/null) || exit 1
done; done
