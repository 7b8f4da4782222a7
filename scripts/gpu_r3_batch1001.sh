#!/bin/bash
# Round 3: the generator's full input for one azimuthal order (1001 rotation rates) through tools/storm_batch, both builds,
# for 1, 2, 4 groups in flight.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
python3 $R/tools/make_sweep_batch.py /tmp/sweep1001.txt 1001 > /dev/null || exit 1
: > $O/r3_batch1001.txt
for A in 1 0; do
  for P in 4 4 2 1; do
    timeout -k 10 200 $R/tools/storm_batch.bin $R/tools/tams_model.bin /tmp/sweep1001.txt /tmp/sweep1001.out --group 32 --summary --pipeline $P --arithmetic $A > /tmp/b.log 2>&1 || { tail -5 /tmp/b.log; exit 1; }
    echo "arithmetic $A pipeline $P: $(grep 'blocks,' /tmp/b.log | cut -c1-200) md5 $(md5sum < /tmp/sweep1001.out | cut -c1-12)" | tee -a $O/r3_batch1001.txt
  done
done
