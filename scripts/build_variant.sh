#!/bin/bash
# usage: build_variant.sh NAME "extra flags" [OPT]   -> storm_amd/lib/variants/libstorm_hip_NAME.so (parity scheme objects rebuilt)
set -e
NAME=$1; FLAGS=$2; OPT=${3:--O3}
cd /root/repo/storm_amd/csrc
mkdir -p ../build/var_$NAME ../lib/variants
for s in 0 1 2 3 4 5 6 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $OPT -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function $FLAGS -DSTORM_SCHEME=$s -c scheme_tu.hip -o ../build/var_$NAME/scheme_$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libstorm_hip_$NAME.so ../build/var_$NAME/scheme_*.o ../build/fast_scheme_*.o ../build/common_kernels.o ../build/fast_common_kernels.o ../build/storm_hip.o
echo built $NAME
