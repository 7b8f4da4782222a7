#!/bin/bash
# usage: build_fast_variant.sh NAME "extra flags" [schemes]  -> storm_amd/lib/variants/libstorm_hip_NAME.so
# The FMA build's scheme objects rebuilt with extra flags (the attribution switches of types.hpp); everything else from the
# regular build.  For tests/attribution_config3.py.
set -e
NAME=$1; FLAGS=$2; SCHEMES=${3:-1}
cd /root/repo/storm_amd/csrc
mkdir -p ../build/fvar_$NAME ../lib/variants
OBJS=""
for s in 0 1 2 3 4 5 6 7; do
  if [[ " $SCHEMES " == *" $s "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fPIC -Wno-unused-function -DSTORM_NS=storm_fast -DSTORM_FAST_BUILD=1 $FLAGS -DSTORM_SCHEME=$s -c scheme_tu.hip -o ../build/fvar_$NAME/fast_scheme_$s.o &
    OBJS="$OBJS ../build/fvar_$NAME/fast_scheme_$s.o"
  else
    OBJS="$OBJS ../build/fast_scheme_$s.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libstorm_hip_$NAME.so ../build/scheme_*.o $OBJS ../build/common_kernels.o ../build/fast_common_kernels.o ../build/postprocess.o ../build/coupling_kernels.o ../build/fast_coupling_kernels.o ../build/storm_hip.o
echo built $NAME
