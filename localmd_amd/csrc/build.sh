#!/bin/bash
# Build libpmd_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libpmd_hip.so
SRCS="capi.hip rng.hip prep.hip tile_gemm.hip small_la.hip pipeline.hip global.hip sytrd.hip sytrd2.hip expand.hip diag.hip comm.hip"
OBJS=""
mkdir -p build
for s in $SRCS; do
  o=build/${s%.hip}.o
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ pmd_common.h -nt "$o" ] || [ pmd_internal.h -nt "$o" ] || [ ../../include/pmd_hip.h -nt "$o" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c "$s" -o "$o" ${PMD_EXTRA_FLAGS} &
  fi
  OBJS="$OBJS $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT -L/opt/rocm/lib -lrocblas -lrocsolver -ldl -Wl,-rpath,/opt/rocm/lib
echo "built $(realpath $OUT)"
