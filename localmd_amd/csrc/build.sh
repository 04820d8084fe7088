#!/bin/bash
# Build libpmd_hip.so for gfx950 (cross-compiles without a GPU).
# Compiles run in parallel; every one is waited for by PID and a failed compile removes its object and fails the
# build, so a stale object can never be linked in place of a source that no longer compiles.
set -e
cd "$(dirname "$0")"
OUT=../libpmd_hip.so
SRCS="capi.hip rng.hip prep.hip tile_gemm.hip small_la.hip wide.hip pipeline.hip global.hip gemm_f16x2.hip sytrd.hip sytrd2.hip expand.hip diag.hip comm.hip"
OBJS=""
PIDS=()
NAMES=()
mkdir -p build
for s in $SRCS; do
  o=build/${s%.hip}.o
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ pmd_common.h -nt "$o" ] || [ pmd_internal.h -nt "$o" ] || [ ../../include/pmd_hip.h -nt "$o" ]; then
    rm -f "$o"
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c "$s" -o "$o" ${PMD_EXTRA_FLAGS} &
    PIDS+=($!)
    NAMES+=("$s")
  fi
  OBJS="$OBJS $o"
done
FAILED=""
for i in "${!PIDS[@]}"; do
  if ! wait "${PIDS[$i]}"; then
    FAILED="$FAILED ${NAMES[$i]}"
    rm -f "build/${NAMES[$i]%.hip}.o"
  fi
done
if [ -n "$FAILED" ]; then
  echo "build.sh: compile failed for:$FAILED" >&2
  exit 1
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT -L/opt/rocm/lib -lrocblas -lrocsolver -lhipblaslt -ldl -Wl,-rpath,/opt/rocm/lib
echo "built $(realpath $OUT)"
