#!/bin/bash
# Debug build of the library with the phase stamps of sytrd.hip (-DPMD_SYMV_TRACE), for scripts/symv_trace.py.
# Usage: bash scripts/build_trace_lib.sh [column]   (column whose symv / advance launches are traced, default 32)
set -e
cd "$(dirname "$0")/../localmd_amd/csrc"
bash build.sh > /dev/null
COL=${1:-32}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DPMD_SYMV_TRACE -DPMD_SYMV_TRACE_J=$COL -c sytrd.hip -o build/sytrd_trace.o
OBJS=$(ls build/*.o | grep -v -e "build/sytrd.o" -e "build/sytrd_trace.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/sytrd_trace.o -o ../libpmd_hip_trace.so -L/opt/rocm/lib -lrocblas -lrocsolver -Wl,-rpath,/opt/rocm/lib
echo "built $(realpath ../libpmd_hip_trace.so) (column $COL)"
