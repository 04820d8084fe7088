#!/bin/bash
# Profiles of one round (run on the GPU box through gpurun): rocprofv3 kernel trace + stats of the headline bench, and the
# two counter passes (FETCH_SIZE of the symv and of tile_atx, each in a run of its own, restricted to the kernel).
#   bash scripts/profile_round.sh r02
set -e
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-host-input > "$OUT/bench_line.json" 2> "$OUT/bench.err" || echo "bench profile failed" >&2
echo "[profile] bench trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv --kernel-include-regex sytrd_symv -d "$OUT/pmc_sytrd" -o pmc -- python3 "$ROOT/scripts/sytrd_prof.py" 10000 1 1 > "$OUT/pmc_sytrd.log" 2>&1 || echo "sytrd pmc failed" >&2
echo "[profile] sytrd counter pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv --kernel-include-regex tile_atx -d "$OUT/pmc_atx" -o pmc -- python3 "$ROOT/scripts/atx_pmc.py" > "$OUT/pmc_atx.log" 2>&1 || echo "atx pmc failed" >&2
echo "[profile] atx counter pass done"
cd "$ROOT"
CSV=$(find "$OUT/pmc_sytrd" -name "*counter_collection.csv" | head -1)
[ -n "$CSV" ] && python3 scripts/pmc_sytrd.py "$CSV" 10000 "$OUT/${TAG}_pmc_sytrd_n10000.json" > /dev/null
CSV=$(find "$OUT/pmc_atx" -name "*counter_collection.csv" | head -1)
[ -n "$CSV" ] && python3 scripts/pmc_atx.py "$OUT/${TAG}_pmc_tile_atx.json" "$CSV" > /dev/null
STATS=$(find "$OUT/bench" -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && cp "$STATS" "$OUT/${TAG}_kernel_stats.csv"
# the raw traces are large: keep the summaries only
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*.db" -delete
ls -la "$OUT"
