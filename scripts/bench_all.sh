#!/bin/bash
# Every bench workload of the round in one go (GPU box): JSON lines under gpurun_out/bench_<tag>/
#   bash scripts/bench_all.sh r03
TAG=${1:-rXX}
OUT=gpurun_out/bench_$TAG
mkdir -p $OUT
run() {  # name, args...
  local name=$1; shift
  timeout -k 10 900 python bench.py "$@" > $OUT/$name.log 2> $OUT/$name.err || { echo "$name FAILED"; tail -3 $OUT/$name.err; return 1; }
  tail -1 $OUT/$name.log > $OUT/$name.json
  python - "$OUT/$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"]["workload"], "n_gpus", d["n_gpus"], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 1), "ms", "host-array", d.get("frames_per_s_from_host_array"), "cold", d.get("cold_first_step_ms"))
PY
}
run default_run || exit 1
run demo_60x80x2000 --config demo_60x80x2000 --steps 5 --warmup 2 --no-cpu-baseline || exit 1
run 256x256x2000_b20_r8 --config 256x256x2000_b20_r8 --steps 5 --warmup 2 --no-cpu-baseline || exit 1
run 1024x1024x2000_b32 --config 1024x1024x2000_b32 --steps 3 --warmup 1 --no-cpu-baseline || exit 1
run 1024x1024x1000_b16 --config 1024x1024x1000_b16 --steps 3 --warmup 1 --no-cpu-baseline || exit 1
run 1024x1024x8000_b32 --config 1024x1024x8000_b32 --steps 3 --warmup 1 --no-cpu-baseline || exit 1
run 1024x1024x20000_b32_one_gpu --config 1024x1024x20000_b32 --steps 1 --warmup 1 --no-cpu-baseline --no-host-input || exit 1
run 2048x2048x5000_b16_one_gpu --config 2048x2048x5000_b16 --steps 2 --warmup 2 --no-cpu-baseline || exit 1
PMD_GEMM_SPLIT=0 run default_sgemm_only --steps 3 --warmup 1 --no-cpu-baseline --no-host-input || exit 1
