mkdir -p gpurun_out/r3
for c in demo_60x80x2000 256x256x2000_b20_r8 1024x1024x2000_b32 1024x1024x1000_b16 1024x1024x8000_b32; do
  timeout -k 10 500 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3/bench_$c.log 2>&1 || exit 1
  tail -1 gpurun_out/r3/bench_$c.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], round(d['value'],1), round(d['ms_per_step'],1), {k:round(v,1) for k,v in list(d['kernel_ms_per_step'].items())[:8]})"
done
timeout -k 10 600 python bench.py --config 1024x1024x20000_b32 --steps 1 --warmup 1 --no-cpu-baseline --no-host-input > gpurun_out/r3/bench_c4.log 2>&1 || exit 1
tail -1 gpurun_out/r3/bench_c4.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], round(d['value'],1), round(d['ms_per_step'],1), {k:round(v,1) for k,v in list(d['kernel_ms_per_step'].items())[:8]})"
timeout -k 10 600 python bench.py --config 2048x2048x5000_b16 --steps 1 --warmup 0 --no-cpu-baseline --no-host-input > gpurun_out/r3/bench_c5.log 2>&1 || exit 1
tail -1 gpurun_out/r3/bench_c5.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], round(d['value'],1), round(d['ms_per_step'],1), {k:round(v,1) for k,v in list(d['kernel_ms_per_step'].items())[:8]})"
