"""profiles/rNN_kernel_stats.md from the rocprofv3 --stats CSV of scripts/profile_round.sh.
    python scripts/kernel_stats_md.py profiles/r02_kernel_stats.csv 5 profiles/r02_bench_under_profiler.json profiles/r02_bench_headline.json > profiles/r02_kernel_stats.md"""
import csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
runs = int(sys.argv[2])
line = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
plain = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1]) if len(sys.argv) > 4 else None
total = sum(float(r["TotalDurationNs"]) for r in rows) / runs / 1e6

def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("Cijk_"):
        m = re.match(r"(Cijk_[A-Za-z]+_[A-Za-z]+_[A-Z_]*?MT\d+x\d+x\d+)", name)
        lib = "hipBLASLt fp16 x fp16 -> fp32 (gemm_f16x2) " if "_HSS_" in name else "rocBLAS sgemm "
        return lib + (m.group(1) if m else name[:40])
    return name.split("(")[0][:70]

tag = re.search(r"(r\d+)_", sys.argv[1]).group(1)
print(f"# rocprofv3 --kernel-trace --stats summary, round {int(tag[1:])} (MI355X, 1 GPU)\n")
print("Command (scripts/profile_round.sh, run from /tmp on the GPU box): `rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-input`")
print(f"(workload {line['config']['workload']}, final kernel sources of the round; the run holds {runs} decompositions: 1 warm-up, 3 timed, 1 instrumented; the columns below are per decomposition).")
print(f"Raw stats: `{sys.argv[1]}`.  Bench line of the same run (under the profiler: {line['ms_per_step']:.0f} ms per step): `{sys.argv[3]}`.  Table written by `scripts/kernel_stats_md.py`.\n")
print(f"Total kernel time {total:.1f} ms per decomposition (kernels of the second stream - Cholesky next to the M^T Z product, result downloads - included).\n")
print("| kernel | launches / decomposition | ms / decomposition | average us | share |\n|---|---|---|---|---|")
for r in rows[:34]:
    ms = float(r["TotalDurationNs"]) / runs / 1e6
    print(f"| `{short(r['Name'])}` | {int(r['Calls']) // runs} | {ms:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {100 * ms / total:.1f} % |")
rl = line["roofline"]
sym = [r for r in rows if "sytrd_symv_kernel" in r["Name"]][0]
atx = [r for r in rows if "tile_atx_dma_kernel" in r["Name"]]
print(f"\nHow these rows map to the `roofline` objects of the bench line: `sytrd_symv_kernel<512>` averages {float(sym['AverageNs']) / 1e3:.1f} us over all launches here; "
      f"`bench.py` times every 64th launch with a HIP event pair on the stream (the pair around one short launch adds the launch gap, the profiler some more): "
      f"{rl['avg_launch_ms'] * 1e3:.1f} us in this profiled run"
      + (f", {plain['roofline']['avg_launch_ms'] * 1e3:.1f} us in the unprofiled one (`{sys.argv[4]}`, {plain['ms_per_step']:.0f} ms per step): "
         f"{plain['roofline']['algorithmic_mb_per_launch']:.1f} MB / {plain['roofline']['avg_launch_ms'] * 1e3:.1f} us = {plain['roofline']['achieved'] / 1e3:.1f} TB/s = {plain['roofline']['frac']:.2f} of the 8 TB/s HBM peak" if plain else "")
      + f" ({rl['algorithmic_mb_per_launch'] / (float(sym['AverageNs']) / 1e3) / 8e3 * 1e3:.2f} with the profiler's kernel duration)."
      + (f"  `tile_atx_dma_kernel<25>`: {float(atx[0]['AverageNs']) / 1e6:.1f} ms average over its launches (the d x T ones take 12.9-13.6 ms, the short sketch launches pull the average down); "
         f"the bench line's `roofline_mfma` has the d x T launches alone: {line['roofline_mfma']['achieved']:.0f} TFLOP/s = {line['roofline_mfma']['frac']:.2f} of the fp32-MFMA peak." if atx else ""))
print("Of the launches of `tile_atx_dma_kernel<25>`, two are the d x T products of the tile stage (13.2 ms each) and one is the "
      "rank-adaptive projection of the whole movie (tiles that kept <= 32 components run half the MFMA work: 7.7 ms).  `f16x2_absmax_kernel` (operand maxima of the "
      "fp16-piece products) reads 1 ms per launch under the profiler; in the unprofiled run the whole split (maxima, read-back, pieces) "
      "costs 8.8 ms per decomposition by HIP events (`f16x2_split` in the bench line).\n")
print("Counter passes (separate runs, FETCH_SIZE only, restricted to the kernel, taken on the same sources - their SHA-256 is recorded and checked by `bench.py`): "
      f"`profiles/{tag}_pmc_sytrd_n10000.json`, `profiles/{tag}_pmc_tile_atx.json`.")
