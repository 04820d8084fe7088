"""Turns the counter CSVs of
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex tile_atx -d <dir> -- python3 scripts/atx_pmc.py
(and optionally a second pass with WRITE_SIZE) into the JSON bench.py reads for `roofline_mfma.traffic`, with the
SHA-256 of the kernel's source file so bench.py can tell when the figure has gone stale.
    python scripts/pmc_atx.py <out.json> <counter_collection.csv> [<counter_collection.csv> ...]"""
import csv, hashlib, json, os, sys

out = sys.argv[1]
acc = {}
for path in sys.argv[2:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            if "tile_atx" in r["Kernel_Name"]:
                a = acc.setdefault(r["Counter_Name"], [0, 0.0, r["Kernel_Name"][:80]])
                a[0] += 1
                a[1] += float(r["Counter_Value"])
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "localmd_amd", "csrc", "tile_gemm.hip")
res = {
    "kernel": "tile_atx (d x T launch of scripts/atx_pmc.py: 2601 tiles, d = 400, T = 10000)",
    "command": "rocprofv3 --pmc <counter> --kernel-trace --kernel-include-regex tile_atx -- python3 scripts/atx_pmc.py (one pass per counter)",
    "kernel_source": "localmd_amd/csrc/tile_gemm.hip", "kernel_source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
    "counters": {k: {"dispatches": v[0], "sum": v[1], "mean": v[1] / v[0], "kernel": v[2]} for k, v in acc.items()},
    "algorithmic_read_bytes_per_launch": 4.0 * 400 * 10000 * 2601,
    "fetch_note": "FETCH_SIZE (KB) x 1024 x 2: gfx950 reports half the bytes of 16 B/lane streaming loads (MI355X_MICROARCH.md, HBM section); L2 -> fabric requests, Infinity Cache hits included",
}
if "FETCH_SIZE" in acc:
    res["fetch_bytes_per_launch"] = acc["FETCH_SIZE"][1] / acc["FETCH_SIZE"][0] * 1024.0 * 2.0
    res["fetch_over_algorithmic"] = res["fetch_bytes_per_launch"] / res["algorithmic_read_bytes_per_launch"]
if "WRITE_SIZE" in acc:
    res["write_bytes_per_launch"] = acc["WRITE_SIZE"][1] / acc["WRITE_SIZE"][0] * 1024.0
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
