"""Turns the counter CSV of
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex sytrd_symv -d <dir> -- python3 scripts/sytrd_prof.py 10000 1 1
into the JSON bench.py reads for `roofline.traffic` (fabric-side bytes per launch of the dominant kernel, with the gfx950
correction of MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request of a wide streaming read -> x2), together with
the SHA-256 of the kernel's source file, so that bench.py can tell when the figure has gone stale.
    python scripts/pmc_sytrd.py <counter_collection.csv> <n> <out.json>"""
import csv, hashlib, json, os, sys

path, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        if "sytrd_symv" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
rows.sort()
vals = [v for _, v in rows][: n - 1]           # one launch per column j = 0 .. n - 2 (first repetition)
kb_to_bytes = 1024.0 * 2.0                     # reported in KB; x2 on gfx950 for 16 B/lane streaming loads
traffic = [v * kb_to_bytes for v in vals]
alg = [4.0 * (n - j - 1) * (n - j) / 2 for j in range(len(vals))]
sample = list(range(32, len(vals), 64))
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "localmd_amd", "csrc", "sytrd.hip")
res = {
    "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex sytrd_symv -- python3 scripts/sytrd_prof.py %d 1 1" % n,
    "kernel": "sytrd_symv_kernel<512>", "matrix_order": n, "dispatches": len(vals),
    "kernel_source": "localmd_amd/csrc/sytrd.hip", "kernel_source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
    "fetch_size_unit": "KB as reported; x2 on gfx950 for 16 B/lane streaming loads (MI355X_MICROARCH.md, HBM section)",
    "traffic_bytes_all_launches": sum(traffic), "algorithmic_bytes_all_launches": sum(alg),
    "traffic_over_algorithmic_all": sum(traffic) / sum(alg),
    "sample": "launches j = 32, 96, 160, ... (the launches bench.py times)",
    "traffic_bytes_per_launch_sample_mean": sum(traffic[j] for j in sample) / len(sample),
    "algorithmic_bytes_per_launch_sample_mean": sum(alg[j] for j in sample) / len(sample),
    "note": "L2 -> fabric reads (Infinity Cache hits included); the trailing triangle is read once per launch, the rest is the partial sums, the reflector and the panel dot products",
}
res["traffic_over_algorithmic_sample"] = res["traffic_bytes_per_launch_sample_mean"] / res["algorithmic_bytes_per_launch_sample_mean"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
