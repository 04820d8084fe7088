"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: dispatches, sum and mean of each counter."""
import collections
import csv
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][-60:]
        a = acc[name][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
out = {k: {c: {"dispatches": v[0], "sum": v[1], "mean": v[1] / max(v[0], 1)} for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, d in sorted(out.items(), key=lambda kv: -max(x["sum"] for x in kv[1].values()))[:8]:
    print(k, {c: (v["dispatches"], round(v["mean"], 1)) for c, v in d.items()})
