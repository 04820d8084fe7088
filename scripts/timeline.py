"""Compact timeline of one decomposition from a rocprofv3 --kernel-trace CSV: runs of consecutive launches of the same
kernel (per queue) with their start offset, wall span, summed kernel time and launch count.
    python scripts/timeline.py <kernel_trace.csv> [index=-2] [from_ms=0] [to_ms=inf] [min_ms=0.3]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
t_from = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
t_to = float(sys.argv[4]) if len(sys.argv) > 4 else 1e30
min_ms = float(sys.argv[5]) if len(sys.argv) > 5 else 0.3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
starts = [i for i, e in enumerate(ev) if "stats_chunk" in e[2]]
first = [s for k, s in enumerate(starts) if k == 0 or ev[s][0] - ev[starts[k - 1]][0] > 50e6]
lo = first[which]
nxt = which + 1
hi = len(ev) if nxt == 0 or nxt >= len(first) else first[nxt]
seg = ev[lo:hi]
t0 = seg[0][0]


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    if name.startswith("Cijk_"):
        parts = name.split("_")
        mt = [p for p in parts if p.startswith("MT")]
        return "tensile " + "_".join(parts[1:4]) + " " + (mt[0] if mt else "")
    return name.split("(")[0][:48]


runs = {}   # queue -> current run
out = []
for s, e, name, q in seg:
    key = short(name)
    cur = runs.get(q)
    if cur is not None and cur["key"] == key and s - cur["end"] < 2e6:
        cur["end"] = max(cur["end"], e)
        cur["busy"] += e - s
        cur["n"] += 1
    else:
        if cur is not None:
            out.append(cur)
        runs[q] = dict(key=key, q=q, start=s, end=e, busy=e - s, n=1)
out.extend(runs.values())
out.sort(key=lambda r: r["start"])
print(f"decomposition {which}: {len(seg)} launches, span {(max(e[1] for e in seg) - t0) / 1e6:.1f} ms")
for r in out:
    a, b = (r["start"] - t0) / 1e6, (r["end"] - t0) / 1e6
    if b < t_from or a > t_to or (r["busy"] / 1e6 < min_ms and b - a < min_ms):
        continue
    print(f"  {a:8.2f} - {b:8.2f} ms  q{r['q']:>3}  busy {r['busy'] / 1e6:8.2f} ms  x{r['n']:<5d} {r['key']}")

# per-queue totals inside the window (every launch, also those below min_ms)
agg = {}
for s, e, name, q in seg:
    a, b = (s - t0) / 1e6, (e - t0) / 1e6
    if b < t_from or a > t_to:
        continue
    k = (q, short(name))
    r = agg.setdefault(k, [0, 0.0, a, b])
    r[0] += 1
    r[1] += (e - s) / 1e6
    r[2] = min(r[2], a)
    r[3] = max(r[3], b)
print("per queue and kernel inside the window:")
for (q, k), r in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[1][2])):
    if r[1] >= min_ms / 4:
        print(f"  q{q:>3}  {r[2]:8.2f} - {r[3]:8.2f} ms  busy {r[1]:8.2f} ms  x{r[0]:<5d} {k}")
