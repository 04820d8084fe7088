"""Idle gaps of the GPU inside one decomposition, from a rocprofv3 --kernel-trace CSV:
    python scripts/gap_trace.py <kernel_trace.csv> [min_gap_ms]   [index]   (default index -2: the last timed decomposition of a bench.py run)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# decompositions start with the statistics kernel
starts = [i for i, e in enumerate(ev) if "stats_chunk" in e[2]]
first = [s for k, s in enumerate(starts) if k == 0 or ev[s][0] - ev[starts[k - 1]][0] > 50e6]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2      # default: the last TIMED decomposition (bench.py ends with an instrumented one)
lo = first[which]
hi = first[which + 1] if which + 1 < 0 and which + 1 + len(first) < len(first) else len(ev)
if which != -1:
    hi = first[which + 1 if which >= 0 else len(first) + which + 1]
seg = ev[lo:hi]
t0 = seg[0][0]
busy_end = seg[0][1]
print(f"decomposition starts at event {lo}, {len(seg)} kernels, span {(max(e[1] for e in seg) - t0) / 1e6:.1f} ms")
tot_gap = 0.0
for s, e, name in seg[1:]:
    if s > busy_end:
        gap = (s - busy_end) / 1e6
        tot_gap += gap
        if gap >= min_gap:
            print(f"  t = {(busy_end - t0) / 1e6:8.1f} ms: idle {gap:6.1f} ms before {name[:70]}")
    busy_end = max(busy_end, e)
print(f"total idle {tot_gap:.1f} ms")
