"""Phase timeline of the workgroups of one symv launch (column 32 of an n = 9999 tridiagonalisation).
Needs the trace build of the library (bash scripts/build_trace_lib.sh [column]):

Stamps (100 MHz wall clock) per workgroup: 0 entry, 1 loads issued, 2 scalars ready (barrier), 3 tile products done
(= loads arrived), 4 cross-lane reduce + barrier done, 5 partial sums stored."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PMD_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "localmd_amd", "libpmd_hip_trace.so"))
from localmd_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9999
ctx = _lib.Context(0)
P = _lib.ptr
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn((n, n + 2000), device="cuda", generator=g)
ld = (n + 3) // 4 * 4
S = torch.zeros((n, ld), device="cuda")
S[:, :n] = X @ X.T
del X
d = torch.zeros(n, device="cuda"); e = torch.zeros(n, device="cuda"); tau = torch.zeros(n, device="cuda")
for rep in range(2):
    A = S.clone()
    ctx.call("pmdk_sytrd", n, P(A), ld, P(d), P(e), P(tau), 1)
    ctx.sync()
cnt = 4096 * 8
buf = (C.c_ulonglong * cnt)()
ctx.lib.pmdk_symv_trace.argtypes = [C.c_void_p, C.c_int]
rc = ctx.lib.pmdk_symv_trace(buf, cnt)
assert rc == 0, rc
full = np.array(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
t = full[:, :6]
live = t[:, 5] > 0          # tile workgroups that ran to the end (invalid tiles / dot workgroups leave early)
wgid = np.nonzero(live)[0]
t = t[live]
extra = full[live][:, 6:8]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
print("workgroups traced:", len(us), " launch span: %.1f us" % (us[:, 5].max()))
names = ["entry->loads issued", "->scalars ready", "->products done", "->reduce done", "->stored"]
for k in range(5):
    dlt = us[:, k + 1] - us[:, k]
    print("  %-22s mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f us" % (names[k], dlt.mean(), *np.percentile(dlt, [10, 50, 90])))
life = us[:, 5] - us[:, 0]
print("  lifetime               mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f us" % (life.mean(), *np.percentile(life, [10, 50, 90])))
# how many workgroups are between 'loads issued' and 'products done' (= have loads in flight) over time
grid = np.linspace(0, us[:, 5].max(), 60)
inflight = [(np.sum((us[:, 1] <= x) & (us[:, 3] > x)), np.sum((us[:, 0] <= x) & (us[:, 5] > x))) for x in grid]
print("  time(us): workgroups with loads in flight / resident")
print("  " + "  ".join("%.0f:%d/%d" % (x, a, b) for x, (a, b) in zip(grid[::3], inflight[::3])))
order = np.argsort(us[:, 0])
print("  entry times of the first 12 and last 12 workgroups:", np.round(us[order[:12], 0], 1), np.round(us[order[-12:], 0], 1))

for lo, hi in ((0, 256), (256, 512), (512, 768), (768, 1024), (1024, 4096)):
    m = (wgid >= lo) & (wgid < hi)
    if m.any():
        print("  workgroups %4d..%4d: entry %.1f..%.1f us, entry->issued %.2f, issued->arrived %.2f, rest %.2f, end %.1f..%.1f" % (
            lo, hi, us[m, 0].min(), us[m, 0].max(), (us[m, 1] - us[m, 0]).mean(), (us[m, 2] - us[m, 1]).mean(),
            (us[m, 5] - us[m, 2]).mean(), us[m, 5].min(), us[m, 5].max()))

pre = (t[:, 0] - extra[:, 0]) / 100.0
mid = (extra[:, 1] - t[:, 0]) / 100.0
iss = (t[:, 1] - extra[:, 1]) / 100.0
for lo, hi in ((0, 256), (256, 512), (1024, 4096)):
    m = (wgid >= lo) & (wgid < hi)
    print("  workgroups %4d..%4d: first instruction -> stamp 0: %.2f us, -> before tile loads %.2f us, -> tile loads issued %.2f us" % (
        lo, hi, pre[m].mean(), mid[m].mean(), iss[m].mean()))

# advance kernel of the same column: 0 entry, 1 partial sums / panel rows / dot partials summed, 2 SP summed,
# 3 first barrier passed, 4 corrections applied + second barrier, 5 end
cnt = 1024 * 8
buf2 = (C.c_ulonglong * cnt)()
ctx.lib.pmdk_adv_trace.argtypes = [C.c_void_p, C.c_int]
assert ctx.lib.pmdk_adv_trace(buf2, cnt) == 0
ta_full = np.array(buf2, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
ta_full = ta_full[ta_full[:, 5] > 0]
ta = ta_full[:, :6]
ua = (ta - ta[:, 0].min()) / 100.0
print("advance workgroups traced:", len(ua), " span %.2f us" % ua[:, 5].max())
for k, nm in enumerate(["entry->partials summed", "->SP summed", "->barrier 1", "->barrier 2", "->end"]):
    dlt = ua[:, k + 1] - ua[:, k]
    print("  %-24s mean %.2f  p10 %.2f  p90 %.2f us" % (nm, dlt.mean(), *np.percentile(dlt, [10, 90])))
print("  entry spread %.2f us" % (ua[:, 0].max()))

if ta_full[:, 6].max() > 0:
    print("  entry -> all loads issued (stamp 6): mean %.2f us" % ((ta_full[:, 6] - ta_full[:, 0]).mean() / 100.0))
