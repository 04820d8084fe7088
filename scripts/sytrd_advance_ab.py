"""Whole tridiagonalisation (n = 9999): A/B of the advance forms (arguments "old" / "new" = PMD_SYTRD_ADVANCE)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localmd_amd import _lib  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9999
ctx = _lib.Context(0); P = _lib.ptr
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn((n, n + 2000), device="cuda", generator=g)
ld = (n + 3) // 4 * 4
S = torch.zeros((n, ld), device="cuda"); S[:, :n] = X @ X.T; del X
d = torch.zeros(n, device="cuda"); e = torch.zeros(n, device="cuda"); tau = torch.zeros(n, device="cuda")
for dl in (sys.argv[2:] or ["old", "new", "old", "new"]):
    os.environ["PMD_SYTRD_ADVANCE"] = dl
    best = 1e9
    for rep in range(2):
        A = S.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.call("pmdk_sytrd", n, P(A), ld, P(d), P(e), P(tau), 1); ctx.sync()
        best = min(best, time.perf_counter() - t0)
    print(f"n={n} PMD_SYTRD_ADVANCE={dl}: {1e3 * best:.1f} ms", flush=True)
