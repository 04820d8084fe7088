"""A/B timing of the tridiagonalisation: PMD_SYMV_PERSIST=0 against the default, n from argv (default 9999)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localmd_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9999
modes = sys.argv[2:] or ["0", "256", "512", "128"]
ctx = _lib.Context(0)
P = _lib.ptr
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn((n, n + 2000), device="cuda", generator=g)
ld = (n + 3) // 4 * 4
S = torch.zeros((n, ld), device="cuda")
S[:, :n] = X @ X.T
del X
d = torch.zeros(n, device="cuda"); e = torch.zeros(n, device="cuda"); tau = torch.zeros(n, device="cuda")
for mode in modes * 2:
    # "512s" = 512 persistent workgroups, single-buffered (PMD_SYMV_PERSIST_DB=0)
    os.environ["PMD_SYMV_PERSIST"] = mode.rstrip("s")
    os.environ["PMD_SYMV_PERSIST_DB"] = "0" if mode.endswith("s") else "1"
    A = S.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.call("pmdk_sytrd", n, P(A), ld, P(d), P(e), P(tau), 1)
    ctx.sync()
    print(f"n={n} PMD_SYMV_PERSIST={mode}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
