"""Timing probe: tridiagonalisation and full eigensolver at n = 10^4 (rocSOLVER vs own kernels)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localmd_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = _lib.Context(0)
P = _lib.ptr
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn((n, n + 2000), device="cuda", generator=g)
S = (X @ X.T).contiguous()
del X
d = torch.zeros(n, device="cuda"); e = torch.zeros(n, device="cuda"); tau = torch.zeros(n, device="cuda")
for impl in (1, 0, 1):
    A = S.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.call("pmdk_sytrd", n, P(A), n, P(d), P(e), P(tau), impl)
    ctx.sync()
    print("sytrd impl", impl, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), "d[:3]", d[:3].tolist(), "e[:3]", e[:3].tolist(), flush=True)
w = torch.zeros(n, device="cuda"); work = torch.zeros(n, device="cuda"); info = torch.zeros(4, dtype=torch.int32, device="cuda")
res = {}
for mode in ("own", "rocsolver", "own"):
    os.environ["PMD_SYEVD"] = mode
    A = S.clone()
    ctx.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.call("pmdk_syevd", n, P(A), n, P(w), P(work), P(info))
    ctx.sync()
    dt = (time.perf_counter() - t0) * 1e3
    prof = ctx.profile_summary()
    ctx.profile_enable(False)
    E = A
    resid = (E @ S - w[:, None] * E).abs().max().item() / w.abs().max().item()
    orth = (E @ E.T - torch.eye(n, device="cuda")).abs().max().item()
    print("syevd", mode, "%.1f ms" % dt, {k: round(v[0], 1) for k, v in prof.items()}, "resid %.2e orth %.2e" % (resid, orth), flush=True)
    res[mode] = w.clone()
print("eigenvalue diff own vs rocsolver (rel to max):", ((res["own"] - res["rocsolver"]).abs().max() / res["own"].abs().max()).item())
