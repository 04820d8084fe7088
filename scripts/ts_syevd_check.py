"""pmdk_syevd through the two-stage route (PMD_SYEVD=twostage) against NumPy: eigenvalues, residual, orthonormality,
and the per-stage times of the library's profiler."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from localmd_amd._lib import Context, ptr
ctx = Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [300, 1030, 4000]
for n in sizes:
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, n + 50)).astype(np.float32) * np.linspace(1, 30, n + 50, dtype=np.float32)[None, :]
    A0 = (X @ X.T).astype(np.float32)
    lda = (n + 3) // 4 * 4
    res = {}
    for mode in ("own", "twostage"):
        os.environ["PMD_SYEVD"] = mode
        for rep in range(2):
            A = torch.zeros((n, lda), dtype=torch.float32, device=ctx.device)
            A[:, :n] = torch.from_numpy(A0).to(ctx.device)
            w = torch.zeros(n, dtype=torch.float32, device=ctx.device)
            work = torch.zeros(n, dtype=torch.float32, device=ctx.device)
            info = torch.zeros(4, dtype=torch.int32, device=ctx.device)
            ctx.sync(); t0 = time.perf_counter()
            ctx.call("pmdk_syevd", n, ptr(A), lda, ptr(w), ptr(work), ptr(info))
            ctx.sync(); dt = time.perf_counter() - t0
        wv = w.cpu().numpy().astype(np.float64)
        E = A[:, :n].double()
        Ad = torch.from_numpy(A0).to(ctx.device).double()
        resid = float(((E @ Ad) - torch.from_numpy(wv).to(ctx.device)[:, None] * E).abs().max()) / wv.max()
        orth = float((E @ E.T - torch.eye(n, dtype=torch.float64, device=ctx.device)).abs().max())
        w0 = np.linalg.eigvalsh(A0.astype(np.float64)) if n <= 4000 else None
        werr = np.abs(w0 - wv).max() / w0.max() if w0 is not None else float("nan")
        print(f"n = {n} {mode}: {dt*1e3:.1f} ms, info {int(info[0])}, eig err {werr:.2e}, residual {resid:.2e}, orth {orth:.2e}", flush=True)
