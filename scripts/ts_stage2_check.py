"""Stages 1 + 2 of the two-stage reduction against NumPy: eigenvalues of the tridiagonal matrix."""
import os, sys, time, ctypes as C
import numpy as np
import scipy.linalg
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from localmd_amd._lib import Context, ptr
ctx = Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [67, 130, 200, 257, 1030, 4000]
for n in sizes:
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, n + 50)).astype(np.float32) * np.linspace(1, 30, n + 50, dtype=np.float32)[None, :]
    A0 = (X @ X.T).astype(np.float32)
    lda = (n + 3) // 4 * 4
    for rep in range(2):
        A = torch.zeros((n, lda), dtype=torch.float32, device=ctx.device)
        A[:, :n] = torch.from_numpy(np.triu(A0)).to(ctx.device)
        tau = torch.zeros(n, dtype=torch.float32, device=ctx.device)
        d = torch.zeros(n, dtype=torch.float32, device=ctx.device); e = torch.zeros(n, dtype=torch.float32, device=ctx.device)
        flag = C.c_int(0)
        ctx.sync(); t0 = time.perf_counter()
        ctx.call("pmdk_sytrd2", n, ptr(A), lda, ptr(tau), ptr(d), ptr(e), C.byref(flag))
        ctx.sync(); dt = time.perf_counter() - t0
    w0 = np.linalg.eigvalsh(A0.astype(np.float64))
    w1 = scipy.linalg.eigvalsh_tridiagonal(d.cpu().numpy().astype(np.float64), e.cpu().numpy()[:n - 1].astype(np.float64))
    print(f"n = {n}: flag {flag.value}, {dt*1e3:.1f} ms (second call), eig rel err {np.abs(w0 - w1).max() / np.abs(w0).max():.2e}", flush=True)
