"""Stage 1 of the two-stage reduction (dense -> band) against NumPy: eigenvalues of the band matrix."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from localmd_amd._lib import Context, ptr
ctx = Context(0)
for n in (67, 130, 200, 257, 1030, 9999):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, n + 50)).astype(np.float32) * np.linspace(1, 30, n + 50, dtype=np.float32)[None, :]
    A0 = (X @ X.T).astype(np.float32)
    lda = (n + 3) // 4 * 4
    A = torch.zeros((n, lda), dtype=torch.float32, device=ctx.device)
    A[:, :n] = torch.from_numpy(np.triu(A0)).to(ctx.device)      # only positions r >= c valid on entry
    tau = torch.zeros(n, dtype=torch.float32, device=ctx.device)
    flag = C.c_int(0)
    t0 = time.perf_counter()
    ctx.call("pmdk_sy2sb", n, ptr(A), lda, ptr(tau), C.byref(flag))
    ctx.sync()
    dt = time.perf_counter() - t0
    Ah = A.cpu().numpy()[:, :n]
    B = np.zeros((n, n))
    for c in range(n):
        hi = min(n, c + 65)
        B[c, c:hi] = Ah[c, c:hi]
    B = B + np.triu(B, 1).T
    w0 = np.linalg.eigvalsh(A0.astype(np.float64)); w1 = np.linalg.eigvalsh(B)
    print(f"n = {n}: flag {flag.value}, {dt*1e3:.1f} ms, eig rel err {np.abs(w0 - w1).max() / np.abs(w0).max():.2e}, tau range [{float(tau.min()):.3f}, {float(tau.max()):.3f}]", flush=True)
