"""Backward error of the symmetric eigensolvers on a graded Gram matrix (the final-SVD situation): NumPy/LAPACK fp32,
rocSOLVER ssyevd and the library's own path (pmdk_syevd), against an fp64 reference."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localmd_amd import _lib

def report(name, G64, V, w, E):
    # E columns = eigenvectors, w eigenvalues (any order)
    E = E.astype(np.float64); w = w.astype(np.float64)
    res = np.abs(G64 @ E - E * w[None, :]).max() / np.abs(G64).max()
    orthE = np.abs(E.T @ E - np.eye(E.shape[1])).max()
    s = np.sqrt(np.maximum(w, 1e-300))
    Vt = (E.T @ V.astype(np.float64)) / s[:, None]
    orthV = np.abs(Vt @ Vt.T - np.eye(len(w))).max()
    wt = np.sort(np.linalg.eigvalsh(G64))
    werr = np.abs(np.sort(w) - wt).max() / wt.max()
    print(f"{name:28s} residual/|G| {res:.2e}  |E^T E - I| {orthE:.2e}  eigenvalue abs err/lambda_max {werr:.2e}  |Vt Vt^T - I| {orthV:.2e}")

for n, T, ratio in ((89, 314, 66.0), (300, 1000, 100.0), (1200, 3000, 100.0)):
    rng = np.random.default_rng(n)
    q1, _ = np.linalg.qr(rng.standard_normal((n, n)))
    q2, _ = np.linalg.qr(rng.standard_normal((T, n)))
    sv = np.geomspace(ratio, 1.0, n)
    V = ((q1 * sv) @ q2.T).astype(np.float32)
    G32 = V @ V.T
    G64 = G32.astype(np.float64)   # the matrix every solver is given
    print(f"n={n} T={T} sigma ratio {ratio}")
    w, E = np.linalg.eigh(G32)
    report("numpy.linalg (double inside)", G64, V, w, E)
    import scipy.linalg
    w, E = scipy.linalg.eigh(G32, driver="evd", check_finite=False)
    report("scipy float32 ssyevd", G64, V, w, E)
    ctx = _lib.Context(0); P = _lib.ptr
    for mode in ("rocsolver", "own"):
        os.environ["PMD_SYEVD"] = mode
        ld = (n + 3) // 4 * 4
        buf = np.zeros((n, ld), np.float32); buf[:, :n] = G32
        A = torch.from_numpy(buf).cuda()
        wd = torch.empty(n, device="cuda"); work = torch.empty(n, device="cuda"); info = torch.zeros(4, dtype=torch.int32, device="cuda")
        ctx.call("pmdk_syevd", n, P(A), ld, P(wd), P(work), P(info)); ctx.sync()
        Eg = A.cpu().numpy()[:, :n].T   # memory row j = eigenvector j
        report("GPU " + mode, G64, V, wd.cpu().numpy(), Eg)
    ctx.close()
