"""NumPy statement (float64) of the two-stage tridiagonalisation that localmd_amd/csrc/sytrd2.hip implements: test
infrastructure only.  It pins the three structural claims the kernels rely on:

  * panel_qr_hr: CholeskyQR2 + Householder reconstruction gives a unit lower trapezoidal V and an upper triangular T with
    (I - V T V^T)^T P = [S R; 0] and tau_i = T_ii = 2 / |v_i|^2 (so the compact-WY identity of sytrd.hip's apply_q holds);
  * bulge chasing in band storage AB[j][d] = B(j + d, j) with the tasks run in wavefront order (task (s, k) at step
    2 s + k, every task of a step independent of the others);
  * the blocked order of the back-transformation (sweep groups descending, task index ascending, sweeps descending inside).

The reference computes this eigendecomposition with jnp.linalg.svd(..., hermitian=True) (decomposition.py:984, :1090).
"""
import numpy as np


def house(x):
    """LAPACK-style reflector: H = I - tau v v^T, v[0] = 1, H x = beta e1."""
    alpha = x[0]
    xn = np.linalg.norm(x[1:])
    if xn == 0:
        return np.r_[1.0, np.zeros(len(x) - 1)], 0.0, alpha
    beta = -np.sign(alpha if alpha != 0 else 1.0) * np.hypot(alpha, xn)
    tau = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1.0
    return v, tau, beta


def panel_qr_hr(P):
    """P (m x b), m >= b -> V (m x b), T (b x b upper), Rs = S R with (I - V T V^T)^T P = [Rs; 0]."""
    m, b = P.shape
    R1 = np.linalg.cholesky(P.T @ P).T
    Q = np.linalg.solve(R1.T, P.T).T
    R2 = np.linalg.cholesky(Q.T @ Q).T
    Q = np.linalg.solve(R2.T, Q.T).T
    R = R2 @ R1
    A = Q.copy()
    S = np.zeros(b)
    L = np.zeros((m, b))
    U = np.zeros((b, b))
    for i in range(b):
        S[i] = -1.0 if A[i, i] >= 0 else 1.0
        A[i, i] -= S[i]
        U[i, i:] = A[i, i:]
        L[i:, i] = A[i:, i] / U[i, i]
        A[i + 1:, i + 1:] -= np.outer(L[i + 1:, i], U[i, i + 1:])
        A[i + 1:, i] = 0
    T = -U @ np.diag(S) @ np.linalg.inv(L[:b, :b]).T
    return L, T, np.diag(S) @ R


def stage1(A, b):
    """dense symmetric -> band of half bandwidth b (dense storage) and the block reflectors [(row0, V, T)]."""
    n = A.shape[0]
    A = A.copy()
    refl = []
    for j0 in range(0, n - b - 1, b):
        r0 = j0 + b
        m = n - r0
        P = A[r0:, j0:j0 + b]
        if m < b:   # the last, narrow panel: any orthogonal factor will do here
            Qf, _ = np.linalg.qr(P, mode="complete")
            A[r0:, :] = Qf.T @ A[r0:, :]
            A[:, r0:] = A[:, r0:] @ Qf
            refl.append((r0, Qf, None))
            continue
        V, T, Rs = panel_qr_hr(P)
        A22 = A[r0:, r0:]
        Y = A22 @ V @ T
        W = Y - 0.5 * V @ (T.T @ (V.T @ Y))
        A[r0:, r0:] = A22 - V @ W.T - W @ V.T
        A[r0:, j0:j0 + b] = 0
        A[r0:r0 + b, j0:j0 + b] = Rs
        A[j0:j0 + b, r0:] = A[r0:, j0:j0 + b].T
        refl.append((r0, V, T))
    return A, refl


def to_band(B, b):
    n = B.shape[0]
    AB = np.zeros((n, 2 * b))
    for j in range(n):
        for d in range(0, min(b, n - 1 - j) + 1):
            AB[j, d] = B[j + d, j]
    return AB


def task(AB, n, b, s, k, vprev, tprev):
    """Task (s, k) of the bulge chase on band storage.  Returns (v zero padded to b, tau) or None if there is no such task."""
    r0 = s + 1 + k * b
    if r0 >= n:
        return None
    L = min(b, n - r0)
    if k == 0:
        if L < 2:
            return None
        v, tau, beta = house(AB[s, 1:1 + L].copy())
        AB[s, 1] = beta
        AB[s, 2:1 + L] = 0
    else:
        c0 = r0 - b
        Bm = np.array([[AB[c0 + j, b + i - j] for j in range(b)] for i in range(L)])
        Bm -= tprev * np.outer(Bm @ vprev, vprev)
        if L >= 2:
            v, tau, beta = house(Bm[:, 0].copy())
            Bm[:, 0] = 0
            Bm[0, 0] = beta
            Bm[:, 1:] -= tau * np.outer(v, v @ Bm[:, 1:])
        else:
            v, tau = np.array([1.0]), 0.0
        for i in range(L):
            for j in range(b):
                AB[c0 + j, b + i - j] = Bm[i, j]
    if tau != 0.0:
        D = np.array([[AB[r0 + j, i - j] if i >= j else AB[r0 + i, j - i] for j in range(L)] for i in range(L)])
        p = tau * (D @ v)
        q = p - 0.5 * tau * (p @ v) * v
        D -= np.outer(v, q) + np.outer(q, v)
        for i in range(L):
            for j in range(i + 1):
                AB[r0 + j, i - j] = D[i, j]
    vv = np.zeros(b)
    vv[:L] = v
    return vv, tau


def stage2_wavefront(AB, n, b):
    """All tasks in step order t = 2 s + k (inside one step the order of the sweeps must not matter: they run descending
    here, the kernel runs them concurrently).  Returns {(s, k): (v, tau)}; AB ends tridiagonal (columns 0 and 1)."""
    prev, V2 = {}, {}
    t_max = 2 * (n - 3) + -(-(n - 1) // b) + 2
    for t in range(t_max + 1):
        for s in range(n - 3, -1, -1):
            k = t - 2 * s
            if k < 0:
                continue
            vp, tp = prev.get(s, (None, 0.0))
            out = task(AB, n, b, s, k, vp, tp)
            if out is not None:
                prev[s] = out
                V2[(s, k)] = out
    return V2


def apply_q2_blocked(V2, Z, n, b, group):
    """Z <- Q2 Z in the order of apply_q2_kernel."""
    Y = Z.copy()
    kmax = max(k for (_, k) in V2) + 1
    for S in range((n - 2 + group - 1) // group - 1, -1, -1):
        for k in range(kmax):
            for s in range(min(n - 2, (S + 1) * group) - 1, S * group - 1, -1):
                if (s, k) not in V2:
                    continue
                v, tau = V2[(s, k)]
                r0 = s + 1 + k * b
                L = min(b, n - r0)
                Y[r0:r0 + L, :] -= tau * np.outer(v[:L], v[:L] @ Y[r0:r0 + L, :])
    return Y


def apply_q1(refl, E):
    E = E.copy()
    for (r0, V, T) in reversed(refl):
        if T is None:
            E[r0:, :] = V @ E[r0:, :]
        else:
            E[r0:, :] -= V @ (T @ (V.T @ E[r0:, :]))
    return E
