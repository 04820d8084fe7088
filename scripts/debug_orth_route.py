"""Which step of the R > frames eigenvector route loses accuracy (fuzz family 'wide', seed 11, case 21)?  From the oracle's
U (float64 CSR) and v_cropped: C = right^T G right formed / diagonalised in several ways; figure = |P^T G64 P - I|.
    python scripts/debug_orth_route.py SEED N CASE"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fuzz import draw_wide_cases
import tests.test_gpu_parity as tp
from tests.util import DeviceSource
from oracle import pmd_oracle as O
import torch
from localmd_amd._lib import Context, ptr

import scipy.linalg
seed, n, case = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c = [c for c in draw_wide_cases(n, seed) if c[0] == case][0]
print(c)
_, T, d1, d2, b1, b2, frames, kw = c
ctx = Context(0)
mov = tp._movie(T, d1, d2, seed=1000 + case)
np.random.seed(7)
ref = O.localmd_decomposition(mov, (b1, b2), frames, rng=DeviceSource(ctx, 123), thresholds=(1.0, 1.7), **kw)
U = ref.u.tocsr().astype(np.float64)
right = np.asarray(ref.diag["v_cropped"], np.float32)
print("U", U.shape, "right", right.shape, flush=True)
G64 = np.asarray((U.T @ U).todense())
C64 = right.astype(np.float64).T @ G64 @ right.astype(np.float64)
lam64 = np.linalg.eigvalsh(C64)
print("C64 eigenvalues: max %.3e, smallest five %s" % (lam64[-1], lam64[:5]))

def figure(name, lam, E):
    lam = np.asarray(lam, np.float64); E = np.asarray(E, np.float64)
    keep = lam != 0
    P = right.astype(np.float64) @ (E[:, keep] * (np.sign(lam[keep]) / np.sqrt(np.abs(lam[keep])))[None, :])
    D = P.T @ G64 @ P - np.eye(P.shape[1])
    order = np.argsort(-np.abs(lam[keep]))
    D = D[np.ix_(order, order)]
    k = len(order)
    dd = np.abs(np.diag(D))
    print(f"{name:46s} |P^T G P - I| all {np.abs(D).max():.2e}, leading half {np.abs(D[:k//2,:k//2]).max():.2e}, directions with |diag - 1| > 0.3: {int((dd > 0.3).sum())}, "
          f"> 0.03: {int((dd > 0.03).sum())}; negative lambdas {int((lam < 0).sum())}, smallest |lambda| {np.sort(np.abs(lam))[:3]}", flush=True)

# the oracle's way: G in float64, G right cast to fp32, product in fp32, LAPACK (double routines on fp32 data -> fp32)
GR32 = (G64 @ right.astype(np.float64)).astype(np.float32)
C32 = right.T @ GR32
lam, E = np.linalg.eigh(C32)
figure("oracle: fp64 G, fp32 product, numpy eigh", lam, E)
lam, E = np.linalg.eigh(C32.astype(np.float64))
figure("same C32, float64 eigh", lam, E)
for drv in ("evd", "ev", "evr"):
    lam, E = scipy.linalg.eigh(C32, driver=drv, check_finite=False)
    figure(f"same C32, LAPACK ssy{drv}", lam, E)
# own eigensolver on the same C32
def own_eig(Cm, mode):
    os.environ["PMD_SYEVD"] = mode
    m = Cm.shape[0]
    ld = (m + 3) // 4 * 4
    A = torch.zeros((m, ld), dtype=torch.float32, device=ctx.device)
    A[:, :m] = torch.from_numpy(np.ascontiguousarray(Cm)).to(ctx.device)
    w = torch.zeros(m, dtype=torch.float32, device=ctx.device); work = torch.zeros(m, dtype=torch.float32, device=ctx.device)
    info = torch.zeros(4, dtype=torch.int32, device=ctx.device)
    ctx.call("pmdk_syevd", m, ptr(A), ld, ptr(w), ptr(work), ptr(info)); ctx.sync()
    os.environ.pop("PMD_SYEVD")
    return w.cpu().numpy(), A[:, :m].cpu().numpy().T
for mode in ("own", "rocsolver", "twostage"):
    lam, E = own_eig(C32, mode)
    figure(f"same C32, pmdk_syevd ({mode})", lam, E)
# symmetrised C32
Cs = 0.5 * (C32 + C32.T)
lam, E = np.linalg.eigh(Cs)
figure("C32 symmetrised, numpy eigh", lam, E)
lam, E = own_eig(Cs, "own")
figure("C32 symmetrised, pmdk_syevd (own)", lam, E)
# fp32 G
G32 = G64.astype(np.float32)
C32b = right.T @ (G32 @ right)
lam, E = np.linalg.eigh(C32b)
figure("fp32 G, fp32 products, numpy eigh", lam, E)
print("asymmetry of C32: %.3e (relative to max %.3e)" % (np.abs(C32 - C32.T).max(), np.abs(C32).max()))
