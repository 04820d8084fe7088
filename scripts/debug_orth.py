import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd._lib import Context
from localmd_amd.synthetic import make_movie
Dm.QUIET = True
ctx = Context(0)
mov = make_movie(2000, 256, 256, seed=0)
for env in ({}, {"PMD_SYEVD": "rocsolver"}):
    os.environ.pop("PMD_SYEVD", None)
    os.environ.update(env)
    np.random.seed(7)
    pmd, diag = localmd_amd.localmd_decomposition(mov, (20, 20), 2000, seed=123, sim_iters=50, max_components=8, return_diagnostics=True, ctx=ctx)
    v = pmd.v.astype(np.float64); s = pmd.s.astype(np.float64)
    E = v @ v.T - np.eye(len(s))
    w = s / s[0]
    Ew = np.abs(E) * np.outer(w, w)
    Ew[-1, :] = 0; Ew[:, -1] = 0
    i, j = np.unravel_index(Ew.argmax(), Ew.shape)
    print(env, "worst weighted", Ew[i, j], "at", i, j, "E", E[i, j], "s_i/s1", w[i], "s_j/s1", w[j])
    print("  row norms^2 -1: first 5", np.diag(E)[:5], " last 8", np.diag(E)[-8:])
    print("  s last 8 / s1", w[-8:])
    top = np.argsort(-Ew.ravel())[:6]
    for t in top:
        a, b = np.unravel_index(t, Ew.shape)
        print("   ", a, b, "E", E[a, b], "w", w[a], w[b], "weighted", Ew[a, b])
