import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd._lib import Context
from localmd_amd.synthetic import make_movie
from oracle import pmd_oracle as O
from tests.util import DeviceSource, sign_align
Dm.QUIET = True
ctx = Context(0)
mov = make_movie(600, 40, 50, seed=1)
np.random.seed(7)
pmd, diag = localmd_amd.localmd_decomposition(mov, (20, 20), 600, max_components=6, background_rank=2, sim_iters=20, seed=123, return_diagnostics=True, ctx=ctx)
np.random.seed(7)
ref = O.localmd_decomposition(mov, (20, 20), 600, max_components=6, background_rank=2, rng=DeviceSource(ctx, 123), thresholds=diag["thresholds"])
print("ranks", diag["tile_ranks"], ref.diag["tile_ranks"])
n = min(len(pmd.s), len(ref.s))
gaps = np.minimum(np.abs(np.diff(ref.s[:n], prepend=np.inf)), np.abs(np.diff(ref.s[:n], append=0))) / ref.s[:n]
va = sign_align(pmd.v[:n], ref.v[:n], axis=1)
for c in range(n):
    e = np.linalg.norm(va[c] - ref.v[c]) / np.linalg.norm(ref.v[c])
    print(c, f"s={ref.s[c]:.3f} ds={abs(pmd.s[c]-ref.s[c])/ref.s[c]:.2e} gap={gaps[c]:.3f} vt_err={e:.2e}")
# per tile U comparison
for t, (a, b) in enumerate(zip(diag["tile_ut"], ref.diag["tile_u"])):
    rk = b.shape[2]
    ub = b.reshape(-1, rk, order="F")
    ua = a[:rk, :400].T
    ua = sign_align(ua, ub)
    print("tile", t, [f"{np.linalg.norm(ua[:,c]-ub[:,c]):.1e}" for c in range(rk)])
