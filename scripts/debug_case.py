import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd._lib import Context
from localmd_amd.synthetic import make_movie
from oracle import pmd_oracle as O
from tests.util import DeviceSource
Dm.QUIET = True
ctx = Context(0)
mov = make_movie(300, 70, 80, seed=3)
np.random.seed(7)
pmd, diag = localmd_amd.localmd_decomposition(mov, (10, 10), 300, max_components=8, background_rank=3, sim_iters=10, seed=123, return_diagnostics=True, ctx=ctx)
np.random.seed(7)
ref = O.localmd_decomposition(mov, (10,10), 300, max_components=8, background_rank=3, rng=DeviceSource(ctx,123), thresholds=diag["thresholds"])
for name, o in (("hip", pmd), ("oracle", ref)):
    ur = o.u @ o.r
    print(name, "R'", o.v.shape, "orth UR", np.abs(ur.T@ur - np.eye(ur.shape[1])).max(), "orth V", np.abs(o.v@o.v.T - np.eye(o.v.shape[0])).max(), "s", o.s[:3], o.s[-3:])
print("ranks equal", np.array_equal(diag["tile_ranks"], ref.diag["tile_ranks"]))
