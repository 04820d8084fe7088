"""Per-tile comparison of the tile bases with the float64 arbiter for one draw of the 'edge' fuzz family.
    python scripts/debug_edge.py SEED N CASE"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fuzz import draw_edge_cases
from tests.util import DeviceSource
from oracle import pmd_oracle as O
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd.synthetic import make_movie
from localmd_amd._lib import Context

seed, n, case = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c = [c for c in draw_edge_cases(n, seed) if c[0] == case][0]
_, T, d1, d2, b1, b2, frames, kw, extra = c
print(c[:8])
ctx = Context(0)
Dm.QUIET = True
mov = make_movie(T, d1, d2, seed=1000 + case)
thr = (1.0, 1.7)
np.random.seed(7)
pmd, diag = localmd_amd.localmd_decomposition(mov, (b1, b2), frames, seed=123, return_diagnostics=True, ctx=ctx, thresholds=thr, **kw)
np.random.seed(7)
ref = O.localmd_decomposition(mov, (b1, b2), frames, rng=DeviceSource(ctx, 123), thresholds=thr, **kw)
np.random.seed(7)
with O.arbiter_precision():
    arb = O.localmd_decomposition(mov, (b1, b2), frames, rng=DeviceSource(ctx, 123), thresholds=thr, dtype="float64", **kw)
print("crop", diag["crop"], "ranks HIP", diag["tile_ranks"], "oracle", ref.diag["tile_ranks"], "arbiter", arb.diag["tile_ranks"])
d = b1 * b2
off = np.concatenate([[0], np.cumsum(diag["tile_ranks"])])
for t in range(len(diag["tile_ranks"])):
    rk = int(diag["tile_ranks"][t])
    ua = np.asarray(arb.diag["tile_u"][t], np.float64).reshape(d, -1, order="F")[:, :rk]
    ur = np.asarray(ref.diag["tile_u"][t], np.float64).reshape(d, -1, order="F")[:, :rk]
    uh = diag["tile_ut"][t, :64, :d].T.astype(np.float64)
    kept = np.nonzero(diag["tile_keep"][t] > 0)[0][:rk]
    uh = uh[:, kept]
    def dist(x, y):
        s = np.sign(np.sum(x * y, axis=0)); s[s == 0] = 1
        return np.linalg.norm(x * s - y, axis=0) / np.linalg.norm(y, axis=0)
    sig = diag["col_sigma"][off[t]:off[t] + rk]
    print(f"tile {t}: rank {rk}, sigma {np.round(sig, 2)}")
    print(f"     HIP    vs arbiter per column {np.array2string(dist(uh, ua), precision=1)}")
    print(f"     oracle vs arbiter per column {np.array2string(dist(ur, ua), precision=1)}")
    # span of all kept columns
    qa, _ = np.linalg.qr(ua); 
    print(f"     span residual: HIP {np.linalg.norm(uh - qa @ (qa.T @ uh)) / np.linalg.norm(uh):.1e}, oracle {np.linalg.norm(ur - qa @ (qa.T @ ur)) / np.linalg.norm(ur):.1e}")

# the assembled sparse U: per column max |diff| (sign aligned) against the arbiter
from tests import parity_metrics as PM
ah, bh = pmd.u.tocsc(), arb.u.tocsc(); rh = ref.u.tocsc()
print("U shapes", ah.shape, bh.shape, "nnz", ah.nnz, bh.nnz, rh.nnz)
for j in range(ah.shape[1]):
    x = np.asarray(ah[:, j].todense()).ravel(); y = np.asarray(bh[:, j].todense()).ravel(); z = np.asarray(rh[:, j].todense()).ravel()
    sx = -1 if x @ y < 0 else 1; sz = -1 if z @ y < 0 else 1
    dh, dr = np.abs(sx * x - y), np.abs(sz * z - y)
    print(f"   column {j}: |U| max {np.abs(y).max():.3f}  HIP diff {dh.max():.2e} at pixel {int(dh.argmax())}  oracle diff {dr.max():.2e}  ratio HIP/arb at that pixel {x[dh.argmax()] / (y[dh.argmax()] + 1e-300):.5f}")
print("block weights equal:", np.array_equal(diag["block_weights"], arb.diag["block_weights"]), "origins", diag["origins"][:4].tolist())
