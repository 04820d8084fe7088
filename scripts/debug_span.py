"""Why does the passing-span figure of tests/test_gpu_fuzz.py differ between HIP and the arbiter in some draws with
background_rank = 0 although U agrees?   python scripts/debug_span.py SEED N CASE"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fuzz import draw_wide_cases
from tests import parity_metrics as PM
import tests.test_gpu_parity as tp
from tests.util import DeviceSource
from oracle import pmd_oracle as O
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd._lib import Context

seed, n, case = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c = [c for c in draw_wide_cases(n, seed) if c[0] == case][0]
_, T, d1, d2, b1, b2, frames, kw = c
print(c)
ctx = Context(0)
Dm.QUIET = True
mov = tp._movie(T, d1, d2, seed=1000 + case)
thr = (1.0, 1.7)
np.random.seed(7)
pmd, diag = localmd_amd.localmd_decomposition(mov, (b1, b2), frames, seed=123, return_diagnostics=True, ctx=ctx, thresholds=thr, **kw)
np.random.seed(7)
with O.arbiter_precision():
    arb = O.localmd_decomposition(mov, (b1, b2), frames, rng=DeviceSource(ctx, 123), thresholds=thr, dtype="float64", **kw)
np.random.seed(7)
ref = O.localmd_decomposition(mov, (b1, b2), frames, rng=DeviceSource(ctx, 123), thresholds=thr, **kw)
ntc = diag["n_tile_cols"]
hp, hg = PM.hip_cols(diag); ap, ag = PM.oracle_cols(arb); rp_, rg = PM.oracle_cols(ref)
print("columns of U:", pmd.u.shape[1], "tile columns", ntc, "passing HIP/arb/ref", hp.sum(), ap.sum(), rp_.sum())
order = kw.get("order", "F")
y = ((mov.astype(np.float64) - arb.mean_img[None]) / arb.std_img[None]).reshape(T, -1, order=order).T

def span_s(res, cols):
    u = np.asarray(res.u.tocsc()[:, cols].todense(), dtype=np.float64)
    nz = np.linalg.norm(u, axis=0) > 0
    q, r = np.linalg.qr(u[:, nz])
    return np.linalg.svd(q.T @ y, compute_uv=False), np.abs(np.diag(r)).min() / np.abs(np.diag(r)).max(), int((~nz).sum())

both = hp & ap
stable = both & (np.minimum(hg, ag) > PM.GAP)
for name, mask in (("passing columns + trailing columns (the test's set)", np.r_[both, np.ones(pmd.u.shape[1] - ntc, bool)]),
                   ("passing tile columns only", np.r_[both, np.zeros(pmd.u.shape[1] - ntc, bool)]),
                   ("stable passing tile columns only", np.r_[stable, np.zeros(pmd.u.shape[1] - ntc, bool)])):
    cols = np.nonzero(mask)[0]
    sa, ca, za = span_s(pmd, cols); sb, cb, zb = span_s(arb, cols); sr, cr, zr = span_s(ref, cols)
    k = max(1, len(sa) // 4)
    print(f"{name}: {len(cols)} columns ({za} all-zero), min/max |diag R| {ca:.2e}; HIP vs arbiter top {np.abs(sa-sb)[:k].max()/1:.3e} abs "
          f"rel top {(np.abs(sa-sb)/sb)[:k].max():.2e} all {(np.abs(sa-sb)/sb).max():.2e}; oracle fp32 vs arbiter rel top {(np.abs(sr-sb)/sb)[:k].max():.2e} all {(np.abs(sr-sb)/sb).max():.2e}")
# per-column differences of U against the arbiter
def coldiff(res):
    a, b = res.u.tocsc(), arb.u.tocsc()
    out = np.zeros(a.shape[1])
    for j in range(a.shape[1]):
        x, z = np.asarray(a[:, j].todense()).ravel(), np.asarray(b[:, j].todense()).ravel()
        if x @ z < 0: x = -x
        out[j] = np.linalg.norm(x - z) / max(np.linalg.norm(z), 1e-300)
    return out
dh, dr = coldiff(pmd), coldiff(ref)
print("relative column distance to the arbiter, passing columns: HIP max %.2e (at %d), oracle fp32 max %.2e" % (dh[:ntc][both].max(), int(np.argmax(np.where(both, dh[:ntc], 0))), dr[:ntc][both].max()))
worst = np.argsort(-np.where(both, dh[:ntc], 0))[:8]
tile, idx, passed, sigma, gap = PM.tile_column_table(arb.diag["tile_ranks"], [np.concatenate([(d[0]["good"] if len(d) == 1 else np.concatenate([w["good"][w["kept"]] for w in d])), np.zeros(64, bool)]) for d in arb.diag["tile_diag"]], arb.diag["v_cropped"], arb.diag.get("max_components"))
for j in worst:
    print(f"   column {j}: tile {tile[j]} index {idx[j]} sigma {sigma[j]:.2f} gap {gap[j]:.3e} HIP dist {dh[j]:.2e} oracle dist {dr[j]:.2e} (tile rank {arb.diag['tile_ranks'][tile[j]]})")
