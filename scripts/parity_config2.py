"""HIP path against the oracle at BASELINE config 2 in full (256 x 256 x 2000, 20 x 20 blocks, <= 8 components per tile):
the figures of tests/test_gpu_parity._check_full, printed instead of asserted."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_parity as tp
from tests.util import sign_align
from localmd_amd._lib import Context
from localmd_amd.synthetic import make_movie

T, d1, d2 = 2000, 256, 256
ctx = Context(0)
mov = make_movie(T, d1, d2, seed=0)
t0 = time.perf_counter()
extra = {"orthogonalizer": sys.argv[1]} if len(sys.argv) > 1 else {}
if extra:
    import localmd_amd
    np.random.seed(7)
    pmd, diag = localmd_amd.localmd_decomposition(mov, (20, 20), T, max_components=8, sim_iters=50, seed=123, return_diagnostics=True, ctx=ctx, **extra)
    np.random.seed(7)
    ref = tp.O.localmd_decomposition(mov, (20, 20), T, max_components=8, rng=tp.DeviceSource(ctx, 123), thresholds=diag["thresholds"])
    print("HIP orthogonalizer:", diag["orthogonalizer"])
else:
    pmd, diag, ref = tp._compare_full(ctx, mov, (20, 20), T, max_components=8, sim_iters=50)
print("both sides done in %.1f s; tiles %d, rank before %d -> after %d" % (time.perf_counter() - t0, len(diag["tile_ranks"]), diag["rank_before"], diag["rank_after"]))
ranks_ref = ref.diag["tile_ranks"]
mism = np.nonzero(diag["tile_ranks"] != ranks_ref)[0]
print("tile ranks differing:", len(mism), "of", len(ranks_ref), list(mism[:10]))
thr = diag["thresholds"]
for t in mism[:5]:
    d0 = ref.diag["tile_diag"][t][0]
    n_eval = int(max(ranks_ref[t], diag["tile_ranks"][t]))
    m = np.minimum(np.abs(d0["spatial"][:n_eval] - thr[0]) / thr[0], np.abs(d0["temporal"][:n_eval] - thr[1]) / thr[1])
    print("   tile", t, "ranks", diag["tile_ranks"][t], ranks_ref[t], "smallest margin to a threshold %.2e" % m.min())
if len(mism) == 0:
    print("CSR indptr equal:", np.array_equal(pmd.u.indptr, ref.u.indptr), " indices equal:", np.array_equal(pmd.u.indices, ref.u.indices),
          " max |U_data diff| %.2e" % np.abs(pmd.u.data - ref.u.data).max())
print("mean_img rel %.2e, std_img rel %.2e" % (np.abs(pmd.mean_img / ref.mean_img - 1).max(), np.abs(pmd.var_img / ref.std_img - 1).max()))
n = min(len(pmd.s), len(ref.s))
rel = np.abs(pmd.s[:n] - ref.s[:n]) / ref.s[:n]
print("s: n %d/%d, rel err max %.2e (top 100: %.2e), sigma_1/sigma_n %.1f" % (len(pmd.s), len(ref.s), rel.max(), rel[:100].max(), ref.s[0] / ref.s[n - 1]))
def rel_gaps(sv):
    return np.minimum(np.abs(np.diff(sv, prepend=np.inf)), np.abs(np.diff(sv, append=0))) / sv
gaps = np.minimum(rel_gaps(ref.s)[:n], rel_gaps(pmd.s)[:n])
va = sign_align(pmd.v[:n], ref.v[:n], axis=1)
cond = 6e-8 * (ref.s[0] / ref.s[:n]) ** 2 / np.maximum(gaps, 1e-12)
sig = (gaps > 2e-2) & (ref.s[:n] > 5e-2 * ref.s[0]) & (cond < 2e-5)
if sig.any():
    print("Vt Frobenius error on the %d separated, well-conditioned signal components: %.2e" % (sig.sum(), np.linalg.norm(va[sig] - ref.v[:n][sig]) / np.linalg.norm(ref.v[:n][sig])))
sep = gaps > 2e-2
errs = np.array([np.linalg.norm(va[c] - ref.v[c]) for c in np.nonzero(sep)[0]])
print("Vt row error over the %d separated components: median %.2e, max %.2e" % (sep.sum(), np.median(errs), errs.max()))
ur, ur0 = pmd.u @ pmd.r, ref.u @ ref.r
rng = np.random.default_rng(0)
pi = rng.integers(0, d1 * d2, 1000); pt = rng.integers(0, T, 1000)
rec = np.einsum("pk,k,kp->p", ur[pi], pmd.s, pmd.v[:, pt]); rec0 = np.einsum("pk,k,kp->p", ur0[pi], ref.s, ref.v[:, pt])
print("reconstruction probes: max |diff| %.3e of peak %.3f (%.2e relative)" % (np.abs(rec - rec0).max(), np.abs(rec0).max(), np.abs(rec - rec0).max() / np.abs(rec0).max()))
print("|(UR)^T(UR) - I| hip %.2e ref %.2e ; |Vt Vt^T - I| hip %.2e ref %.2e" % (np.abs(ur.T @ ur - np.eye(ur.shape[1])).max(), np.abs(ur0.T @ ur0 - np.eye(ur0.shape[1])).max(),
      np.abs(pmd.v @ pmd.v.T - np.eye(len(pmd.s))).max(), np.abs(ref.v @ ref.v.T - np.eye(len(ref.s))).max()))
