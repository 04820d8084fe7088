"""Per-tile comparison of the HIP tile bases against the float64 arbiter for one fuzz case."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_parity as tp
from tests.test_gpu_fuzz import draw_cases
from tests.util import DeviceSource
from oracle import pmd_oracle as O
from localmd_amd._lib import Context

case_id = int(sys.argv[1]) if len(sys.argv) > 1 else 27
ctx = Context(0)
case, T, d1, d2, b1, b2, frames, kw = [c for c in draw_cases(case_id + 1) if c[0] == case_id][0]
print(case, T, d1, d2, b1, b2, frames, kw)
mov = tp._movie(T, d1, d2, seed=1000 + case)
pmd, diag, ref = tp._compare_full(ctx, mov, (b1, b2), frames, thresholds=(1.0, 1.7), **kw)
np.random.seed(7)
with O.arbiter_precision():
    arb = O.localmd_decomposition(mov, (b1, b2), frames, rng=DeviceSource(ctx, 123), thresholds=diag["thresholds"], dtype="float64", **kw)
d = b1 * b2
ranks = diag["tile_ranks"]
print("mean/std rel diff hip-arb", np.abs(pmd.mean_img / arb.mean_img - 1).max(), np.abs(pmd.var_img / arb.std_img - 1).max())
off = np.concatenate([[0], np.cumsum(ranks)])
for t in range(len(ranks)):
    rk = int(ranks[t])
    uh = diag["tile_ut"][t, :rk, :d].T.astype(np.float64)            # (d, rk), q = il + b1*jl
    ua = arb.diag["tile_u"][t].reshape((d, -1), order="F")[:, :rk]
    ur = ref.diag["tile_u"][t].reshape((d, -1), order="F")[:, :rk]
    ch = np.abs(np.sum(uh * ua, axis=0))
    cr = np.abs(np.sum(ur * ua, axis=0))
    # principal angles of the whole kept subspaces
    sv = np.linalg.svd(np.linalg.qr(uh)[0].T @ np.linalg.qr(ua)[0], compute_uv=False)
    sig_h = diag["col_sigma"][off[t]:off[t + 1]]
    sig_a = np.linalg.norm(arb.diag["v_cropped"][off[t]:off[t + 1]], axis=1)
    print(f"tile {t}: rank {rk}; 1-|cos| hip {np.array2string(1 - ch, precision=1)}; oracle32 {np.array2string(1 - cr, precision=1)}; "
          f"subspace min cos {sv.min():.6f}; sigma hip/arb-1 {np.array2string(sig_h / sig_a - 1, precision=1)}")
    if t >= 7:
        break
