import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from localmd_amd._lib import Context
import test_gpu_parity as P
from util import sign_align
ctx = Context(0)
mov = P._movie(400, 36, 40, seed=6)
rng = np.random.default_rng(3)
pw = (0.5 + rng.random((36, 40))).astype(np.float32)
for name, kw in (("pwC", {"pixel_weighting": pw, "order": "C"}), ("pw", {"pixel_weighting": pw})):
    pmd, diag, ref = P._compare_full(ctx, mov, (18, 20), 400, max_components=5, background_rank=1, sim_iters=10, **kw)
    n = min(len(pmd.s), len(ref.s))
    va = sign_align(pmd.v[:n], ref.v[:n], axis=1)
    errs = np.linalg.norm(va - ref.v[:n], axis=1)
    print(name, "ranks equal", np.array_equal(diag["tile_ranks"], ref.diag["tile_ranks"]), "s rel", np.abs(pmd.s[:n] / ref.s[:n] - 1)[:8],
          "vt err", errs[:8])
    def rel_gaps(sv):
        return np.minimum(np.abs(np.diff(sv, prepend=np.inf)), np.abs(np.diff(sv, append=0))) / sv
    gaps = np.minimum(rel_gaps(ref.s)[:n], rel_gaps(pmd.s)[:n])
    rel = errs / np.linalg.norm(ref.v[:n], axis=1)
    sig = (gaps > 2e-2) & (ref.s[:n] > 5e-2 * ref.s[0])
    print("n", n, "sig count", sig.sum(), "err_sig", np.linalg.norm(va[sig] - ref.v[:n][sig]) / np.linalg.norm(ref.v[:n][sig]))
    for c in np.argsort(-rel * sig)[:6]:
        print("  comp", c, "s/s1 %.3f" % (ref.s[c] / ref.s[0]), "gap %.4f" % gaps[c], "err %.2e" % rel[c], "bound %.2e" % (6e-8 * ref.s[0] / (ref.s[c] * gaps[c])))
    # per-tile sigma comparison
