"""One draw of the 'wide' fuzz family through the HIP path under its A/B settings (orthogonaliser route, null-direction
rule) next to the oracle forms: fit to the data, singular values, orthonormality.
    python scripts/debug_wide.py SEED N_DRAWN CASE [FAMILY]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fuzz import draw_wide_cases, draw_option_cases, draw_cases, probe_fit
import tests.test_gpu_parity as tp
from tests.util import DeviceSource
from oracle import pmd_oracle as O
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd._lib import Context

seed, n, case = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
family = sys.argv[4] if len(sys.argv) > 4 else "wide"
c = [c for c in {"wide": draw_wide_cases, "options": draw_option_cases, "base": draw_cases}[family](n, seed) if c[0] == case][0]
_, T, d1, d2, b1, b2, frames, kw = c[:8]
extra = c[8] if len(c) > 8 else {"noise": 1.0}
print(c, flush=True)
ctx = Context(0)
Dm.QUIET = True
from localmd_amd.synthetic import make_movie
mov = make_movie(T, d1, d2, seed=1000 + case, noise=extra["noise"])
order = kw.get("order", "F")

def orc(lapack="double", fp64=False, thr=None):
    O.LAPACK_PRECISION = lapack
    np.random.seed(7)
    try:
        if fp64:
            with O.arbiter_precision():
                return O.localmd_decomposition(mov, (b1, b2), frames, rng=DeviceSource(ctx, 123), thresholds=thr, dtype="float64", **kw)
        return O.localmd_decomposition(mov, (b1, b2), frames, rng=DeviceSource(ctx, 123), thresholds=thr, **kw)
    finally:
        O.LAPACK_PRECISION = "double"

def describe(name, res):
    s = np.asarray(res.s, np.float64)
    ur = np.asarray(res.u @ res.r, np.float64)
    v = np.asarray(res.v, np.float64)
    eo = np.abs(ur.T @ ur - np.eye(len(s)))
    ev = np.abs(v @ v.T - np.eye(len(s)))
    print(f"{name:34s} fit {probe_fit(res, mov, ref.mean_img, ref.std_img, order):10.4f}  n {len(s)}  s[:3] {np.round(s[:3], 2)}  s[-3:] {s[-3:]}  "
          f"|UR^T UR - I| {eo.max():.2e}  |Vt Vt^T - I| {ev.max():.2e}  max|R| {np.abs(res.r).max():.2e}", flush=True)

thr = (1.0, 1.7)
ref = orc(thr=thr)
describe("oracle fp32", ref)
describe("oracle fp32 single-LAPACK", orc("single", thr=thr))
arb = orc(fp64=True, thr=thr)
describe("arbiter fp64", arb)
for orth, nd, cut in (("auto", "keep", 0.0), ("auto", "drop", 0.0), ("eigh", "keep", 0.0), ("eigh", "drop", 0.0), ("auto", "drop", 1e-6),
                      ("auto", "drop", 1e-5), ("eigh", "drop", 1e-5)):
    if True:
        np.random.seed(7)
        pmd, diag = localmd_amd.localmd_decomposition(mov, (b1, b2), frames, seed=123, return_diagnostics=True, ctx=ctx, thresholds=thr,
                                                      orthogonalizer=orth, null_directions=nd, null_cutoff=cut, **kw)
        describe(f"HIP {orth}/{nd}/{cut:g} -> {diag['orthogonalizer']}", pmd)
        n = min(len(pmd.s), len(arb.s))
        rel = np.abs(pmd.s[:n] - arb.s[:n]) / arb.s[:n]
        rel0 = np.abs(ref.s[:n] - arb.s[:n]) / arb.s[:n]
        print(f"      s rel err vs arbiter: top 20 max {rel[:20].max():.2e} (oracle fp32 {rel0[:20].max():.2e}), all max {rel.max():.2e} ({rel0.max():.2e}); "
              f"crop {diag['crop']} R {diag['rank_before']} -> {diag['rank_after']}", flush=True)
