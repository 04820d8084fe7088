"""
Referee fixtures for the parity tests at BASELINE-like sizes (tests/test_gpu_baseline_parity.py), generated ONCE in the
build container from the CPU oracle - the reference itself cannot run here (jax is absent) - with
the HOST Philox source (oracle/philox.py), so that the GPU box compares the HIP path with committed vectors instead of
running minutes of oracle:

    python tests/golden/make_parity_fixtures.py rle        # 128 x 128 x 10000, R <= frames: eigenvector route
    python tests/golden/make_parity_fixtures.py headline   # 352 x 352 x 10000, R > frames: the regime bench.py times
    python tests/golden/make_parity_fixtures.py headline-single   # appends the distances of the single-precision-LAPACK oracle

Each fixture holds two referees: the fp32 oracle (the reference's arithmetic up to LAPACK rounding) and the float64
arbiter (the exact result of the reference's algorithm on the same inputs).  The movies carry a ladder of bright sources
(localmd_amd.synthetic.make_movie(ladder=24)) so that >= 20 leading components have separated singular values -
SVD vectors can only be compared one by one where the spectrum is separated.
What is kept: tests/parity_metrics.fixture_from_result.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pmd_oracle as O, philox  # noqa: E402
from localmd_amd.synthetic import make_movie  # noqa: E402
from tests import parity_metrics as PM  # noqa: E402

CASES = {
    # name: movie arguments, decomposition arguments (shared verbatim with the test through the fixture's "case" entry)
    "rle": dict(T=10000, d1=128, d2=128, block=20, movie_seed=5, ladder=24, ladder_top=120.0, ladder_ratio=0.9,
                max_components=50, seed=321, np_seed=11, sim_iters=20),
    "headline": dict(T=10000, d1=352, d2=352, block=20, movie_seed=6, ladder=28, ladder_top=80.0, ladder_ratio=0.91, ladder_smooth=8.0,
                     max_components=50, seed=654, np_seed=12, sim_iters=20),
}


def case_movie(c):
    return make_movie(c["T"], c["d1"], c["d2"], seed=c["movie_seed"], ladder=c["ladder"], ladder_top=c["ladder_top"],
                      ladder_ratio=c["ladder_ratio"], ladder_smooth=float(c.get("ladder_smooth", 0.0)))


def main(name):
    c = CASES[name]
    mov = case_movie(c)
    shape = mov.shape
    out = {"case_" + k: np.asarray(v) for k, v in c.items()}
    thresholds = None
    for label, fp64 in (("f32", False), ("f64", True)):
        t0 = time.perf_counter()
        np.random.seed(c["np_seed"])
        kw = dict(max_components=c["max_components"], rng=philox.PhiloxSource(c["seed"]), sim_iters=c["sim_iters"], thresholds=thresholds)
        if fp64:
            with O.arbiter_precision():
                res = O.localmd_decomposition(mov, (c["block"], c["block"]), c["T"], dtype="float64", **kw)
        else:
            res = O.localmd_decomposition(mov, (c["block"], c["block"]), c["T"], **kw)
        thresholds = res.diag["thresholds"]      # simulated once (fp32 run), injected into the arbiter and into the HIP run
        print(f"{name} {label}: {time.perf_counter() - t0:.0f} s, rank before {res.diag['rank_before']} -> {len(res.s)}, "
              f"mean tile rank {np.mean(res.diag['tile_ranks']):.1f}", flush=True)
        fx = PM.fixture_from_result(res, shape)
        print(f"   signal components {len(fx['signal'])}: s = {np.round(fx['s'][fx['signal']][:30]).tolist()}", flush=True)
        for k, v in fx.items():
            out[label + "_" + k] = v
        del res
    path = os.path.join(HERE, f"parity_{name}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def add_single_lapack_distances(name):
    """Third referee, distances only: the oracle with TRUE single-precision LAPACK (scipy's s-routines: the arithmetic jaxlib's
    CPU kernels run the reference in) against the float64 arbiter already in the fixture, on the arbiter's signal components.
    Appended to the .npz as ref32s_*: what the reference's own arithmetic loses in this regime, the yardstick for the HIP path."""
    path = os.path.join(HERE, f"parity_{name}.npz")
    g = dict(np.load(path, allow_pickle=False))
    c = CASES[name]
    mov = case_movie(c)
    O.LAPACK_PRECISION = "single"
    t0 = time.perf_counter()
    try:
        np.random.seed(c["np_seed"])
        res = O.localmd_decomposition(mov, (c["block"], c["block"]), c["T"], max_components=c["max_components"],
                                      rng=philox.PhiloxSource(c["seed"]), thresholds=tuple(g["f32_thresholds"]))
    finally:
        O.LAPACK_PRECISION = "double"
    sig = g["f64_signal"].astype(np.int64)
    s_b = g["f64_s"].astype(np.float64)
    g["ref32s_tile_ranks_equal"] = np.array(np.array_equal(res.diag["tile_ranks"], g["f64_tile_ranks"]))
    g["ref32s_s_rel_signal"] = np.abs(np.asarray(res.s, np.float64)[sig] - s_b[sig]) / s_b[sig]
    va, vb = np.asarray(res.v[sig], np.float64), g["f64_Vt_signal"].astype(np.float64)
    sgn = np.where(np.sum(va * vb, axis=1) < 0, -1.0, 1.0)
    g["ref32s_vt_row_err"] = np.linalg.norm(va * sgn[:, None] - vb, axis=1) / np.linalg.norm(vb, axis=1)
    u = res.u.tocsr()
    pix = g["f64_pix_sample"].astype(np.int64)
    ur = np.asarray(u[pix] @ np.asarray(res.r[:, sig], np.float64)) * sgn[None, :]
    urb = g["f64_UR_signal_sample"].astype(np.float64)
    g["ref32s_ur_col_err"] = np.linalg.norm(ur - urb, axis=0) / np.linalg.norm(urb, axis=0)
    pi, pt = g["f64_probe_pix"].astype(np.int64), g["f64_probe_frame"].astype(np.int64)
    rec = np.einsum("pk,k,kp->p", np.asarray(u[pi] @ res.r), res.s, res.v[:, pt])
    g["ref32s_probes"] = np.array(np.abs(rec - g["f64_probe_rec"]).max() / np.abs(g["f64_probe_rec"]).max())
    print(f"{name} single-LAPACK oracle: {time.perf_counter() - t0:.0f} s; vs arbiter: s {g['ref32s_s_rel_signal'].max():.2e}, "
          f"Vt rows {g['ref32s_vt_row_err'].max():.2e}, (U R) {g['ref32s_ur_col_err'].max():.2e}, probes {float(g['ref32s_probes']):.2e}, "
          f"tile ranks equal {bool(g['ref32s_tile_ranks_equal'])}", flush=True)
    np.savez_compressed(path, **g)


if __name__ == "__main__":
    for nm in sys.argv[1:] or ["rle"]:
        if nm.endswith("-single"):
            add_single_lapack_distances(nm[:-7])
        else:
            main(nm)
