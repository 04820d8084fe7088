"""
Generates tests/golden/oracle_small.npz from the CPU oracle (the reference itself cannot run
in the build container: jax is absent).  The fixture is a regression
pin of the oracle and the expected output of the HIP path on the same seeded input.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pmd_oracle as O, philox  # noqa: E402
from localmd_amd.synthetic import make_movie  # noqa: E402

mov = make_movie(400, 30, 36, seed=11)
np.random.seed(3)
res = O.localmd_decomposition(mov, (20, 16), 400, max_components=5, background_rank=2,
                              rng=philox.PhiloxSource(5), sim_iters=8)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_small.npz")
np.savez_compressed(
    out, movie_seed=11, movie_shape=np.array(mov.shape), tile_ranks=res.diag["tile_ranks"],
    thresholds=np.array(res.diag["thresholds"]), U_data=res.u.data, U_indices=res.u.indices, U_indptr=res.u.indptr,
    U_shape=np.array(res.u.shape), R=res.r, s=res.s, Vt=res.v, mean_img=res.mean_img, std_img=res.std_img,
    frames=np.array(res.diag["frames"]))
print("wrote", out, os.path.getsize(out), "bytes")
