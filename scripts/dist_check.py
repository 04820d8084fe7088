"""N ranks on one GPU (gloo) vs one rank.  Every rank keeps only the pixel slab of its band of tile rows; sums over
pixels (background projection) and over component rows (the two Gram matrices of the Cholesky route) are all-reduced,
R is gathered on rank 0: tile ranks and the CSR structure must equal the single-rank ones, statistics images bit for
bit, all floating-point results up to fp32 summation order.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/dist_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd.synthetic import make_movie
Dm.QUIET = True
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
import datetime
# a stuck collective raises after three minutes instead of waiting for gloo's default half hour
dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=180))
rank = dist.get_rank()
ok_all = True
cases = [
    ("R<=frames", make_movie(400, 50, 46, seed=5), (20, 20), dict(max_components=6, background_rank=2)),
    ("R>frames", make_movie(300, 70, 80, seed=3), (10, 10), dict(max_components=8, background_rank=3)),
    # a random subset of the frames (drawn on rank 0, broadcast), pixel weights, C order
    ("subset+weights", make_movie(900, 44, 60, seed=8), (16, 20),
     dict(frame_range=300, max_components=5, background_rank=2, order="C",
          pixel_weighting=(0.5 + np.random.default_rng(1).random((44, 60))).astype(np.float32))),
    # several temporal windows (residual fitting on the later ones)
    ("windows", make_movie(900, 40, 44, seed=6), (20, 20), dict(frame_range=600, window_chunks=200, max_components=8, background_rank=2)),
    # denoiser hooks (every rank applies them to its own tiles) and rank_prune (the random projection is keyed by the seed)
    ("hooks", make_movie(500, 40, 50, seed=9), (20, 20), dict(max_components=6, background_rank=2,
                                                              temporal_denoiser=lambda v: 0.5 * v + 0.25 * (v.roll(1, -1) + v.roll(-1, -1)))),
    ("rank_prune", make_movie(300, 60, 60, seed=10), (10, 10), dict(max_components=8, background_rank=2, rank_prune=True)),
]
for name, mov, blk, kw in cases:
    kw = dict(kw)
    T = mov.shape[0]
    fr = kw.pop("frame_range", T)
    np.random.seed(3 + dist.get_rank())   # different host RNG states: rank 0's draws must win
    a, da = localmd_amd.localmd_decomposition(mov, blk, fr, seed=4, sim_iters=8, distributed=True, return_diagnostics=True, **kw)
    np.random.seed(3)
    b, db = localmd_amd.localmd_decomposition(mov, blk, fr, seed=4, sim_iters=8, distributed=False, return_diagnostics=True, **kw)
    # every rank filters its own pixel slab with a background projection that is an all-reduced sum over the ranks:
    # the tile inputs agree with the single-rank ones to fp32 summation order, not bit for bit
    ok = np.array_equal(da["tile_ranks"], db["tile_ranks"]) and np.allclose(da["tile_ut"], db["tile_ut"], atol=5e-4)
    detail = ""
    if a is not None:
        ok = ok and np.array_equal(a.u.indices, b.u.indices) and np.allclose(a.u.data, b.u.data, atol=5e-4)
        ok = ok and np.array_equal(a.mean_img, b.mean_img) and np.array_equal(a.var_img, b.var_img)
        # strong singular values to fp32 summation order; the weak ones of the ill-conditioned routes (eigenvector
        # orthogonaliser, rank_prune) carry eps (s_1 / s_c)^2 of relative error on either side
        ok = ok and a.s.shape == b.s.shape and np.all(np.abs(a.s - b.s) <= np.maximum(2e-4, 2e-6 * (b.s[0] / np.maximum(b.s, 1e-30)) ** 2) * b.s)
        rng = np.random.default_rng(0)
        pi = rng.integers(0, mov.shape[1] * mov.shape[2], 300)
        pt = rng.integers(0, T, 300)
        ra = np.einsum("pk,k,kp->p", (a.u @ a.r)[pi], a.s, a.v[:, pt])
        rb = np.einsum("pk,k,kp->p", (b.u @ b.r)[pi], b.s, b.v[:, pt])
        err = np.abs(ra - rb).max() / np.abs(rb).max()
        ok = ok and err < 1e-3
        detail = f"orthogonalizer {da['orthogonalizer']}, rank_before {da['rank_before']}, crop {da['crop']}, recon err {err:.2e}, s err {np.abs(a.s / b.s - 1).max():.2e}"
    else:
        detail = "non-root rank: results live on rank 0" if da["orthogonalizer"] == "cholesky" and da["rank_before"] > da["crop"] else "MISSING RESULT"
        ok = ok and da["frames"] == db["frames"]
        ok = ok and detail.startswith("non-root")
    print(f"rank {rank} case {name}: ok={ok} {detail}", flush=True)
    ok_all = ok_all and ok
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok_all else 1)
