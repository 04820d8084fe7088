"""Time of the generic-width tile path (max_components = 80) against the 64-row path (50) on 625 tiles (GPU box)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd._lib import Context
from localmd_amd.synthetic import make_movie_torch
Dm.QUIET = True
ctx = Context(0)
mov = make_movie_torch(2000, 256, 256, ctx.device, seed=0)
for r in (50, 80, 80):
    np.random.seed(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.profile_enable(True)
    pmd, diag = localmd_amd.localmd_decomposition(mov, (20, 20), 2000, max_components=r, seed=1, thresholds=(1.0, 1.6), ctx=ctx, return_diagnostics=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    prof = ctx.profile_summary(); ctx.profile_enable(False)
    top = sorted(prof.items(), key=lambda kv: -kv[1][0])[:6]
    print(f"max_components {r}: {dt:.3f} s, tiles phase {diag['timings']['tiles']*1e3:.0f} ms, mean rank {diag['tile_ranks'].mean():.1f};", {k: round(v[0], 1) for k, v in top}, flush=True)
