import cProfile, io, os, pstats, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd._lib import Context
from localmd_amd.synthetic import make_movie_torch
Dm.QUIET = True
cfg = bench.CONFIGS[sys.argv[1]]
dev = torch.device("cuda", 0)
movie = make_movie_torch(cfg["T"], cfg["d1"], cfg["d2"], dev, seed=0)
ctx = Context(0)
def step():
    np.random.seed(0)
    return localmd_amd.localmd_decomposition(movie, (cfg["block"],) * 2, cfg["frames"], max_components=cfg["max_components"], seed=2024, ctx=ctx)
step(); step()
torch.cuda.synchronize()
import _thread
orig = ctx.call
slow = []
def timed_call(name, *a):
    t0 = time.perf_counter(); r = orig(name, *a); dt = time.perf_counter() - t0
    if dt > 0.004: slow.append((name, round(dt * 1e3, 1)))
    return r
ctx.call = timed_call
sc = ctx.side(); so = sc.call
def timed_sc(name, *a):
    t0 = time.perf_counter(); r = so(name, *a); dt = time.perf_counter() - t0
    if dt > 0.004: slow.append(("side:" + name, round(dt * 1e3, 1)))
    return r
sc.call = timed_sc
for rep in range(9):
    slow.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize()
    print(f"step {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms", slow, flush=True)
