"""cProfile of the host side of one decomposition (after a warm-up) on a bench configuration (GPU box):
    python scripts/host_profile.py 1024x1024x1000_b16"""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd._lib import Context
from localmd_amd.synthetic import make_movie_torch
Dm.QUIET = True
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "1024x1024x1000_b16"]
dev = torch.device("cuda", 0)
movie = make_movie_torch(cfg["T"], cfg["d1"], cfg["d2"], dev, seed=0)
ctx = Context(0)
def step():
    np.random.seed(0)
    return localmd_amd.localmd_decomposition(movie, (cfg["block"],) * 2, cfg["frames"], max_components=cfg["max_components"], seed=2024, ctx=ctx)
step(); step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
print(st.getvalue())
import time
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize()
    print(f"step {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
