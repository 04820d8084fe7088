"""The headline workload (and the many-tile one) with every host-driver allocation poisoned: bit-identical results?
    python scripts/poison_headline.py [512 512 10000 20 | 1024 1024 1000 16]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd.synthetic import make_movie_torch
from localmd_amd._lib import Context
from tests.test_gpu_poison import poisoned_allocations

d1, d2, T, b = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (512, 512, 10000, 20)
Dm.QUIET = True
ctx = Context(0)
movie = make_movie_torch(T, d1, d2, torch.device("cuda:0"), seed=0)

def run():
    np.random.seed(0)
    t0 = time.perf_counter()
    out = localmd_amd.localmd_decomposition(movie, (b, b), T, max_components=50, seed=2024, ctx=ctx)
    return out, time.perf_counter() - t0

clean, t1 = run()
ctx.release_workspace()
torch.cuda.empty_cache()
with poisoned_allocations():
    pois, t2 = run()
print(f"{d1}x{d2}x{T} b{b}: clean {t1:.2f} s, poisoned {t2:.2f} s, components {len(clean.s)}")
for name in ("s", "r", "v"):
    a, c = np.asarray(getattr(clean, name)), np.asarray(getattr(pois, name))
    print(f"   {name}: equal {bool(a.shape == c.shape and np.array_equal(a, c))}, NaN {int(np.isnan(c).sum())}, max |diff| {float(np.nanmax(np.abs(a - c))) if a.shape == c.shape else 'shape'}")
ua, ub = clean.u.tocsr(), pois.u.tocsr()
print("   U equal", bool(np.array_equal(ua.indptr, ub.indptr) and np.array_equal(ua.indices, ub.indices) and np.array_equal(ua.data, ub.data)))
