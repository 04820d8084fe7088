import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PMD_DEBUG"] = "1"
import localmd_amd
from localmd_amd.synthetic import make_movie
mov = make_movie(300, 70, 80, seed=3)
np.random.seed(7)
try:
    pmd, diag = localmd_amd.localmd_decomposition(mov, (10, 10), 300, max_components=8, background_rank=3, sim_iters=10, seed=123, return_diagnostics=True)
    print("ok", diag["rank_before"], diag["rank_after"], pmd.s[:5], pmd.s[-5:])
except Exception as e:
    print("FAILED", e)
