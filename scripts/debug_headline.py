"""Headline fixture case under A/B settings of the global stage (GPU box):  debug_headline.py [ENV=VALUE ...]"""
import os, sys, time
for a in sys.argv[1:]:
    k, v = a.split("=")
    os.environ[k] = v
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd._lib import Context
from localmd_amd.synthetic import make_movie
from tests import parity_metrics as PM
Dm.QUIET = True
g = np.load(os.path.join(ROOT, "tests", "golden", "parity_headline.npz"))
c = {k[5:]: g[k].item() for k in g.files if k.startswith("case_")}
mov = make_movie(c["T"], c["d1"], c["d2"], seed=c["movie_seed"], ladder=c["ladder"], ladder_top=c["ladder_top"], ladder_ratio=c["ladder_ratio"],
                 ladder_smooth=float(c.get("ladder_smooth", 0.0)))
ctx = Context(0)
fx = {k[4:]: g[k] for k in g.files if k.startswith("f64_")}
for rep in range(2):
    np.random.seed(c["np_seed"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pmd, diag = localmd_amd.localmd_decomposition(mov, (c["block"],) * 2, c["T"], max_components=c["max_components"], seed=c["seed"],
                                                  thresholds=tuple(g["f32_thresholds"]), return_diagnostics=True, ctx=ctx)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
m = PM.measure_fixture(pmd, diag, fx)
print(sys.argv[1:], f"{dt:.3f} s; orthogonalizer {diag['orthogonalizer']}; null {diag['null_direction']}", flush=True)
for ln in PM.fixture_summary("HIP vs arbiter", m)[1:3]:
    print("   ", ln, flush=True)
