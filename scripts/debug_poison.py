"""Find reads of uninitialised device memory: every buffer the host driver allocates (torch.empty, the context's workspaces)
is filled with NaN (floats) / a large sentinel (ints) before use, then one draw of a fuzz family runs.  A NaN in the
results = something was read before it was written.   python scripts/debug_poison.py FAMILY SEED N CASE"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_gpu_fuzz as F
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd.synthetic import make_movie
from localmd_amd._lib import Context

family, seed, n, case = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
c = [c for c in {"base": F.draw_cases, "wide": F.draw_wide_cases, "options": F.draw_option_cases, "edge": F.draw_edge_cases}[family](n, seed) if c[0] == case][0]
T, d1, d2, b1, b2, frames, kw = c[1:8]
extra = c[8] if len(c) > 8 else {"noise": 1.0, "dtype": "float32"}
print(c[:8])
mov = make_movie(T, d1, d2, seed=1000 + case, noise=extra["noise"]).astype(extra["dtype"] if extra["dtype"] != "uint16" else "float32")
Dm.QUIET = True
ctx = Context(0)

def run():
    np.random.seed(7)
    return localmd_amd.localmd_decomposition(mov, (b1, b2), frames, seed=123, return_diagnostics=True, ctx=ctx, thresholds=(1.0, 1.7), **kw)

clean, dclean = run()
orig_empty = torch.empty
def poisoned_empty(*a, **k):
    t = orig_empty(*a, **k)
    if t.is_cuda and t.numel() > 0:
        if t.dtype in (torch.float32, torch.float64): t.fill_(float("nan"))
        elif t.dtype == torch.uint8: t.fill_(0xFF)      # NaN pattern for fp32 / fp64 views of a byte workspace
        elif t.dtype in (torch.int32, torch.int64): t.fill_(0x3FFFFFF0)
    return t
torch.empty = poisoned_empty
ctx.release_workspace()
os.environ["PMD_DEBUG"] = "1"
try:
    pois, dpois = run()
finally:
    torch.empty = orig_empty
for name in ("s", "r", "v"):
    a, b = np.asarray(getattr(clean, name)), np.asarray(getattr(pois, name))
    print(f"{name}: shape {a.shape} vs {b.shape}, NaN in poisoned run: {int(np.isnan(b).sum())}, max |diff| {np.nanmax(np.abs(a - b)) if a.shape == b.shape else 'n/a'}")
ua, ub = clean.u.tocsr(), pois.u.tocsr()
print("U: nnz", ua.nnz, ub.nnz, "NaN", int(np.isnan(ub.data).sum()), "max |diff|", np.nanmax(np.abs(ua.data - ub.data)) if ua.nnz == ub.nnz else "n/a")
print("tile ranks", dclean["tile_ranks"], dpois["tile_ranks"])
