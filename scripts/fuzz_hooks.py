"""Draws of the base / options fuzz families with linear denoiser hooks (temporal, spatial, both; per tile or batched) through
the assertions of tests/test_gpu_fuzz.py.    python scripts/fuzz_hooks.py FAMILY SEED N"""
import os, sys, time, traceback
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_fuzz as F
import tests.test_gpu_parity as tp
from localmd_amd._lib import Context

family, seed, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cases = {"base": F.draw_cases, "options": F.draw_option_cases}[family](n, seed)
rng = np.random.default_rng(seed + 999)
ctx = Context(0)
failed = []
for c in cases:
    c = list(c)
    kw = dict(c[7])
    if kw.get("window_chunks") and rng.random() < 0.5:
        kw.pop("window_chunks")
    which = rng.choice(["temporal", "spatial", "both"])
    if which in ("temporal", "both"): kw["temporal_denoiser"] = tp._smooth_time
    if which in ("spatial", "both"): kw["spatial_denoiser"] = tp._smooth_space
    c[7] = kw
    lines, t0 = [], time.time()
    try:
        fig = F.run_case(ctx, *c, out=lines.append)
        F.check_case(fig)
        print(f"case {c[0]} hooks {which}: OK ({time.time() - t0:.1f} s)", flush=True)
    except Exception:
        failed.append(c[0])
        print("\n".join(l[:300] for l in lines) + f"\n   hooks {which} FAILED:\n" + traceback.format_exc()[-1500:], flush=True)
print(f"hooks fuzz {family} seed {seed}: {len(failed)} of {len(cases)} cases failed {failed}", flush=True)
