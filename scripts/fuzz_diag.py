"""Random shapes / modes / lags / chunk sizes through the four local correlation images against oracle/diag_oracle.py
(the NumPy restatement of the reference's diagnostic_plots.py).    python scripts/fuzz_diag.py SEED N"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import diag_oracle as DO
from localmd_amd import diagnostic_images as DI
from localmd_amd.synthetic import make_movie
from localmd_amd._lib import Context

seed, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = Context(0)
bad = 0
default_chunk = DI.CHUNK_BYTES
for case in range(n):
    T = int(rng.integers(3, 400)); d1 = int(rng.integers(1, 24)); d2 = int(rng.integers(1, 24))
    mode = str(rng.choice(["max", "mean"])); lag = int(rng.integers(1, 6))
    mov = make_movie(T, d1, d2, seed=case, noise=float(rng.choice([0.2, 1.0, 3.0])))
    if rng.random() < 0.3: mov = mov + np.float32(rng.choice([0.0, 1e3, -50.0]))
    x = mov.reshape(T, -1).astype(np.float64); mu = x.mean(axis=0, keepdims=True)
    u, s, vt = np.linalg.svd(x - mu, full_matrices=False); k = min(3, len(s))
    pmd = ((u[:, :k] * s[:k]) @ vt[:k] + mu).reshape(T, d1, d2).astype(np.float32)
    DI.CHUNK_BYTES = default_chunk if rng.random() < 0.5 else 4 * d1 * d2 * int(rng.integers(lag + 2, max(lag + 3, T)))
    checks = [("corr", lambda: DI.make_correlation_image(mov, mode=mode, ctx=ctx), lambda: DO.make_correlation_image(mov, mode), 1e-5),
              ("pmd", lambda: DI.make_pmd_correlation_image(mov, pmd, mode=mode, ctx=ctx), lambda: DO.make_pmd_correlation_image(mov, pmd, mode), 1e-5),
              ("resid", lambda: DI.make_residual_correlation_image(mov, pmd, mode=mode, ctx=ctx), lambda: DO.make_residual_correlation_image(mov, pmd, mode), 2e-4),
              ("auto", lambda: DI.make_autocorrelation_image(mov, lag=lag, ctx=ctx), lambda: DO.make_autocorrelation_image(mov, lag), 1e-5)]
    for name, f, g, tol in checks:
        try: a = f()
        except Exception as e: a = e       # noqa: BLE001
        try: b = g()
        except Exception as e: b = e       # noqa: BLE001
        if isinstance(a, Exception) or isinstance(b, Exception):
            if type(a) is not type(b):
                bad += 1
                print(f"case {case} T={T} fov={d1}x{d2} mode={mode} lag={lag} {name}: HIP {type(a).__name__ if isinstance(a, Exception) else 'ok'} ({a if isinstance(a, Exception) else ''}) oracle {type(b).__name__ if isinstance(b, Exception) else 'ok'} ({b if isinstance(b, Exception) else ''})")
            continue
        both_nan = np.isnan(a) & np.isnan(b)
        ok = a.shape == b.shape and np.allclose(np.where(both_nan, 0, a), np.where(both_nan, 0, b), rtol=tol * 10, atol=2e-5) and np.array_equal(np.isnan(a), np.isnan(b))
        if not ok:
            bad += 1
            d = np.abs(np.where(both_nan, 0, a) - np.where(both_nan, 0, b)) if a.shape == b.shape else None
            print(f"case {case} T={T} fov={d1}x{d2} mode={mode} lag={lag} chunk={DI.CHUNK_BYTES} {name}: shapes {a.shape} {b.shape}, max diff {np.nanmax(d) if d is not None else 'n/a'}, NaN HIP {int(np.isnan(a).sum())} oracle {int(np.isnan(b).sum())}")
print(f"diagnostic images fuzz seed {seed}: {bad} of {4 * n} comparisons disagree")
