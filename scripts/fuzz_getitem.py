"""Random keys through PMDArray.__getitem__: the device expansion (to_device) against the host expansion (the reference's
SciPy / NumPy formula, pmdarray.py:132-171); a key the host path refuses must be refused with the same exception class.
    python scripts/fuzz_getitem.py SEED N"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd.synthetic import make_movie
from localmd_amd._lib import Context

seed, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = Context(0)
Dm.QUIET = True
bad = 0
for order in ("F", "C"):
    T, d1, d2 = 157, 33, 41
    mov = make_movie(T, d1, d2, seed=40 + seed)
    np.random.seed(1)
    pmd = localmd_amd.localmd_decomposition(mov, (16, 20), T, max_components=5, background_rank=2, seed=9, thresholds=(1.0, 1.7), ctx=ctx, order=order)

    def one(dim):
        kind = rng.integers(0, 8)
        if kind == 0: return int(rng.integers(0, dim))
        if kind == 1: return int(rng.integers(-dim, 0))
        if kind == 2: return slice(None)
        if kind == 3:
            a, b = sorted(int(x) for x in rng.integers(0, dim + 1, 2)); return slice(a, b, int(rng.integers(1, 4)))
        if kind == 4: return slice(int(rng.integers(-dim, dim)), None, int(rng.choice([1, 2, -1])))
        if kind == 5: return [int(x) for x in rng.integers(0, dim, int(rng.integers(1, 6)))]
        if kind == 6: return np.asarray(rng.integers(-dim, dim, int(rng.integers(1, 5))))
        return slice(int(rng.integers(0, dim)), int(rng.integers(0, dim)))      # possibly empty

    keys = []
    for _ in range(n):
        nk = int(rng.choice([1, 2, 3, 3, 3]))
        k = tuple(one(dm) for dm in (T, d1, d2)[:nk])
        keys.append(k if nk > 1 else k[0])
    host = []
    for k in keys:
        try: host.append(pmd[k])
        except Exception as e: host.append(e)     # noqa: BLE001
    pmd.to_device(ctx=ctx)
    for k, h in zip(keys, host):
        try: g = pmd[k]
        except Exception as e: g = e                # noqa: BLE001
        if isinstance(h, Exception) or isinstance(g, Exception):
            if type(h) is not type(g):
                bad += 1
                print(f"order {order} key {k}: host {type(h).__name__ if isinstance(h, Exception) else h.shape} device {type(g).__name__ if isinstance(g, Exception) else g.shape}  {g if isinstance(g, Exception) else ''} {h if isinstance(h, Exception) else ''}")
            continue
        ok = g.shape == h.shape and g.dtype == h.dtype and np.allclose(g, h, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(h).max()) if h.size else 1.0))
        if not ok:
            bad += 1
            print(f"order {order} key {k}: shapes {g.shape} {h.shape}, max diff {np.abs(g - h).max() if g.shape == h.shape and g.size else 'n/a'}")
    pmd.to_host()
print(f"getitem fuzz seed {seed}: {bad} of {2 * n} keys disagree")
