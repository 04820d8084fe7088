"""Random shapes through the two public SVD helpers (decomposition.py:936-1060 of the reference) against the oracle: shapes,
singular values, the product they factorise; degenerate inputs (zero rows / columns, rank-deficient data, one row).
    python scripts/fuzz_svd_helpers.py SEED N"""
import os, sys
import numpy as np, scipy.sparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import localmd_amd
from oracle import pmd_oracle as O
seed, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0
for case in range(n):
    # projected_svd(projection (D x n1), data (n1 x n2))
    D, n1, n2 = int(rng.integers(1, 80)), int(rng.integers(1, 40)), int(rng.integers(1, 120))
    p = rng.standard_normal((D, n1)).astype(np.float32)
    v = (rng.standard_normal((n1, n2)) * np.geomspace(10, 0.1, n1)[:, None]).astype(np.float32)
    kind = rng.integers(0, 5)
    if kind == 1 and n1 > 1: v[-1] = 0
    if kind == 2 and n2 > 1: v[:, 0] = 0
    if kind == 3 and n1 > 2: v[1] = v[0]
    try: a = localmd_amd.projected_svd(p, v)
    except Exception as e: a = e       # noqa: BLE001
    try: b = O.projected_svd(p, v)
    except Exception as e: b = e       # noqa: BLE001
    if isinstance(a, Exception) or isinstance(b, Exception):
        if type(a) is not type(b):
            bad += 1; print(f"projected_svd case {case} D={D} n1={n1} n2={n2} kind={kind}: {type(a).__name__ if isinstance(a, Exception) else 'ok'} ({a if isinstance(a, Exception) else ''}) vs {type(b).__name__ if isinstance(b, Exception) else 'ok'}")
    else:
        ok = all(x.shape == y.shape for x, y in zip(a, b))
        if ok:
            full, full0 = (a[0] * a[1]) @ a[2], (b[0] * b[1]) @ b[2]
            scale = max(np.abs(p @ v).max(), 1e-30)
            ok = np.allclose(a[1], b[1], rtol=1e-3, atol=1e-3 * max(b[1].max(), 1e-30)) and np.abs(full - p @ v).max() < 2e-3 * scale and np.abs(full0 - p @ v).max() < 2e-3 * scale
        if not ok:
            bad += 1; print(f"projected_svd case {case} D={D} n1={n1} n2={n2} kind={kind}: shapes {[x.shape for x in a]} vs {[y.shape for y in b]}; s {a[1][:4]} vs {b[1][:4]}")
    # compute_lowrank_factorized_svd(u sparse (D x R), v (R x T))
    D, R, T = int(rng.integers(20, 200)), int(rng.integers(1, 30)), int(rng.integers(1, 60))
    u = scipy.sparse.random(D, R, density=float(rng.choice([0.05, 0.2, 0.6])), random_state=int(rng.integers(1 << 30)), format="coo")
    vm = rng.standard_normal((R, T)).astype(np.float32)
    for only_left in (True, False):
        try: a = localmd_amd.compute_lowrank_factorized_svd(u, vm, only_left=only_left)
        except Exception as e: a = e   # noqa: BLE001
        try: b = O.compute_lowrank_factorized_svd(u, vm, only_left=only_left)
        except Exception as e: b = e   # noqa: BLE001
        if isinstance(a, Exception) or isinstance(b, Exception):
            if type(a) is not type(b):
                bad += 1; print(f"factorized case {case} D={D} R={R} T={T} only_left={only_left}: {type(a).__name__ if isinstance(a, Exception) else 'ok'} ({str(a)[:100] if isinstance(a, Exception) else ''}) vs {type(b).__name__ if isinstance(b, Exception) else 'ok'} ({str(b)[:100] if isinstance(b, Exception) else ''})")
            continue
        if only_left:
            ok = a.shape == b.shape
        else:
            ok = all(x.shape == y.shape for x, y in zip(a, b))
            if ok:
                fa, fb = (u @ a[0]) * a[1] @ a[2], (u @ b[0]) * b[1] @ b[2]
                ok = np.abs(fa - fb).max() < 5e-3 * max(np.abs(fb).max(), 1e-30)
        if not ok:
            bad += 1; print(f"factorized case {case} D={D} R={R} T={T} only_left={only_left}: shapes {a.shape if only_left else [x.shape for x in a]} vs {b.shape if only_left else [y.shape for y in b]}")
print(f"svd helper fuzz seed {seed}: {bad} disagreements in {n} cases")
