"""
TEST INFRASTRUCTURE (oracle/) -- NumPy restatement of the image routines of /root/reference/localmd/diagnostic_plots.py.
Only tests/ may import this module.  PARITY UNPINNED (the reference cannot run here: jax is absent; its tests do not cover
these routines).  The per-pixel / per-neighbour loops are kept as in the reference (small inputs only); arithmetic in
float32 where jnp computes in float32, accumulated into the float64 images the reference allocates with np.zeros.
"""
import numpy as np


def _neighbours(k, j, d1, d2):
    """diagnostic_plots.py:137-145: the 3 x 3 window minus the centre, clipped to the field of view."""
    for c1 in range(k - 1, k + 2):
        for c2 in range(j - 1, j + 2):
            if 0 <= c1 < d1 and 0 <= c2 < d2 and not (c1 == k and c2 == j):
                yield c1, c2


def _cov01(a, b):
    """jnp.cov(a, b)[0, 1]: ddof = 1."""
    a = a.astype(np.float64); b = b.astype(np.float64)
    return np.sum((a - a.mean()) * (b - b.mean())) / (len(a) - 1)


def _scaled_cov_image(num_movie, original_movie, mode):
    """diagnostic_plots.py:117-160 / :187-221 with the numerator traces given."""
    T, d1, d2 = original_movie.shape
    counts = np.zeros((d1, d2)); net = np.zeros((d1, d2))
    var = original_movie.astype(np.float64).var(axis=0)     # jnp.var: ddof = 0
    for k in range(d1):
        for j in range(d2):
            for c1, c2 in _neighbours(k, j, d1, d2):
                val = _cov01(num_movie[:, k, j], num_movie[:, c1, c2]) / np.sqrt(var[k, j] * var[c1, c2])
                if mode == "mean":
                    net[k, j] += val
                elif mode == "max":
                    net[k, j] = max(val, net[k, j])
                else:
                    raise ValueError(f"mode {mode} not supported")
                counts[k, j] += 1
    return net / counts if mode == "mean" else net


def make_residual_correlation_image(original_movie, pmd_movie, mode="max"):
    """diagnostic_plots.py:100-163."""
    return _scaled_cov_image(original_movie.astype(np.float64) - pmd_movie.astype(np.float64), original_movie, mode)


def make_pmd_correlation_image(original_movie, pmd_movie, mode="max"):
    """diagnostic_plots.py:166-223."""
    return _scaled_cov_image(pmd_movie, original_movie, mode)


def _corr(t1, t2):
    """diagnostic_plots.py:235-241 (compute_correlation)."""
    a = t1.astype(np.float64) - t1.astype(np.float64).mean(axis=0, keepdims=True)
    b = t2.astype(np.float64) - t2.astype(np.float64).mean(axis=0, keepdims=True)
    return np.sum((a / np.linalg.norm(a, axis=0, keepdims=True)) * (b / np.linalg.norm(b, axis=0, keepdims=True)), axis=0)


def make_correlation_image(movie, mode="max"):
    """diagnostic_plots.py:225-271."""
    T, d1, d2 = movie.shape
    counts = np.zeros((d1, d2)); net = np.zeros((d1, d2))
    for k in range(d1):
        for j in range(d2):
            for c1, c2 in _neighbours(k, j, d1, d2):
                val = float(_corr(movie[:, k, j], movie[:, c1, c2]))
                if mode == "mean":
                    net[k, j] += val
                elif mode == "max":
                    net[k, j] = max(val, net[k, j])
                else:
                    raise ValueError(f"mode {mode} not supported")
                counts[k, j] += 1
    return net / counts if mode == "mean" else net


def make_autocorrelation_image(movie, lag=1):
    """diagnostic_plots.py:274-304."""
    T, d1, d2 = movie.shape
    out = np.zeros((d1, d2))
    for k in range(d1):
        out[k] = _corr(movie[lag:, k, :], movie[:-lag, k, :])
    return out
