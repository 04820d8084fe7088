"""
Local correlation images of a movie and of its PMD approximation on the MI355X.

Same functions, arguments and results as the image routines of /root/reference/localmd/diagnostic_plots.py
(make_correlation_image :225-271, make_autocorrelation_image :274-304, make_pmd_correlation_image :166-223,
make_residual_correlation_image :100-163); the plotly figure builders of that file are not part of the hot path and
are not reproduced.  The reference loops over the pixels in Python and calls a jitted two-trace routine per neighbour
pair; here the frames stream through HBM once and two kernels (csrc/diag.hip) form every first and second moment, from
which all four images follow.  Movies: anything that slices like a (T, d1, d2) array - NumPy arrays, torch tensors (host
or device), a localmd_amd.PMDArray (expanded chunk by chunk).  Results: float64 (d1, d2) NumPy arrays, like the reference.
"""
import numpy as np

from ._lib import Context, ptr

CHUNK_BYTES = 1 << 30


def _chunks(T, d1, d2, overlap=0):
    step = max(overlap + 1, CHUNK_BYTES // (4 * d1 * d2))
    t0 = 0
    while t0 < T:
        t1 = min(T, t0 + step)
        yield max(0, t0 - overlap), t0, t1
        t0 = t1


def _frames(movie, lo, hi, device):
    import torch

    x = movie[lo:hi]
    if not isinstance(x, torch.Tensor):
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
    d1, d2 = (int(v) for v in movie.shape[1:])
    return x.to(device=device, dtype=torch.float32).reshape(hi - lo, d1, d2).contiguous()


def _neighbour_moments(ctx, movie, minus=None):
    """[10, D] float64 device tensor of the neighbour moments of movie (- minus)."""
    import torch

    T, d1, d2 = (int(v) for v in movie.shape)
    D = d1 * d2
    mom = torch.zeros((10, D), dtype=torch.float64, device=ctx.device)
    ref = None
    for lo, t0, t1 in _chunks(T, d1, d2):
        a = _frames(movie, t0, t1, ctx.device)
        b = _frames(minus, t0, t1, ctx.device) if minus is not None else None
        if ref is None:
            ref = (a[0] - b[0] if b is not None else a[0]).reshape(-1).clone()
        ws = ctx.workspace(ctx.lib.pmd_diag_workspace_bytes(t1 - t0, D))
        ctx.call("pmd_neighbour_moments", ptr(a), ptr(b), ptr(ref), t1 - t0, d1, d2, 1, ptr(mom), ptr(ws), ws.numel())
    return mom


def _image(ctx, num, den, T, d1, d2, kind, mode):
    import torch

    if mode not in ("max", "mean"):
        raise ValueError(f"mode {mode} not supported")
    out = torch.empty(d1 * d2, dtype=torch.float64, device=ctx.device)
    ctx.call("pmd_neighbour_image", ptr(num), ptr(den), T, d1, d2, kind, 0 if mode == "max" else 1, ptr(out))
    ctx.sync()
    return out.cpu().numpy().reshape(d1, d2)


def _with_ctx(fn):
    def wrapper(*args, device=None, ctx=None, **kw):
        own = ctx is None
        if own:
            ctx = Context(0 if device is None else device)
        try:
            return fn(ctx, *args, **kw)
        finally:
            if own:
                ctx.release_workspace()
                ctx.close()
    wrapper.__doc__ = fn.__doc__
    wrapper.__name__ = fn.__name__
    return wrapper


@_with_ctx
def make_correlation_image(ctx, movie, mode: str = "max"):
    """Pixel i = max (from 0) or mean over the adjacent pixels j of corr(movie_i, movie_j)  (diagnostic_plots.py:225-271)."""
    T, d1, d2 = (int(v) for v in movie.shape)
    mom = _neighbour_moments(ctx, movie)
    return _image(ctx, mom, None, T, d1, d2, 0, mode)


@_with_ctx
def make_pmd_correlation_image(ctx, original_movie, pmd_movie, mode: str = "max"):
    """Pixel i = Cov(pmd_i, pmd_j) / sqrt(Var(original_i) Var(original_j)), max / mean over the adjacent j
    (diagnostic_plots.py:166-223; jnp.cov has ddof 1, jnp.var ddof 0)."""
    T, d1, d2 = (int(v) for v in original_movie.shape)
    den = _neighbour_moments(ctx, original_movie)
    num = _neighbour_moments(ctx, pmd_movie)
    return _image(ctx, num, den, T, d1, d2, 1, mode)


@_with_ctx
def make_residual_correlation_image(ctx, original_movie, pmd_movie, mode: str = "max"):
    """Pixel i = Cov(original_i - pmd_i, original_j - pmd_j) / sqrt(Var(original_i) Var(original_j))
    (diagnostic_plots.py:100-163)."""
    T, d1, d2 = (int(v) for v in original_movie.shape)
    den = _neighbour_moments(ctx, original_movie)
    num = _neighbour_moments(ctx, original_movie, minus=pmd_movie)
    return _image(ctx, num, den, T, d1, d2, 1, mode)


@_with_ctx
def make_autocorrelation_image(ctx, movie, lag: int = 1):
    """Pixel i = corr(movie_i[lag:], movie_i[:-lag])  (diagnostic_plots.py:274-304)."""
    import torch

    T, d1, d2 = (int(v) for v in movie.shape)
    lag = int(lag)
    if lag < 1 or lag >= T:
        raise ValueError("lag must be in [1, frames)")
    D = d1 * d2
    mom = torch.zeros((5, D), dtype=torch.float64, device=ctx.device)
    ref = None
    for lo, t0, t1 in _chunks(T, d1, d2, overlap=lag):
        if t1 - lo <= lag:
            continue
        a = _frames(movie, lo, t1, ctx.device)
        if ref is None:
            ref = a[0].reshape(-1).clone()
        if lo > t0 - lag:       # first chunk: it has no `lag` frames in front of it, its pairs start at t = lag
            assert lo == 0
        ws = ctx.workspace(ctx.lib.pmd_diag_workspace_bytes(t1 - lo, D))
        ctx.call("pmd_lag_moments", ptr(a), ptr(ref), t1 - lo, D, lag, 1, ptr(mom), ptr(ws), ws.numel())
    out = torch.empty(D, dtype=torch.float64, device=ctx.device)
    ctx.call("pmd_lag_image", ptr(mom), D, T - lag, ptr(out))
    ctx.sync()
    return out.cpu().numpy().reshape(d1, d2)
