"""Local correlation images (localmd_amd/diagnostic_images.py, csrc/diag.hip) against the NumPy restatement of
/root/reference/localmd/diagnostic_plots.py (oracle/diag_oracle.py), on a movie whose mean is 100 x its standard deviation
(the regime where uncentred fp32 second moments would lose everything)."""
import numpy as np
import pytest

from oracle import diag_oracle as DO

pytestmark = pytest.mark.gpu


def _movie(T=700, d1=13, d2=17, seed=0):
    from localmd_amd.synthetic import make_movie

    return make_movie(T, d1, d2, seed=seed)


def _pmd_like(mov, seed=1):
    """A smooth low-rank stand-in for the PMD movie (any second movie of the same shape serves the formulas)."""
    T, d1, d2 = mov.shape
    x = mov.reshape(T, -1).astype(np.float64)
    mu = x.mean(axis=0, keepdims=True)
    u, s, vt = np.linalg.svd(x - mu, full_matrices=False)
    return ((u[:, :4] * s[:4]) @ vt[:4] + mu).reshape(T, d1, d2).astype(np.float32)


@pytest.mark.parametrize("mode", ["max", "mean"])
def test_correlation_images_match_reference_formulas(gpu_ctx, mode):
    from localmd_amd import diagnostic_images as DI

    mov = _movie()
    pmd = _pmd_like(mov)
    np.testing.assert_allclose(DI.make_correlation_image(mov, mode=mode, ctx=gpu_ctx), DO.make_correlation_image(mov, mode), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(DI.make_pmd_correlation_image(mov, pmd, mode=mode, ctx=gpu_ctx), DO.make_pmd_correlation_image(mov, pmd, mode),
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(DI.make_residual_correlation_image(mov, pmd, mode=mode, ctx=gpu_ctx),
                               DO.make_residual_correlation_image(mov, pmd, mode), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("lag", [1, 3])
def test_autocorrelation_image(gpu_ctx, lag):
    from localmd_amd import diagnostic_images as DI

    mov = _movie(T=900, seed=2)
    np.testing.assert_allclose(DI.make_autocorrelation_image(mov, lag=lag, ctx=gpu_ctx), DO.make_autocorrelation_image(mov, lag), rtol=1e-5, atol=1e-6)


def test_chunked_streaming_and_inputs(gpu_ctx, monkeypatch):
    """Several resident chunks (incl. the lag overlap), device tensors and a PMDArray as inputs; bad mode raises."""
    import torch
    import localmd_amd
    from localmd_amd import diagnostic_images as DI, decomposition as Dm

    Dm.QUIET = True
    mov = _movie(T=600, d1=20, d2=22, seed=3)
    full = DI.make_correlation_image(mov, ctx=gpu_ctx)
    full_ac = DI.make_autocorrelation_image(mov, lag=2, ctx=gpu_ctx)
    monkeypatch.setattr(DI, "CHUNK_BYTES", 4 * 20 * 22 * 97)      # 97-frame chunks
    np.testing.assert_allclose(DI.make_correlation_image(torch.from_numpy(mov).to(gpu_ctx.device), ctx=gpu_ctx), full, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(DI.make_autocorrelation_image(mov, lag=2, ctx=gpu_ctx), full_ac, rtol=2e-5, atol=1e-6)
    np.random.seed(0)
    arr = localmd_amd.localmd_decomposition(mov, (20, 20), 600, max_components=4, background_rank=1, seed=3, sim_iters=5, ctx=gpu_ctx)
    dense = np.asarray(arr[:, :, :], dtype=np.float32)
    got = DI.make_residual_correlation_image(mov, arr, ctx=gpu_ctx)
    np.testing.assert_allclose(got, DO.make_residual_correlation_image(mov, dense, "max"), rtol=1e-3, atol=1e-5)
    with pytest.raises(ValueError):
        DI.make_correlation_image(mov, mode="median", ctx=gpu_ctx)


@pytest.mark.parametrize("mode", ["max", "mean"])
def test_dead_pixels_follow_python_max_semantics(gpu_ctx, mode):
    """A zero-variance (dead / saturated) pixel makes every correlation with it NaN.  The reference's running maximum is
    Python's max(cov, net_corr) (diagnostic_plots.py:150-151), which lets a NaN replace the maximum and the next neighbour
    replace the NaN without the floor at 0: pixels whose LAST neighbour is the dead one come out NaN, others may come out
    negative.  The kernel reproduces exactly that, NaN positions included."""
    import warnings
    from localmd_amd import diagnostic_images as DI

    mov = _movie(T=500, d1=11, d2=12, seed=5).copy()
    mov[:, 4, 6] = 37.0
    mov[:, 0, 0] = 5.0
    mov[:, 10, 11] = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = DO.make_correlation_image(mov, mode)
    got = DI.make_correlation_image(mov, mode=mode, ctx=gpu_ctx)
    assert np.isnan(want).any() and not np.isnan(want).all()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
