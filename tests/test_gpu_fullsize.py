"""
BASELINE-size runs (GPU): the oracle cannot finish these shapes in seconds, so the HIP path is
checked through size-independent properties of the decomposition Y ~ mean + std * ([UR] diag(s) Vt):
orthonormal [UR] and Vt (probed on random column/row subsets), sorted non-negative singular values,
canonical CSR structure consistent with the tile ranks, denoising against the noiseless ground truth
of the synthetic movie, and run-to-run determinism.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _decompose(gpu_ctx, T, d1, d2, block, max_components, seed=11):
    import torch
    import localmd_amd
    from localmd_amd import decomposition as Dm
    from localmd_amd.synthetic import make_movie_torch

    Dm.QUIET = True
    dev = gpu_ctx.device
    noisy = make_movie_torch(T, d1, d2, dev, seed=0, noise=1.0)
    np.random.seed(0)
    pmd, diag = localmd_amd.localmd_decomposition(noisy, (block, block), T, max_components=max_components, seed=seed,
                                                  ctx=gpu_ctx, return_diagnostics=True, sim_iters=50)
    return pmd, diag, noisy


def _check_properties(pmd, diag, noisy, T, d1, d2, block, clean_band=None, noisy_is_band=False):
    """clean_band = (i_lo, i_hi): build the noiseless ground truth for these FOV rows only and probe inside them (the
    84 GB movies of BASELINE configs 4 / 5 leave no room for a second full movie)."""
    import torch
    from localmd_amd.synthetic import make_movie_torch
    from localmd_amd import grid

    D = d1 * d2
    rng = np.random.default_rng(0)
    # structure: canonical CSR, column count = sum of tile ranks + background columns
    u = pmd.u
    it1, it2 = grid.tile_origins((d1, d2), (block, block))
    assert len(diag["tile_ranks"]) == len(it1) * len(it2)
    assert np.all(diag["tile_ranks"] >= 1) and np.all(diag["tile_ranks"] <= diag["max_components"])
    assert u.shape == (D, int(diag["tile_ranks"].sum()) + 15)
    assert u.indptr[0] == 0 and np.all(np.diff(u.indptr) > 0) and u.indptr[-1] == u.nnz
    rows = rng.integers(0, D, 2000)
    for r_ in rows[:200]:
        seg = u.indices[u.indptr[r_]:u.indptr[r_ + 1]]
        assert np.all(np.diff(seg) > 0)
    assert u.data.dtype == np.float64 and u.indices.dtype == np.int32 and np.all(np.isfinite(u.data))
    # every pixel is covered by 1..9 tiles plus the 15 background entries
    per_row = np.diff(u.indptr)
    assert per_row.min() >= 16 and per_row.max() <= 9 * diag["max_components"] + 15
    # spectrum
    s = pmd.s
    assert np.all(np.isfinite(s)) and np.all(s > 0) and np.all(np.diff(s) <= 1e-3 * s[0])
    assert pmd.r.shape == (u.shape[1], len(s)) and pmd.v.shape == (len(s), T)
    # orthonormality probes
    cols = np.sort(rng.choice(len(s), size=min(48, len(s)), replace=False))
    ur = u @ pmd.r[:, cols]
    g = np.abs(ur.T @ ur - np.eye(len(cols)))
    tol_u = 2e-2 + 1e-6 * (s[0] / s[cols]) ** 2
    assert np.all(g <= np.maximum(tol_u[:, None], tol_u[None, :])), g.max()
    strong = cols[s[cols] > 0.1 * s[0]]
    if len(strong) > 1:
        us = u @ pmd.r[:, strong]
        assert np.abs(us.T @ us - np.eye(len(strong))).max() < 2e-3
    # Vt = W^T V / s comes from an fp32 Gram eigendecomposition (decomposition.py:1089-1097): component c is
    # orthonormal only to ~eps * (s_1 / s_c)^2, in the reference as well
    vr = pmd.v[cols]
    dev_v = np.abs(vr @ vr.T - np.eye(len(cols)))
    tol_c = 5e-3 + 1e-6 * (s[0] / s[cols]) ** 2
    assert np.all(dev_v <= np.maximum(tol_c[:, None], tol_c[None, :])), dev_v.max()
    # denoising: on random probes the reconstruction is closer to the noiseless movie than the input is
    band = (0, d1) if clean_band is None else clean_band
    clean = make_movie_torch(T, d1, d2, noisy.device, seed=0, noise=0.0, rows=band)
    pi = rng.integers(band[0], band[1], 300)
    pj = rng.integers(0, d2, 300)
    err_rec, err_in = [], []
    # traces through the device expansion (PMDArray.to_device): the host form would build the (R x T) matrix
    pmd.to_device(device=noisy.device.index)
    traces = pmd[:, [int(a) for a in pi[:40]], [int(b) for b in pj[:40]]]  # pairwise fancy indexing: (T, 40)
    pmd.to_host()
    for q, (a, b) in enumerate(zip(pi[:40], pj[:40])):
        trace = traces[:, q]
        c = clean[:, a - band[0], b].cpu().numpy()
        y = noisy[:, a - (band[0] if noisy_is_band else 0), b].cpu().numpy()
        err_rec.append(np.mean((trace - c) ** 2))
        err_in.append(np.mean((y - c) ** 2))
    assert np.mean(err_rec) < 0.6 * np.mean(err_in), (np.mean(err_rec), np.mean(err_in))
    # statistics images
    assert abs(float(pmd.var_img.mean()) - 1.0) < 0.1  # Welch noise sigma of the N(0,1) noise
    assert abs(float(pmd.mean_img.mean()) - float(noisy[:200].mean())) < 1.0


def test_config2_256x256x2000_properties_and_determinism(gpu_ctx):
    T, d1, d2, block = 2000, 256, 256, 20
    pmd, diag, noisy = _decompose(gpu_ctx, T, d1, d2, block, 8)
    _check_properties(pmd, diag, noisy, T, d1, d2, block)
    pmd2, diag2, _ = _decompose(gpu_ctx, T, d1, d2, block, 8)
    np.testing.assert_array_equal(diag["tile_ranks"], diag2["tile_ranks"])
    np.testing.assert_array_equal(pmd.u.indices, pmd2.u.indices)
    np.testing.assert_array_equal(pmd.u.data, pmd2.u.data)
    np.testing.assert_allclose(pmd.s, pmd2.s, rtol=1e-5)


def test_config3_512x512x10000_properties(gpu_ctx):
    import torch

    free, total = torch.cuda.mem_get_info()
    if total < 150 * 2 ** 30:
        pytest.skip("needs an MI355X-class HBM capacity")
    T, d1, d2, block = 10000, 512, 512, 20
    pmd, diag, noisy = _decompose(gpu_ctx, T, d1, d2, block, 50)
    assert len(diag["tile_ranks"]) == 2601
    _check_properties(pmd, diag, noisy, T, d1, d2, block)
    gpu_ctx.release_workspace()


def test_many_small_tiles_1024x1024_b16(gpu_ctx):
    """Block 16x16 with 50 % overlap on a 1024x1024 field of view (the tile shape of BASELINE config 5 on a
    quarter of its FOV): 16129 tiles, R ~ 3e5 >> frames."""
    import torch

    free, total = torch.cuda.mem_get_info()
    if total < 100 * 2 ** 30:
        pytest.skip("needs an MI355X-class HBM capacity")
    T, d1, d2, block = 600, 1024, 1024, 16
    pmd, diag, noisy = _decompose(gpu_ctx, T, d1, d2, block, 50)
    assert len(diag["tile_ranks"]) == 127 * 127
    assert diag["rank_before"] > diag["crop"] and diag["orthogonalizer"] == "cholesky"
    _check_properties(pmd, diag, noisy, T, d1, d2, block)
    gpu_ctx.release_workspace()


def test_large_tiles_1024x1024_b32(gpu_ctx):
    """Block 32x32 (1024-pixel tiles, the K-split tile kernels) on a 1024x1024 field of view: the spatial
    shape of BASELINE config 4 with a short time axis."""
    import torch

    free, total = torch.cuda.mem_get_info()
    if total < 100 * 2 ** 30:
        pytest.skip("needs an MI355X-class HBM capacity")
    T, d1, d2, block = 800, 1024, 1024, 32
    pmd, diag, noisy = _decompose(gpu_ctx, T, d1, d2, block, 50)
    assert len(diag["tile_ranks"]) == 63 * 63
    _check_properties(pmd, diag, noisy, T, d1, d2, block)
    gpu_ctx.release_workspace()


def test_beyond_2g_elements_1024x1024x8000_b32(gpu_ctx):
    """33.5 GB movie: every movie-sized array has more than 2^31 elements (64-bit indexing in the movie passes, the tile
    kernels and the projection), R = 1.1e5 > frames, eigenproblem of order 7999."""
    import torch

    free, total = torch.cuda.mem_get_info()
    if total < 250 * 2 ** 30:
        pytest.skip("needs an MI355X-class HBM capacity")
    T, d1, d2, block = 8000, 1024, 1024, 32
    pmd, diag, noisy = _decompose(gpu_ctx, T, d1, d2, block, 50)
    assert len(diag["tile_ranks"]) == 63 * 63 and diag["rank_before"] > diag["crop"]
    _check_properties(pmd, diag, noisy, T, d1, d2, block)
    del pmd, noisy
    gpu_ctx.release_workspace()
    torch.cuda.empty_cache()


def test_synthetic_slab_source_matches_whole_movie(gpu_ctx):
    """A band of FOV rows generated on its own equals the same rows of the whole synthetic movie."""
    import torch
    from localmd_amd.synthetic import make_movie_torch, SyntheticSlabSource

    dev = gpu_ctx.device
    whole = make_movie_torch(300, 200, 90, dev, seed=3)
    src = SyntheticSlabSource(300, 200, 90, dev, seed=3)
    for lo, hi in ((0, 200), (0, 70), (50, 130), (128, 200), (63, 65)):
        assert torch.equal(src.slab(lo, hi), whole[:, lo:hi, :])


def test_config4_1024x1024x20000_one_gpu_properties(gpu_ctx):
    """BASELINE config 4 at full size on ONE GPU (BASELINE.json quotes it on 8): the single-copy memory plan keeps the
    84 GB movie, its one standardised copy and the order-20 000 global stage inside 288 GB (peak 244 GB measured)."""
    import torch

    free, total = torch.cuda.mem_get_info()
    if total < 250 * 2 ** 30:
        pytest.skip("needs 288 GB of HBM")
    T, d1, d2, block = 20000, 1024, 1024, 32
    pmd, diag, noisy = _decompose(gpu_ctx, T, d1, d2, block, 50)
    assert len(diag["tile_ranks"]) == 3969 and pmd.s.shape == (T,)
    gpu_ctx.release_workspace()
    torch.cuda.empty_cache()
    _check_properties(pmd, diag, noisy, T, d1, d2, block, clean_band=(480, 544))


def test_config5_2048x2048x5000_one_gpu_properties(gpu_ctx):
    """BASELINE config 5 at full size on ONE GPU: 65 025 tiles of 16 x 16 pixels, > 10^6 tile components.  The movie is handed
    over through a one-shot source so that the decomposition can release the raw copy (peak 202 GB measured); tiles run in
    batches."""
    import torch
    import localmd_amd
    from localmd_amd import decomposition as Dm
    from localmd_amd.synthetic import make_movie_torch

    free, total = torch.cuda.mem_get_info()
    if total < 250 * 2 ** 30:
        pytest.skip("needs 288 GB of HBM")
    Dm.QUIET = True
    T, d1, d2, block = 5000, 2048, 2048, 16

    class OneShot:
        def __init__(self, t):
            self.shape, self._t = tuple(t.shape), t

        def slab(self, i_lo, i_hi):
            t, self._t = self._t, None
            return t[:, i_lo:i_hi, :]

    src = OneShot(make_movie_torch(T, d1, d2, gpu_ctx.device, seed=0, noise=1.0))
    np.random.seed(0)
    pmd, diag = localmd_amd.localmd_decomposition(src, (block, block), T, max_components=50, seed=11, ctx=gpu_ctx,
                                                  return_diagnostics=True, sim_iters=50)
    assert len(diag["tile_ranks"]) == 65025 and pmd.s.shape == (T,)
    gpu_ctx.release_workspace()
    torch.cuda.empty_cache()
    band = (1000, 1064)
    noisy_band = make_movie_torch(T, d1, d2, gpu_ctx.device, seed=0, noise=1.0, rows=band)
    _check_properties(pmd, diag, noisy_band, T, d1, d2, block, clean_band=band, noisy_is_band=True)
