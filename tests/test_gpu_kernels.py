"""
Kernel-level parity tests (GPU): every hand-written kernel is called through the C ABI
(include/pmd_hip.h, pmdk_* and pmd_* entry points) and compared with NumPy in float64 or
with the oracle's restatement of the reference function it replaces.
"""
import numpy as np
import pytest

from oracle import pmd_oracle as O, philox
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _t():
    import torch

    return torch


def dev(ctx, a, dtype=None):
    torch = _t()
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(ctx.device)


def P(t):
    from localmd_amd._lib import ptr

    return ptr(t)


def test_rng_matches_numpy_restatement(gpu_ctx):
    torch = _t()
    ctx = gpu_ctx
    seed = 0x1234ABCD5678
    rows, cols = 37, 11
    out = torch.zeros((3, 16, 40), dtype=torch.float32, device=ctx.device)
    # transposed, batched: out[b][col][row]
    ctx.call("pmd_rng_normal", seed, 4, 7, 2, 3, rows, cols, 1, P(out), 40, 16 * 40)
    ctx.sync()
    got = out.cpu().numpy()
    for b in range(3):
        ref = philox.normals(seed, 4, 7 + 2 * b, rows * cols).reshape(rows, cols)
        np.testing.assert_allclose(got[b, :cols, :rows], ref.T, rtol=0, atol=3e-6)
        assert np.all(got[b, cols:, :] == 0) and np.all(got[b, :, rows:] == 0)
    z = torch.empty(1 << 20, dtype=torch.float32, device=ctx.device)
    ctx.call("pmd_rng_normal", 99, 2, 0, 0, 1, 1 << 20, 1, 0, P(z), 1, 0)
    ctx.sync()
    zz = z.cpu().numpy()
    assert abs(zz.mean()) < 5e-3 and abs(zz.std() - 1) < 5e-3


def _atx_case(ctx, d, T, n_tiles, rows, slices, use_pix=True, seed=0):
    torch = _t()
    lib = ctx.lib
    rng = np.random.default_rng(seed)
    ld = lib.pmd_time_ld(T)
    dpad = lib.pmd_tile_dpad(d)
    X = np.zeros((rows, ld), dtype=np.float32)
    X[:, :T] = rng.standard_normal((rows, T)).astype(np.float32)
    if use_pix:
        pix = np.stack([rng.choice(rows, size=d, replace=False) for _ in range(n_tiles)]).astype(np.int32)
    else:
        assert rows >= n_tiles * d
        pix = (np.arange(n_tiles)[:, None] * d + np.arange(d)[None, :]).astype(np.int32)
    A = np.zeros((n_tiles, 64, dpad), dtype=np.float32)
    r = 50
    A[:, :r, :d] = rng.standard_normal((n_tiles, r, d)).astype(np.float32)
    Out = torch.full((n_tiles, 64, ld), np.nan, dtype=torch.float32, device=ctx.device)
    Xd, Ad = dev(ctx, X), dev(ctx, A)
    pd = dev(ctx, pix) if use_pix else None
    ctx.call("pmdk_tile_atx", P(Xd), ld, P(pd), d, d, d, P(Ad), 64 * dpad, dpad, P(Out), 64 * ld, ld, n_tiles, T, slices)
    ctx.sync()
    got = Out.cpu().numpy()[:, :, :T]
    ref = np.einsum("ncq,nqt->nct", A[:, :, :d].astype(np.float64), X[pix][:, :, :T].astype(np.float64))
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-6, (d, T, err)


@pytest.mark.parametrize("d,T", [(400, 1000), (400, 77), (256, 320), (100, 64), (512, 200), (784, 130), (1024, 96), (1600, 70)])
def test_tile_atx(gpu_ctx, d, T):
    _atx_case(gpu_ctx, d, T, n_tiles=5, rows=max(2 * d, 900), slices=3)


def test_tile_atx_consecutive_rows(gpu_ctx):
    _atx_case(gpu_ctx, 100, 250, n_tiles=7, rows=700, slices=1, use_pix=False)


@pytest.mark.parametrize("d,T", [(400, 1000), (400, 77), (256, 333)])
def test_tiles_project_ranked(gpu_ctx, d, T):
    """pmd_tiles_project_ranked: tiles with rank <= 32 take the half-work route; rows < rank must equal the plain projection
    and rows >= 32 of such a tile stay untouched."""
    ctx, torch = gpu_ctx, _t()
    lib = ctx.lib
    rng = np.random.default_rng(5)
    n_tiles, rows = 9, 1200
    ld, dpad = lib.pmd_time_ld(T), lib.pmd_tile_dpad(d)
    X = np.zeros((rows, ld), dtype=np.float32)
    X[:, :T] = rng.standard_normal((rows, T)).astype(np.float32)
    pix = np.stack([rng.choice(rows, size=d, replace=False) for _ in range(n_tiles)]).astype(np.int32)
    ranks = np.array([0, 1, 16, 17, 31, 32, 33, 50, 64], dtype=np.int32)
    A = np.zeros((n_tiles, 64, dpad), dtype=np.float32)
    for t, r in enumerate(ranks):
        A[t, :r, :d] = rng.standard_normal((r, d)).astype(np.float32)
    Out = torch.full((n_tiles, 64, ld), np.nan, dtype=torch.float32, device=ctx.device)
    Xd, Ad, pd, rd = dev(ctx, X), dev(ctx, A), dev(ctx, pix), dev(ctx, ranks)
    ctx.call("pmd_tiles_project_ranked", P(Xd), ld, T, P(pd), n_tiles, d, P(Ad), dpad, P(Out), ld, 2, P(rd))
    ctx.sync()
    got = Out.cpu().numpy()[:, :, :T]
    ref = np.einsum("ncq,nqt->nct", A[:, :, :d].astype(np.float64), X[pix][:, :, :T].astype(np.float64))
    scale = np.abs(ref).max()
    for t, r in enumerate(ranks):
        top = 32 if r <= 32 else 64
        assert np.abs(got[t, :top] - ref[t, :top]).max() / scale < 2e-6, (t, r)
        if top == 32:
            # untouched (the staged variant for 256 < d <= 400) or the plain product (every other variant)
            assert np.isnan(got[t, 32:]).all() or np.abs(got[t, 32:] - ref[t, 32:]).max() / scale < 2e-6, (t, r)


@pytest.mark.parametrize("d,T,r", [(1024, 333, 15), (1024, 64, 7), (1024, 1000, 16), (400, 1000, 50), (1024, 200, 17)])
def test_tile_atx_row_hint(gpu_ctx, d, T, r):
    """pmdk_tile_atx_rows: the kernel specialised on the number of rows of A that carry data - one row tile, eight pixel slices,
    for <= 16 rows on 1024-pixel tiles (the background projection) - against fp64; the rows it skips stay untouched, and a
    hint no kernel is specialised for changes nothing."""
    ctx, torch = gpu_ctx, _t()
    lib = ctx.lib
    rng = np.random.default_rng(11)
    n_tiles, rows = 7, max(2 * d, 900)
    ld, dpad = lib.pmd_time_ld(T), lib.pmd_tile_dpad(d)
    X = np.zeros((rows, ld), dtype=np.float32)
    X[:, :T] = rng.standard_normal((rows, T)).astype(np.float32)
    pix = np.stack([rng.choice(rows, size=d, replace=False) for _ in range(n_tiles)]).astype(np.int32)
    A = np.zeros((n_tiles, 64, dpad), dtype=np.float32)
    A[:, :r, :d] = rng.standard_normal((n_tiles, r, d)).astype(np.float32)
    Out = torch.full((n_tiles, 64, ld), np.nan, dtype=torch.float32, device=ctx.device)
    Xd, Ad, pd = dev(ctx, X), dev(ctx, A), dev(ctx, pix)
    ctx.call("pmdk_tile_atx_rows", P(Xd), ld, P(pd), d, d, d, P(Ad), 64 * dpad, dpad, P(Out), 64 * ld, ld, n_tiles, T, 3, r)
    ctx.sync()
    got = Out.cpu().numpy()[:, :, :T]
    ref = np.einsum("ncq,nqt->nct", A[:, :, :d].astype(np.float64), X[pix][:, :, :T].astype(np.float64))
    scale = np.abs(ref).max()
    assert np.abs(got[:, :r] - ref[:, :r]).max() / scale < 2e-6, (d, T, r)
    rest = got[:, r:]
    assert np.all(np.isnan(rest) | (np.abs(np.nan_to_num(rest)) < 1e-6 * scale)), "rows beyond the hint: untouched or zero"


def _xbt_case(ctx, d, T, n_tiles, rows, slices, shared_b=False, seed=1):
    torch = _t()
    lib = ctx.lib
    rng = np.random.default_rng(seed)
    ld = lib.pmd_time_ld(T)
    s_ld = 16 * ((d + 15) // 16)
    X = np.zeros((rows, ld), dtype=np.float32)
    X[:, :T] = rng.standard_normal((rows, T)).astype(np.float32)
    pix = np.stack([rng.choice(rows, size=d, replace=False) for _ in range(n_tiles)]).astype(np.int32)
    nb = 1 if shared_b else n_tiles
    B = np.zeros((nb, 64, ld), dtype=np.float32)
    B[:, :50, :T] = rng.standard_normal((nb, 50, T)).astype(np.float32)
    S = torch.full((n_tiles, slices, 64, s_ld), np.nan, dtype=torch.float32, device=ctx.device)
    Xd, Bd, pd = dev(ctx, X), dev(ctx, B), dev(ctx, pix)
    ctx.call("pmdk_tile_xbt", P(Xd), ld, P(pd), d, 0, d, P(Bd), 0 if shared_b else 64 * ld, ld, P(S), slices * 64 * s_ld,
             64 * s_ld, s_ld, n_tiles, T, slices)
    ctx.sync()
    got = S.cpu().numpy().astype(np.float64).sum(axis=1)[:, :, :d]
    Bu = np.broadcast_to(B, (n_tiles, 64, ld)) if shared_b else B
    ref = np.einsum("nct,nqt->ncq", Bu[:, :, :T].astype(np.float64), X[pix][:, :, :T].astype(np.float64))
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 2e-6, (d, T, err)
    if s_ld > d:
        pad = S.cpu().numpy()[:, :, :, d:s_ld]
        assert np.all(pad == 0)


@pytest.mark.parametrize("d,T,slices", [(400, 1000, 4), (400, 50, 1), (100, 333, 2), (256, 1000, 4), (1024, 100, 2), (1600, 48, 1)])
def test_tile_xbt(gpu_ctx, d, T, slices):
    _xbt_case(gpu_ctx, d, T, n_tiles=4, rows=max(2 * d, 800), slices=slices)


def test_tile_xbt_shared_b(gpu_ctx):
    _xbt_case(gpu_ctx, 256, 500, n_tiles=6, rows=2000, slices=4, shared_b=True)


def test_tile_gram_and_rowmix(gpu_ctx):
    torch = _t()
    ctx = gpu_ctx
    rng = np.random.default_rng(2)
    n, T = 5, 1234
    ld = ctx.lib.pmd_time_ld(T)
    In = np.zeros((n, 64, ld), dtype=np.float32)
    In[:, :50, :T] = rng.standard_normal((n, 50, T)).astype(np.float32) * np.logspace(0, 3, 50)[None, :, None].astype(np.float32)
    G = torch.empty((n, 4, 64, 64), dtype=torch.float64, device=ctx.device)
    Ind = dev(ctx, In)
    ctx.call("pmdk_tile_gram", P(Ind), 64 * ld, ld, T, n, 4, P(G))
    ctx.sync()
    got = G.cpu().numpy().sum(axis=1)
    ref = np.einsum("nit,njt->nij", In[:, :, :T].astype(np.float64), In[:, :, :T].astype(np.float64))
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    # rowmix, in place, fp64 accumulation
    N = rng.standard_normal((n, 64, 64))
    Nd = dev(ctx, N)
    ctx.call("pmdk_tile_rowmix", P(Ind), 64 * ld, ld, P(Nd), 4096, 50, 37, P(Ind), 64 * ld, ld, T, n)
    ctx.sync()
    out = Ind.cpu().numpy()
    ref = np.einsum("npc,npt->nct", N[:, :50, :37], In[:, :50, :T].astype(np.float64))
    assert np.abs(out[:, :37, :T] - ref).max() / np.abs(ref).max() < 2e-7
    assert np.all(out[:, 37:, :T] == 0)


@pytest.mark.parametrize("Prow,l", [(100, 60), (100, 20), (25, 20), (400, 11), (256, 60), (400, 60)])
def test_small_qr(gpu_ctx, Prow, l):
    torch = _t()
    ctx = gpu_ctx
    rng = np.random.default_rng(3)
    n = 6
    ldy = 16 * ((Prow + 15) // 16)
    Y = rng.standard_normal((n, Prow, l)).astype(np.float32)
    # make two tiles rank deficient (the reference hits this whenever frames/avg_factor < rank + 10)
    Y[0] = (rng.standard_normal((Prow, 5)) @ rng.standard_normal((5, l))).astype(np.float32)
    Y[1, :, l // 2:] = 0
    Yt = np.zeros((n, 64, ldy), dtype=np.float32)
    Yt[:, :l, :Prow] = Y.transpose(0, 2, 1)
    Qt = torch.zeros((n, 64, ldy), dtype=torch.float32, device=ctx.device)
    Ytd = dev(ctx, Yt)
    ctx.call("pmdk_small_qr", P(Ytd), 64 * ldy, ldy, Prow, l, P(Qt), 64 * ldy, ldy, n)
    ctx.sync()
    q = Qt.cpu().numpy()
    nref = min(Prow, l)
    for b in range(n):
        Q = q[b, :nref, :Prow].T.astype(np.float64)
        assert np.abs(Q.T @ Q - np.eye(nref)).max() < 5e-6, b
        resid = Y[b].astype(np.float64) - Q @ (Q.T @ Y[b].astype(np.float64))
        assert np.abs(resid).max() < 1e-4 * max(np.abs(Y[b]).max(), 1), b
        if b >= 2:
            Qn, _ = np.linalg.qr(Y[b].astype(np.float64))
            # same columns up to sign
            dots = np.abs(np.sum(Qn[:, :nref] * Q, axis=0))
            assert np.all(dots > 1 - 1e-5), b
    assert np.all(q[:, nref:, :] == 0)


def test_small_eig(gpu_ctx):
    torch = _t()
    ctx = gpu_ctx
    rng = np.random.default_rng(4)
    n_t, n = 7, 50
    G = np.zeros((n_t, 2, 64, 64))
    mats = []
    for b in range(n_t):
        A = rng.standard_normal((n, 300)) * np.logspace(0, -4 if b else 0, n)[:, None]
        if b == 1:
            A[n - 5:] = 0  # null directions
        M = A @ A.T
        mats.append(M)
        G[b, 0, :n, :n] = 0.25 * M
        G[b, 1, :n, :n] = 0.75 * M
    Gd = dev(ctx, G)
    Nout = torch.empty((n_t, 64, 64), dtype=torch.float64, device=ctx.device)
    lam = torch.empty((n_t, 64), dtype=torch.float64, device=ctx.device)
    ctx.call("pmdk_small_eig", P(Gd), 2, n, 0, 0.0, P(Nout), P(lam), n_t)
    ctx.sync()
    V = Nout.cpu().numpy()
    L = lam.cpu().numpy()
    for b in range(n_t):
        w = np.linalg.eigvalsh(mats[b])[::-1]
        assert np.all(np.diff(L[b, :n]) <= 1e-9 * w[0])
        np.testing.assert_allclose(L[b, :n], w, rtol=1e-9, atol=1e-12 * w[0])
        Vb = V[b, :n, :n]
        assert np.abs(Vb.T @ Vb - np.eye(n)).max() < 1e-12
        assert np.abs(mats[b] @ Vb - Vb * L[b, :n][None, :]).max() < 1e-11 * w[0]
    # mode 1: scaled columns, null guard
    ctx.call("pmdk_small_eig", P(Gd), 2, n, 1, 1e-10, P(Nout), P(lam), n_t)
    ctx.sync()
    V1 = Nout.cpu().numpy()
    b = 1
    kept = L[b, :n] > 1e-10 * L[b, 0]
    assert kept.sum() == n - 5
    Nn = V1[b, :n, :n]
    W = Nn.T @ mats[b] @ Nn
    # (the QL kernel resolves eigenvalues to eps64 lambda_max absolutely, the Jacobi kernel relatively: the smallest kept one is
    # 1e-8 lambda_max here, so N^T M N is the identity to ~1e-8 / 1e-16 = a few 1e-9 with QL, 1e-12 with PMD_SMALL_EIG=jacobi)
    assert np.abs(W[np.ix_(kept, kept)] - np.eye(kept.sum())).max() < 1e-7
    assert np.all(Nn[:, ~kept] == 0)


def test_pool_bin(gpu_ctx):
    torch = _t()
    ctx = gpu_ctx
    from localmd_amd import grid

    rng = np.random.default_rng(5)
    b1, b2, a, T = 20, 12, 10, 230
    d1, d2 = 30, 26
    ld = ctx.lib.pmd_time_ld(T)
    X = np.zeros((d1 * d2, ld), dtype=np.float32)
    X[:, :T] = rng.standard_normal((d1 * d2, T)).astype(np.float32)
    it1, it2 = grid.tile_origins((d1, d2), (b1, b2))
    pix, origins = grid.tile_pixel_lists((d1, d2), (b1, b2), it1, it2)
    pool_q, pool_idx, pool_w, shp = grid.pooling_maps((b1, b2), 2)
    n = pix.shape[0]
    Pn = pool_q.shape[0]
    nb = T // a
    ldb = ctx.lib.pmd_time_ld(nb)
    ab = torch.zeros((n, Pn, ldb), dtype=torch.float32, device=ctx.device)
    xbar = torch.zeros((d1 * d2, ldb), dtype=torch.float32, device=ctx.device)
    Xd, pd, pq = dev(ctx, X), dev(ctx, pix), dev(ctx, pool_q)
    ctx.call("pmdk_tile_pool_bin", P(Xd), ld, d1 * d2, P(pd), n, b1 * b2, P(pq), pool_q.shape[1], Pn, a, nb, P(xbar), P(ab),
             ldb, Pn * ldb)
    ctx.sync()
    got = ab.cpu().numpy()[:, :, :nb]
    mov = X[:, :T].reshape(d1, d2, T)
    for t, (k, j) in enumerate(origins):
        block = mov[k:k + b1, j:j + b2, : nb * a]
        ds = O.downsample_average_pooling(block, 2)
        ref = np.mean(np.reshape(ds, (ds.shape[0] * ds.shape[1], a, nb), order="F"), axis=1)
        np.testing.assert_allclose(got[t], ref, rtol=0, atol=2e-6)


def test_roughness_stats(gpu_ctx):
    torch = _t()
    ctx = gpu_ctx
    rng = np.random.default_rng(6)
    b1, b2, T, n, r = 20, 14, 777, 3, 9
    d = b1 * b2
    dpad = ctx.lib.pmd_tile_dpad(d)
    ld = ctx.lib.pmd_time_ld(T)
    Ut = np.zeros((n, 64, dpad), dtype=np.float32)
    Ut[:, :r, :d] = rng.standard_normal((n, r, d)).astype(np.float32)
    V = np.zeros((n, 64, ld), dtype=np.float32)
    V[:, :r, :T] = np.cumsum(rng.standard_normal((n, r, T)), axis=2).astype(np.float32)
    stats = torch.zeros((n, 64, 2), dtype=torch.float32, device=ctx.device)
    Ud, Vd = dev(ctx, Ut), dev(ctx, V)
    ctx.call("pmdk_roughness", P(Ud), 64 * dpad, dpad, b1, b2, P(Vd), 64 * ld, ld, T, r, P(stats), n)
    ctx.sync()
    got = stats.cpu().numpy()
    for t in range(n):
        for c in range(r):
            img = Ut[t, c, :d].reshape((b1, b2), order="F")
            sp = O.spatial_roughness_stat(img)
            tp = O.temporal_roughness_stat(V[t, c, :T])
            assert abs(got[t, c, 0] - sp) <= 3e-6 * abs(sp)
            assert abs(got[t, c, 1] - tp) <= 3e-6 * abs(tp)


def test_syevd_and_gemm(gpu_ctx):
    torch = _t()
    ctx = gpu_ctx
    rng = np.random.default_rng(7)
    n = 300
    A = rng.standard_normal((n, n)).astype(np.float32)
    S = (A @ A.T).astype(np.float32)
    Sd = dev(ctx, S.copy())
    w = torch.empty(n, dtype=torch.float32, device=ctx.device)
    work = torch.empty(n, dtype=torch.float32, device=ctx.device)
    info = torch.zeros(4, dtype=torch.int32, device=ctx.device)
    ctx.call("pmdk_syevd", n, P(Sd), n, P(w), P(work), P(info))
    ctx.sync()
    assert int(info[0]) == 0
    wv = w.cpu().numpy()
    E = Sd.cpu().numpy()  # rows are eigenvectors
    np.testing.assert_allclose(wv, np.linalg.eigvalsh(S.astype(np.float64)), rtol=2e-4, atol=1e-3)
    assert np.abs(E @ S.astype(np.float64) - wv[:, None] * E).max() < 2e-3 * wv.max()
    # row-major gemm with both transposes
    Bm = rng.standard_normal((70, 50)).astype(np.float32)
    Cm = rng.standard_normal((70, 30)).astype(np.float32)
    out = torch.empty((50, 30), dtype=torch.float32, device=ctx.device)
    Bd, Cd = dev(ctx, Bm), dev(ctx, Cm)  # keep the tensors alive until the call has run
    ctx.call("pmd_gemm", 1, 0, 50, 30, 70, 1.0, P(Bd), 50, P(Cd), 30, 0.0, P(out), 30)
    ctx.sync()
    np.testing.assert_allclose(out.cpu().numpy(), Bm.T @ Cm, rtol=1e-4, atol=1e-4)
    out2 = torch.empty((70, 70), dtype=torch.float32, device=ctx.device)
    ctx.call("pmd_gemm", 0, 1, 70, 70, 50, 1.0, P(Bd), 50, P(Bd), 50, 0.0, P(out2), 70)
    ctx.sync()
    np.testing.assert_allclose(out2.cpu().numpy(), Bm @ Bm.T, rtol=1e-4, atol=1e-4)


def _sym_matrix(rng, n):
    A = rng.standard_normal((n, n)).astype(np.float32)
    S = A @ A.T / np.float32(n) + np.diag(rng.standard_normal(n).astype(np.float32) * 3)
    S = 0.5 * (S + S.T)
    return S.astype(np.float32)


def _run_sytrd(ctx, S, impl):
    torch = _t()
    n = S.shape[0]
    ld = (n + 3) // 4 * 4
    buf = np.full((n, ld), np.nan, dtype=np.float32)  # the padding must never be read into a result
    buf[:, :n] = S
    Ad = dev(ctx, buf)
    d = torch.zeros(n, dtype=torch.float32, device=ctx.device)
    e = torch.zeros(n, dtype=torch.float32, device=ctx.device)
    tau = torch.zeros(n, dtype=torch.float32, device=ctx.device)
    ctx.call("pmdk_sytrd", n, P(Ad), ld, P(d), P(e), P(tau), impl)
    ctx.sync()
    return Ad.cpu().numpy()[:, :n], d.cpu().numpy(), e.cpu().numpy()[:n - 1], tau.cpu().numpy()[:n - 1]


@pytest.mark.parametrize("n", [3, 5, 67, 300, 1030])
def test_sytrd_own_kernels_reconstruct(gpu_ctx, n):
    """A = Q T Q^T with Q rebuilt on the host (fp64) from the stored reflectors (LAPACK ssytrd('L') layout)."""
    from scipy.linalg import eigvalsh_tridiagonal

    rng = np.random.default_rng(n)
    S = _sym_matrix(rng, n)
    Am, d, e, tau = _run_sytrd(gpu_ctx, S, 1)
    assert np.all(np.isfinite(d)) and np.all(np.isfinite(e)) and np.all(np.isfinite(tau))
    Q = np.eye(n)
    for j in range(n - 2, -1, -1):
        v = np.zeros(n)
        v[j + 1] = 1.0
        v[j + 2:] = Am[j, j + 2:]
        Q -= tau[j] * np.outer(v, v @ Q)
    Tm = np.diag(d.astype(np.float64)) + np.diag(e.astype(np.float64), 1) + np.diag(e.astype(np.float64), -1)
    S64 = S.astype(np.float64)
    scale = np.abs(S64).max() * np.sqrt(n)
    assert np.abs(Q.T @ Q - np.eye(n)).max() < 2e-5 * np.sqrt(n)
    assert np.abs(Q.T @ S64 @ Q - Tm).max() < 3e-6 * scale
    ev = eigvalsh_tridiagonal(d.astype(np.float64), e.astype(np.float64))
    np.testing.assert_allclose(ev, np.linalg.eigvalsh(S64), atol=3e-6 * scale)
    if n == 67:
        # same conventions as rocSOLVER's ssytrd: the tridiagonal matrices agree entry by entry
        _, d0, e0, tau0 = _run_sytrd(gpu_ctx, S, 0)
        np.testing.assert_allclose(d, d0, atol=2e-4 * np.abs(d0).max())
        np.testing.assert_allclose(e, e0, atol=2e-4 * np.abs(d0).max())
        np.testing.assert_allclose(tau, tau0, atol=2e-4)


def test_sytrd_eigenvalues_2500(gpu_ctx):
    from scipy.linalg import eigvalsh_tridiagonal

    n = 2500
    rng = np.random.default_rng(1)
    S = _sym_matrix(rng, n)
    _, d, e, tau = _run_sytrd(gpu_ctx, S, 1)
    ev = eigvalsh_tridiagonal(d.astype(np.float64), e.astype(np.float64))
    ref = np.linalg.eigvalsh(S.astype(np.float64))
    np.testing.assert_allclose(ev, ref, atol=3e-6 * np.abs(S).max() * np.sqrt(n))
    _, d2, e2, _ = _run_sytrd(gpu_ctx, S, 1)
    np.testing.assert_array_equal(d, d2)  # fixed summation order: bitwise reproducible
    np.testing.assert_array_equal(e, e2)


@pytest.mark.parametrize("n", [3, 40, 72, 191, 512])
def test_syevd_small_orders_in_double(gpu_ctx, n):
    """Orders <= 512 (PMD_SYEVD_F64_MAX) are diagonalised in double precision and rounded: the result is the correctly
    rounded eigendecomposition of the fp32 matrix - graded spectrum, small eigenvalues to high RELATIVE accuracy."""
    torch = _t()
    ctx = gpu_ctx
    rng = np.random.default_rng(n)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.geomspace(1e6, 1.0, n)
    S = ((q * lam) @ q.T).astype(np.float32)
    S = np.triu(S) + np.triu(S, 1).T
    ld = (n + 3) // 4 * 4
    buf = np.zeros((n, ld), dtype=np.float32)
    buf[:, :n] = S
    Sd = dev(ctx, buf)
    w = torch.empty(n, dtype=torch.float32, device=ctx.device)
    work = torch.empty(n, dtype=torch.float32, device=ctx.device)
    info = torch.zeros(4, dtype=torch.int32, device=ctx.device)
    ctx.profile_enable(True)
    ctx.call("pmdk_syevd", n, P(Sd), ld, P(w), P(work), P(info))
    ctx.sync()
    prof = ctx.profile_summary()
    ctx.profile_enable(False)
    assert "rocsolver_dsyevd" in prof and "sytrd" not in prof, prof
    assert int(info[0]) == 0
    w64, e64 = np.linalg.eigh(S.astype(np.float64))
    wv = w.cpu().numpy().astype(np.float64)
    E = Sd.cpu().numpy()[:, :n].astype(np.float64)
    np.testing.assert_allclose(wv, w64, rtol=3e-7, atol=0)
    # P = E / sqrt(lambda): the quantity the pipeline forms (decomposition.py:990-996); an fp32 divide-and-conquer solver
    # leaves 1e-2 ... 1e-1 here
    Pm = E.T / np.sqrt(wv)[None, :]
    assert np.abs(Pm.T @ S.astype(np.float64) @ Pm - np.eye(n)).max() < 5e-4


@pytest.mark.parametrize("n,force", [(150, True), (301, True), (1200, False)])
def test_syevd_own_path(gpu_ctx, n, force, monkeypatch):
    torch = _t()
    ctx = gpu_ctx
    if force:
        monkeypatch.setenv("PMD_SYEVD", "own")
    rng = np.random.default_rng(n)
    S = _sym_matrix(rng, n)
    ld = (n + 3) // 4 * 4
    buf = np.zeros((n, ld), dtype=np.float32)
    buf[:, :n] = S
    Sd = dev(ctx, buf)
    w = torch.empty(n, dtype=torch.float32, device=ctx.device)
    work = torch.empty(n, dtype=torch.float32, device=ctx.device)
    info = torch.zeros(4, dtype=torch.int32, device=ctx.device)
    ctx.profile_enable(True)
    ctx.call("pmdk_syevd", n, P(Sd), ld, P(w), P(work), P(info))
    ctx.sync()
    prof = ctx.profile_summary()
    ctx.profile_enable(False)
    assert "sytrd" in prof, prof  # the library's own tridiagonalisation ran
    assert int(info[0]) == 0
    wv = w.cpu().numpy().astype(np.float64)
    E = Sd.cpu().numpy()[:, :n].astype(np.float64)  # rows are eigenvectors
    S64 = S.astype(np.float64)
    assert np.all(np.diff(wv) >= 0)
    np.testing.assert_allclose(wv, np.linalg.eigvalsh(S64), atol=3e-6 * np.abs(S64).max() * np.sqrt(n))
    assert np.abs(E @ E.T - np.eye(n)).max() < 5e-5 * np.sqrt(n)
    assert np.abs(E @ S64 - wv[:, None] * E).max() < 1e-5 * np.abs(S64).max() * np.sqrt(n)


@pytest.mark.parametrize("n,rank", [(300, None), (1030, None), (2117, None), (600, 100)])
def test_syevd_two_stage_route(gpu_ctx, n, rank, monkeypatch):
    """PMD_SYEVD=twostage: dense -> band -> tridiagonal (sytrd2.hip), sstedc, both back-transformations.  An exactly zero
    trailing block (rank given: only the leading rank x rank block is set) makes a panel of stage 1 singular: the call has to
    fall back to the one-stage route, same contract."""
    torch = _t()
    ctx = gpu_ctx
    monkeypatch.setenv("PMD_SYEVD", "twostage")
    rng = np.random.default_rng(n)
    if rank is None:
        S = _sym_matrix(rng, n)
    else:
        S = np.zeros((n, n), dtype=np.float32)
        S[:rank, :rank] = _sym_matrix(rng, rank)
    ld = (n + 3) // 4 * 4
    buf = np.zeros((n, ld), dtype=np.float32)
    buf[:, :n] = S
    Sd = dev(ctx, buf)
    w = torch.empty(n, dtype=torch.float32, device=ctx.device)
    work = torch.empty(n, dtype=torch.float32, device=ctx.device)
    info = torch.zeros(4, dtype=torch.int32, device=ctx.device)
    ctx.profile_enable(True)
    ctx.call("pmdk_syevd", n, P(Sd), ld, P(w), P(work), P(info))
    ctx.sync()
    prof = ctx.profile_summary()
    ctx.profile_enable(False)
    assert "sy2sb" in prof, prof
    if rank is None:
        assert "sb2st" in prof and "apply_q2" in prof and "sytrd" not in prof, prof
    else:
        assert "sytrd" in prof, prof  # the fallback ran
    assert int(info[0]) == 0
    wv = w.cpu().numpy().astype(np.float64)
    E = Sd.cpu().numpy()[:, :n].astype(np.float64)
    S64 = S.astype(np.float64)
    assert np.all(np.diff(wv) >= 0)
    np.testing.assert_allclose(wv, np.linalg.eigvalsh(S64), atol=3e-6 * np.abs(S64).max() * np.sqrt(n))
    assert np.abs(E @ E.T - np.eye(n)).max() < 5e-5 * np.sqrt(n)
    assert np.abs(E @ S64 - wv[:, None] * E).max() < 1e-5 * np.abs(S64).max() * np.sqrt(n)


@pytest.mark.parametrize("m", [90, 300, 515])
def test_orthogonalize_chol_blocked(gpu_ctx, m):
    """Et = U_c^{-T} with M^T (G M) = U_c^T U_c: lower triangular, Et C Et^T = I (several 128-blocks)."""
    import ctypes as C

    torch = _t()
    ctx = gpu_ctx
    lib = ctx.lib
    rng = np.random.default_rng(m)
    Rc = 2 * m + 37
    M = rng.standard_normal((Rc, m)).astype(np.float32)
    Md = dev(ctx, M)
    Et = torch.full((m, m), float("nan"), dtype=torch.float32, device=ctx.device)
    ok = C.c_int(0)
    ws = ctx.workspace(lib.pmd_orthogonalize_chol_workspace_bytes(Rc, m))
    ctx.call("pmd_orthogonalize_chol", P(Md), Rc, m, m, P(Md), m, P(Et), m, C.byref(ok), P(ws), ws.numel())
    ctx.sync()
    assert ok.value == 1
    E = Et.cpu().numpy().astype(np.float64)
    assert np.all(np.isfinite(E))
    assert np.abs(np.triu(E, 1)).max() == 0.0
    Cm = M.astype(np.float64).T @ M.astype(np.float64)
    assert np.abs(E @ Cm @ E.T - np.eye(m)).max() < 2e-4
    # not positive definite -> ok = 0, no exception
    Z = torch.zeros((Rc, m), dtype=torch.float32, device=ctx.device)
    ctx.call("pmd_orthogonalize_chol", P(Z), Rc, m, m, P(Z), m, P(Et), m, C.byref(ok), P(ws), ws.numel())
    assert ok.value == 0


@pytest.mark.parametrize("n", [10600, 13000])
def test_sytrd_beyond_one_batch_of_partials(gpu_ctx, n):
    """n = 10600: more row blocks (166 > 160), dot chunks (42 > 40) and partial norms (332 > 320) than one load batch
    of the advance / symv kernels holds, so their continuation loops run; n = 13000 adds the row-chunk partials
    (26 > 24).  Checked against rocSOLVER's ssytrd (same conventions)."""
    torch = _t()
    ctx = gpu_ctx
    g = torch.Generator(device=ctx.device).manual_seed(5)
    X = torch.randn((n, n + 500), device=ctx.device, generator=g)
    S = (X @ X.T) / n
    del X
    out = []
    for impl in (1, 0):
        A = S.clone()
        d = torch.zeros(n, device=ctx.device)
        e = torch.zeros(n, device=ctx.device)
        tau = torch.zeros(n, device=ctx.device)
        ctx.call("pmdk_sytrd", n, P(A), n, P(d), P(e), P(tau), impl)
        ctx.sync()
        out.append((d.cpu().numpy(), e.cpu().numpy()[:n - 1], tau.cpu().numpy()[:n - 1]))
        del A
    (d1, e1, t1), (d0, e0, t0) = out
    from scipy.linalg import eigvalsh_tridiagonal

    assert np.all(np.isfinite(d1)) and np.all(np.isfinite(e1))
    # The entries of T drift apart between two fp32 reductions of a matrix with a dense spectrum (the
    # tridiagonal form is not a well-conditioned function of A); its eigenvalues are: compare those.
    ev1 = eigvalsh_tridiagonal(d1.astype(np.float64), e1.astype(np.float64))
    ev0 = eigvalsh_tridiagonal(d0.astype(np.float64), e0.astype(np.float64))
    assert np.abs(ev1 - ev0).max() < 2e-5 * np.abs(ev0).max()
    assert abs(d1.sum() - d0.sum()) < 1e-5 * abs(d0.sum())   # trace is invariant
    np.testing.assert_allclose(d1[:50], d0[:50], rtol=1e-4)   # the first columns have not drifted yet


def test_sytrd_advance_forms_agree(gpu_ctx, monkeypatch):
    """The advance step with 32 positions x 8 parts per workgroup (default) and the 64 x 4 form (PMD_SYTRD_ADVANCE=old)
    sum the same partials in different orders: same tridiagonal matrix up to fp32 rounding (compared through its
    eigenvalues, which are well conditioned; the entries of T are not), each form reproducible bit for bit."""
    from scipy.linalg import eigvalsh_tridiagonal

    torch = _t()
    ctx = gpu_ctx
    n = 3001
    g = torch.Generator(device=ctx.device).manual_seed(9)
    X = torch.randn((n, n + 300), device=ctx.device, generator=g)
    ld = (n + 3) // 4 * 4
    Sp = torch.zeros((n, ld), device=ctx.device)
    Sp[:, :n] = (X @ X.T) / n
    del X

    def run(form):
        if form is None:
            monkeypatch.delenv("PMD_SYTRD_ADVANCE", raising=False)
        else:
            monkeypatch.setenv("PMD_SYTRD_ADVANCE", form)
        A = Sp.clone()
        d = torch.zeros(n, device=ctx.device)
        e = torch.zeros(n, device=ctx.device)
        tau = torch.zeros(n, device=ctx.device)
        ctx.call("pmdk_sytrd", n, P(A), ld, P(d), P(e), P(tau), 1)
        ctx.sync()
        return d.cpu().numpy(), e.cpu().numpy()[:n - 1]

    d1, e1 = run(None)
    d1b, e1b = run(None)
    d0, e0 = run("old")
    np.testing.assert_array_equal(d1, d1b)
    np.testing.assert_array_equal(e1, e1b)
    ev1 = eigvalsh_tridiagonal(d1.astype(np.float64), e1.astype(np.float64))
    ev0 = eigvalsh_tridiagonal(d0.astype(np.float64), e0.astype(np.float64))
    assert np.abs(ev1 - ev0).max() < 2e-5 * np.abs(ev0).max()
    assert abs(d1.sum() - d0.sum()) < 1e-5 * abs(d0.sum())


@pytest.mark.parametrize("kind", ["normal", "positive", "decades", "tiny", "huge"])
def test_gemm_fp16_pieces(kind, monkeypatch):
    """Large products of pmd_gemm run as three fp16-piece products on the fp16 matrix cores (csrc/gemm_f16x2.hip); forced
    here onto a small product: row-major C = alpha op(A) op(B) + beta C, all four transpose combinations, leading
    dimensions larger than the rows, operands of every scale.  The error against fp64 must be at the level of an fp32
    product (the sgemm path of the same library is measured next to it), padding columns of C untouched."""
    torch = _t()
    from localmd_amd._lib import Context

    monkeypatch.setenv("PMD_GEMM_SPLIT_MIN_GFLOP", "0")
    monkeypatch.setenv("PMD_GEMM_SPLIT_MIN_DIM", "1")
    ctx = Context(0)  # the options are read when a context is created
    monkeypatch.setenv("PMD_GEMM_SPLIT", "0")
    ref_ctx = Context(0)
    try:
        rng = np.random.default_rng(3)
        # (sizes that are not multiples of four and odd leading dimensions take the element-wise paths of the split kernels)
        m, n, k = (190, 157, 2999) if kind in ("positive", "decades") else (192, 160, 3000)
        pad = (5, 7, 3) if kind == "decades" else (4, 8, 4)
        assert ctx.lib.pmd_gemm_split_active(ctx.handle, m, n, k) == 1
        assert ref_ctx.lib.pmd_gemm_split_active(ref_ctx.handle, m, n, k) == 0
        for ta, tb in ((0, 0), (1, 0), (0, 1), (1, 1)):
            a = rng.standard_normal((k, m) if ta else (m, k))
            b = rng.standard_normal((n, k) if tb else (k, n))
            if kind == "positive":
                a, b = np.abs(a), np.abs(b)
            elif kind == "decades":   # inner index spread over six decades in both operands, outer over three
                wk = 10.0 ** (-6 * rng.random(k))
                a = a * (wk[:, None] if ta else wk[None, :]) * 10.0 ** (-3 * rng.random((1, m) if ta else (m, 1)))
                b = b * (wk[None, :] if tb else wk[:, None])
            elif kind == "tiny":
                a, b = a * 1e-20, b * 1e-12
            elif kind == "huge":
                a, b = a * 1e15, b * 1e12
            a, b = a.astype(np.float32), b.astype(np.float32)
            c0 = (rng.standard_normal((m, n)) * np.abs(a).max() * np.abs(b).max() * 50).astype(np.float32)
            lda, ldb, ldc = a.shape[1] + pad[0], b.shape[1] + pad[1], n + pad[2]
            ab = np.zeros((a.shape[0], lda), np.float32); ab[:, :a.shape[1]] = a
            bb = np.zeros((b.shape[0], ldb), np.float32); bb[:, :b.shape[1]] = b
            cb = np.zeros((m, ldc), np.float32); cb[:, :n] = c0
            ref = 0.5 * ((a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)) - 2.0 * c0
            errs = []
            for cx in (ctx, ref_ctx):
                ad, bd, cd = dev(cx, ab), dev(cx, bb), dev(cx, cb)
                cx.call("pmd_gemm", ta, tb, m, n, k, 0.5, P(ad), lda, P(bd), ldb, -2.0, P(cd), ldc)
                cx.sync()
                got = cd.cpu().numpy()
                assert np.all(got[:, n:] == 0)
                errs.append(np.linalg.norm(got[:, :n] - ref) / np.linalg.norm(ref))
            assert errs[0] < 3e-7 and errs[0] < 2.0 * errs[1] + 1e-7, (kind, ta, tb, errs)
        # operands the pieces cannot hold go to the fp32 path: NaN / Inf propagate, an all-zero operand gives beta C
        a = rng.standard_normal((m, k)).astype(np.float32)
        b = rng.standard_normal((k, n)).astype(np.float32)
        c0 = rng.standard_normal((m, n)).astype(np.float32)
        for poison, expect in ((np.nan, "nan"), (np.inf, "inf"), (0.0, "zero")):
            a2 = a.copy()
            if expect == "zero":
                a2[:] = 0
            else:
                a2[7, 11] = poison
            ad, bd, cd = dev(ctx, a2), dev(ctx, b), dev(ctx, c0)
            ctx.call("pmd_gemm", 0, 0, m, n, k, 1.0, P(ad), k, P(bd), n, 3.0, P(cd), n)
            ctx.sync()
            got = cd.cpu().numpy()
            if expect == "zero":
                np.testing.assert_array_equal(got, 3.0 * c0)
            else:
                assert not np.isfinite(got[7]).any() and np.isfinite(np.delete(got, 7, axis=0)).all()
    finally:
        ctx.close()
        ref_ctx.close()


def test_gemm_large_product_takes_the_fp16_piece_path(gpu_ctx):
    """A product above the library's own size gate (>= 100 GFLOP, no dimension below 256) runs from fp16 pieces without any
    option set; sampled entries against fp64, next to the sgemm path of a second context (PMD_GEMM_SPLIT=0)."""
    torch = _t()
    from localmd_amd._lib import Context
    import os

    ctx = gpu_ctx
    m, n, k = 1024, 1536, 40000
    assert ctx.lib.pmd_gemm_split_active(ctx.handle, m, n, k) == 1
    g = torch.Generator(device=ctx.device).manual_seed(7)
    a = torch.randn((m, k), device=ctx.device, generator=g) * torch.logspace(-3, 0, k, device=ctx.device)[None, :]
    b = torch.randn((k, n), device=ctx.device, generator=g).abs_()
    c = torch.empty((m, n), device=ctx.device)
    ctx.call("pmd_gemm", 0, 0, m, n, k, 1.0, P(a), k, P(b), n, 0.0, P(c), n)
    ctx.sync()
    rows = torch.arange(0, m, 37, device=ctx.device)
    ref = (a[rows].double() @ b.double()).cpu().numpy()
    err = np.linalg.norm(c[rows].cpu().numpy() - ref) / np.linalg.norm(ref)
    old = os.environ.get("PMD_GEMM_SPLIT")
    os.environ["PMD_GEMM_SPLIT"] = "0"
    try:
        ref_ctx = Context(0)
    finally:
        if old is None:
            del os.environ["PMD_GEMM_SPLIT"]
        else:
            os.environ["PMD_GEMM_SPLIT"] = old
    try:
        c2 = torch.empty((m, n), device=ctx.device)
        ref_ctx.call("pmd_gemm", 0, 0, m, n, k, 1.0, P(a), k, P(b), n, 0.0, P(c2), n)
        ref_ctx.sync()
        err_sgemm = np.linalg.norm(c2[rows].cpu().numpy() - ref) / np.linalg.norm(ref)
    finally:
        ref_ctx.close()
    # (measured: 4.7e-7 from pieces, 1.2e-6 from the chunked sgemm path)
    assert err < 1e-6 and err < 1.5 * err_sgemm, (err, err_sgemm)
    assert not torch.equal(c, c2), "the two contexts were meant to take different paths"


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm_long_inner_dimension_split(gpu_ctx, ta, tb):
    """pmd_gemm with few output tiles and a very long inner dimension takes the split-K path (strided-batched slices +
    fixed-order sum, global.hip): same result as one product, for every operand orientation, beta != 0 and a ragged
    last slice."""
    torch = _t()
    ctx = gpu_ctx
    m, n, k = 200, 333, 70001
    g = torch.Generator(device="cuda").manual_seed(ta * 2 + tb)
    A = torch.randn((k, m) if ta else (m, k), device=ctx.device, generator=g)
    B = torch.randn((n, k) if tb else (k, n), device=ctx.device, generator=g)
    C0 = torch.randn((m, n + 5), device=ctx.device, generator=g)
    C = C0.clone()
    ctx.call("pmd_gemm", ta, tb, m, n, k, 0.5, P(A), A.shape[1], P(B), B.shape[1], 2.0, P(C), n + 5)
    ctx.sync()
    ref = 0.5 * ((A.T if ta else A).double() @ (B.T if tb else B).double()) + 2.0 * C0[:, :n].double()
    err = (C[:, :n].double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err
    assert torch.equal(C[:, n:], C0[:, n:])


def test_comm_entry_points_single_rank(gpu_ctx):
    """pmd_comm_* (RCCL at the C ABI): a one-rank communicator on the test box's only GPU - unique id,
    init, in-place all-reduce and all-gather on the context's stream, destroy.  (More ranks need one GPU each; the
    multi-rank data path of the package runs over torch.distributed, tests/test_gpu_distributed.py.)"""
    import ctypes as C

    torch = _t()
    ctx = gpu_ctx
    uid = C.create_string_buffer(128)
    assert ctx.lib.pmd_comm_unique_id(uid) == 0
    ctx.call("pmd_comm_init", uid, 0, 1)
    try:
        x = torch.arange(1000, dtype=torch.float32, device=ctx.device)
        ctx.call("pmd_comm_all_reduce_f32", P(x), x.numel())
        y = torch.empty(4096, dtype=torch.uint8, device=ctx.device)
        src = torch.arange(4096, device=ctx.device).to(torch.uint8)
        ctx.call("pmd_comm_all_gather", P(src), P(y), 4096)
        ctx.sync()
        assert torch.equal(x, torch.arange(1000, dtype=torch.float32, device=ctx.device))
        assert torch.equal(y, src)
        with pytest.raises(Exception):
            ctx.call("pmd_comm_init", uid, 0, 1)     # one communicator per context
    finally:
        ctx.call("pmd_comm_destroy")
