"""
Known-answer tests that pin the CPU oracle without the reference:
closed forms, an independent Welch implementation (scipy.signal.welch, which
jax.scipy.signal.welch mirrors), the committed golden fixture, and structural identities.
CPU only.
"""
import os

import numpy as np
import pytest
import scipy.signal

from oracle import pmd_oracle as O, philox

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "oracle_small.npz")


def test_filter_by_failures_truth_table():
    f = O.filter_by_failures
    np.testing.assert_array_equal(f(np.array([True, True, False, True]), 1), [True, True, True, False])
    np.testing.assert_array_equal(f(np.array([False, True, True]), 1), [True, False, False])
    np.testing.assert_array_equal(f(np.array([True, False, True, False, False, True]), 2), [True, True, True, True, True, False])
    np.testing.assert_array_equal(f(np.array([True, True]), 1), [True, True])


def test_tile_grid_matches_appendix_c():
    assert [len(x) for x in O.tile_grid((512, 512), (20, 20))] == [51, 51]
    assert O.tile_grid((512, 512), (20, 20))[0][-1] == 492
    assert [len(x) for x in O.tile_grid((256, 256), (20, 20))] == [25, 25]
    assert [len(x) for x in O.tile_grid((1024, 1024), (32, 32))] == [63, 63]
    assert [len(x) for x in O.tile_grid((2048, 2048), (16, 16))] == [255, 255]
    assert [len(x) for x in O.tile_grid((60, 80), (20, 20))] == [5, 7]
    assert [len(x) for x in O.tile_grid((150, 150), (32, 32))] == [9, 9]
    assert [len(x) for x in O.tile_grid((150, 150), (28, 28))] == [10, 10]
    assert [len(x) for x in O.tile_grid((150, 150), (40, 40))] == [7, 7]
    assert O.tile_grid((20, 20), (20, 20)) == ([0], [0])


def test_block_weights_partition_of_unity():
    bw = O.block_weight_matrix((20, 20))
    assert bw.min() == 1 and bw.max() == 10 and bw[9, 9] == 10 and bw[0, 5] == 1
    np.testing.assert_array_equal(bw, bw[::-1, :])
    np.testing.assert_array_equal(bw, bw[:, ::-1])
    with pytest.raises(ValueError):
        O.block_weight_matrix((21, 20))


def test_welch_matches_scipy():
    rng = np.random.default_rng(0)
    for n in (1024, 784, 256):
        x = (100 + rng.standard_normal((7, n)) * np.linspace(0.5, 3, 7)[:, None]).astype(np.float32)
        mine = O.welch_psd(x)
        _, ref = scipy.signal.welch(x.astype(np.float64), noverlap=128, axis=-1)
        # DC/low bins carry fp32 detrending residue of the 100-count baseline; bins 65..128 are what the path uses
        np.testing.assert_allclose(mine[:, 2:], ref[:, 2:], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(mine[:, :2], ref[:, :2], rtol=2e-4, atol=2e-5)


def test_welch_noise_recovers_white_noise_sigma():
    rng = np.random.default_rng(1)
    for sigma in (0.5, 1.0, 4.0):
        x = (50 + sigma * rng.standard_normal((64, 4096))).astype(np.float32)
        est = O.get_noise_estimate_vmap(x)
        assert abs(est.mean() / sigma - 1) < 0.03


def test_roughness_closed_forms():
    i = np.arange(12, dtype=np.float32)[:, None] + np.zeros((1, 9), dtype=np.float32)
    img = 2.0 * i + 1.0  # vertical gradient 2, horizontal 0
    n_v, n_h = 11 * 9, 12 * 8
    expect = (2.0 * n_v) / (n_v + n_h) / np.mean(np.abs(img))
    assert abs(O.spatial_roughness_stat(img) - expect) < 1e-6
    t = np.arange(50, dtype=np.float32)
    assert O.temporal_roughness_stat(3 * t + 2) == 0
    alt = np.where(np.arange(50) % 2 == 0, 1.0, -1.0).astype(np.float32)
    assert abs(O.temporal_roughness_stat(alt) - 4.0) < 1e-6


def test_pooling_same_padding():
    x = np.arange(5 * 4 * 2, dtype=np.float32).reshape(5, 4, 2)
    out = O.downsample_average_pooling(x, 2)
    assert out.shape == (3, 2, 2)
    np.testing.assert_allclose(out[0, 0], x[0:2, 0:2].mean(axis=(0, 1)))
    np.testing.assert_allclose(out[2, 1], x[4:5, 2:4].mean(axis=(0, 1)))  # clipped window: true count divisor
    y = np.arange(6 * 6, dtype=np.float32).reshape(6, 6, 1)
    np.testing.assert_allclose(O.downsample_average_pooling(y, 2)[1, 2, 0], y[2:4, 4:6, 0].mean())


def test_svd_hermitian_convention():
    rng = np.random.default_rng(2)
    a = rng.standard_normal((6, 6))
    a = ((a + a.T) / 2).astype(np.float32)
    u, s, vh = O.svd_hermitian(a)
    assert np.all(np.diff(s) <= 0) and np.all(s >= 0)
    np.testing.assert_allclose((u * s) @ vh, a, atol=1e-5)


def test_exact_low_rank_tile_recovery():
    rng = np.random.default_rng(3)
    b1, b2, t, q, r = 20, 20, 400, 4, 6
    foot = np.stack([np.exp(-((np.arange(b1)[:, None] - c1) ** 2 + (np.arange(b2)[None, :] - c2) ** 2) / 18.0)
                     for c1, c2 in [(5, 5), (14, 6), (8, 15), (15, 15)]], axis=2)
    traces = np.cumsum(rng.standard_normal((q, t)), axis=1)
    block = np.tensordot(foot, traces, axes=(2, 0)).astype(np.float32)
    omega = rng.standard_normal((t // 10, r + 10)).astype(np.float32)
    u, good, v, _ = O.single_block_md(block, omega, r, 10, 2, 10.0, 10.0)
    u2 = u.reshape((b1 * b2, r), order="F")[:, :q]
    x2 = block.reshape((b1 * b2, t), order="F")
    assert np.linalg.norm(x2 - u2 @ (u2.T @ x2)) / np.linalg.norm(x2) < 1e-4
    assert np.abs(u2.T @ u2 - np.eye(q)).max() < 1e-5
    np.testing.assert_allclose(np.linalg.norm(v[:q], axis=1), np.linalg.svd(x2, compute_uv=False)[:q], rtol=1e-4)


def test_projected_svd_identity():
    rng = np.random.default_rng(4)
    p = np.linalg.qr(rng.standard_normal((40, 12)))[0].astype(np.float32)
    v = (rng.standard_normal((12, 90)) * np.linspace(10, 1, 12)[:, None]).astype(np.float32)
    r, s, vt = O.projected_svd(p, v)
    np.testing.assert_allclose((r * s) @ vt, p @ v, atol=2e-4)
    assert np.abs(vt @ vt.T - np.eye(12)).max() < 1e-4
    # tall branch
    v2 = v.T.copy()
    p2 = rng.standard_normal((7, 90)).astype(np.float32)
    r2, s2, vt2 = O.projected_svd(p2, v2)
    np.testing.assert_allclose((r2 * s2) @ vt2, p2 @ v2, atol=5e-4)


def test_lowrank_factorized_svd_orthonormalises():
    import scipy.sparse

    rng = np.random.default_rng(5)
    u = scipy.sparse.random(300, 20, density=0.2, random_state=1, format="coo")
    v = rng.standard_normal((20, 50)).astype(np.float32)
    p = O.compute_lowrank_factorized_svd(u, v, only_left=True)
    up = u @ p
    assert np.abs(up.T @ up - np.eye(p.shape[1])).max() < 1e-3
    v_small = rng.standard_normal((20, 8)).astype(np.float32)  # R > frames: right_mat = v branch
    p2 = O.compute_lowrank_factorized_svd(u, v_small, only_left=True)
    assert p2.shape == (20, 8)
    up2 = u @ p2
    assert np.abs(up2.T @ up2 - np.eye(8)).max() < 1e-3


def test_philox_known_answer_and_layout():
    # Random123 known-answer vector for philox4x32-10 (counter = key = 0)
    out = philox.philox4x32_10(np.zeros((1, 4), dtype=np.uint32), np.zeros(2, dtype=np.uint32))
    assert [hex(int(x)) for x in out[0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    ones = np.full((1, 4), 0xFFFFFFFF, dtype=np.uint32)
    out = philox.philox4x32_10(ones, np.full(2, 0xFFFFFFFF, dtype=np.uint32))
    assert [hex(int(x)) for x in out[0]] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    src = philox.PhiloxSource(9)
    z = src.noise(3, 4, 5, 6)
    flat = philox.normals(9, philox.STREAM_SIM_NOISE, 3, 4 * 5 * 6).reshape(20, 6)
    np.testing.assert_array_equal(z[2, 3], flat[2 + 4 * 3])
    big = philox.normals(1, 1, 0, 200000)
    assert abs(big.mean()) < 0.01 and abs(big.std() - 1) < 0.01


def test_end_to_end_oracle_invariants_and_golden():
    from localmd_amd.synthetic import make_movie

    mov = make_movie(400, 30, 36, seed=11)
    np.random.seed(3)
    res = O.localmd_decomposition(mov, (20, 16), 400, max_components=5, background_rank=2,
                                  rng=philox.PhiloxSource(5), sim_iters=8)
    ur = res.u @ res.r
    assert np.abs(ur.T @ ur - np.eye(ur.shape[1])).max() < 1e-3
    assert np.abs(res.v @ res.v.T - np.eye(res.v.shape[0])).max() < 1e-3
    # R diag(s) Vt == P V (projected_svd identity, A.9 vi)
    np.testing.assert_allclose((res.r * res.s) @ res.v, res.diag["p"] @ res.diag["v_proj"], atol=2e-3 * res.s[0])
    # denoised movie is close to the noisy input (noise sigma = 1)
    rec = ((ur * res.s) @ res.v).reshape(30, 36, -1, order="F") * res.std_img[:, :, None] + res.mean_img[:, :, None]
    assert np.std(mov - rec.transpose(2, 0, 1)) < 1.1
    assert np.all(res.diag["tile_ranks"] >= 1)
    with np.load(GOLDEN) as g:
        np.testing.assert_array_equal(res.diag["tile_ranks"], g["tile_ranks"])
        np.testing.assert_array_equal(res.u.indices, g["U_indices"])
        np.testing.assert_array_equal(res.u.indptr, g["U_indptr"])
        np.testing.assert_allclose(res.mean_img, g["mean_img"], rtol=1e-6)
        np.testing.assert_allclose(res.std_img, g["std_img"], rtol=1e-5)
        np.testing.assert_allclose(res.s, g["s"], rtol=1e-3)
        np.testing.assert_allclose(np.array(res.diag["thresholds"]), g["thresholds"], rtol=1e-4)
