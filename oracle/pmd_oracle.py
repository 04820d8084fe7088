"""
TEST INFRASTRUCTURE (oracle/) -- CPU restatement of the reference algorithm.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker / timed CPU baseline.  The product (localmd_amd/) never
imports it.

PARITY UNPINNED: the reference (apasarkar/localmd @ 2025-01-31) cannot be executed in the
build container (``import localmd`` -> ModuleNotFoundError: jax; no network), its tests
assert no numerical result and it ships no golden vectors (SURVEY.md sections 4, 8(c)).
This file therefore restates the reference line by line in NumPy/SciPy -- arrays and products
in fp32 where JAX computes in fp32, factorisations through numpy.linalg (the geqrf/orgqr, gesdd,
syevd family; NB numpy.linalg computes them in DOUBLE for float32 input and rounds the results to
float32 (numpy/linalg/_linalg.py, _commonType), so these steps are the exact-arithmetic limit of
the reference's fp32 LAPACK calls, not a bit-for-bit model of their rounding; measured in
scripts/eig_accuracy_probe.py) -- and is pinned
only by (a) closed-form known-answer tests (tests/test_oracle_known_answers.py),
(b) scipy.signal.welch as an independent implementation of the Welch semantics that
jax.scipy.signal.welch mirrors.  Third-party semantics restated here: jax/jaxlib
(unpinned in /root/reference/setup.py:14-23): jnp.linalg.qr/svd/svd(hermitian=True),
jax.lax.reduce_window SAME padding, jax.scipy.signal.welch.  Gaussian test matrices are
*injected* through a random source (see oracle/philox.py) because jax.random's stream is
not reproducible here.

Every function cites the reference file:line it follows (paths under /root/reference/).
"""
import datetime
import math
import sys

import numpy as np
import scipy.sparse
from scipy.sparse import coo_matrix, hstack, diags

from . import philox

F32 = np.float32
VERBOSE = False
# "double": numpy.linalg (computes in double for float32 input, rounds the result) - the default referee.
# "single": scipy.linalg on float32 arrays (true sgeqrf/sorgqr, sgesdd, ssyevd), the precision jaxlib's CPU LAPACK
# kernels run in for the reference's float32 arrays.
LAPACK_PRECISION = "single" if __import__("os").environ.get("ORACLE_LAPACK", "") == "single" else "double"


class arbiter_precision:
    """``with arbiter_precision():`` every array, product and factorisation of this module runs in float64 (pass
    dtype="float64" to localmd_decomposition inside the block).  This is NOT the reference's arithmetic (fp32
    throughout): it is the exact-arithmetic limit of the reference's ALGORITHM on the same inputs, used by the
    tests to rank two fp32 implementations (the HIP path, this oracle in fp32) by their distance to it."""

    def __enter__(self):
        global F32
        self._saved = F32
        F32 = np.float64
        return self

    def __exit__(self, *exc):
        global F32
        F32 = self._saved
        return False


def _qr(a):
    if LAPACK_PRECISION == "single" and a.dtype == np.float32:
        import scipy.linalg

        return scipy.linalg.qr(a, mode="economic", check_finite=False)
    return np.linalg.qr(a)


def _svd(a):
    if LAPACK_PRECISION == "single" and a.dtype == np.float32:
        import scipy.linalg

        return scipy.linalg.svd(a, full_matrices=False, check_finite=False, lapack_driver="gesdd")
    return np.linalg.svd(a, full_matrices=False)


def _eigh(a):
    if LAPACK_PRECISION == "single" and a.dtype == np.float32:
        import scipy.linalg

        return scipy.linalg.eigh(a, check_finite=False, driver="evd")
    return np.linalg.eigh(a)


def display(msg):
    """localmd/decomposition.py:28-34 (timestamped, flushed print)."""
    if not VERBOSE:
        return
    tag = "[" + datetime.datetime.today().strftime("%y-%m-%d %H:%M:%S") + "]: "
    sys.stdout.write(tag + msg + "\n")
    sys.stdout.flush()


# --------------------------------------------------------------------------------------
# third-party semantics (JAX) restated
# --------------------------------------------------------------------------------------
def svd_hermitian(a: np.ndarray):
    """jnp.linalg.svd(a, hermitian=True): eigh -> sort by |w| descending -> u = v*sign(w).

    Used at decomposition.py:984, :1090, :1129.  Returns (u, s, vh)."""
    w, v = _eigh(a)
    s = np.abs(w)
    idx = np.argsort(s, kind="stable")[::-1]
    s = s[idx]
    v = v[:, idx]
    sign = np.sign(w[idx])
    sign = np.where(sign == 0, 1, sign).astype(a.dtype)
    u = v * sign[None, :]
    return u, s, v.conj().T


def reduce_window_sum_same(array: np.ndarray, n: int) -> np.ndarray:
    """jax.lax.reduce_window(add, window (n,n,1), strides (n,n,1), padding SAME).

    XLA SAME padding: out = ceil(in/stride); pad_total = max((out-1)*stride + window - in, 0);
    pad_low = pad_total // 2."""
    d1, d2, t = array.shape

    def pad_amounts(size):
        out = -(-size // n)
        total = max((out - 1) * n + n - size, 0)
        lo = total // 2
        return out, lo, total - lo

    o1, lo1, hi1 = pad_amounts(d1)
    o2, lo2, hi2 = pad_amounts(d2)
    padded = np.zeros((d1 + lo1 + hi1, d2 + lo2 + hi2, t), dtype=array.dtype)
    padded[lo1 : lo1 + d1, lo2 : lo2 + d2, :] = array
    out = np.zeros((o1, o2, t), dtype=array.dtype)
    # fixed summation order: row offset outer, column offset inner
    for a in range(n):
        for b in range(n):
            out += padded[a : a + o1 * n : n, b : b + o2 * n : n, :]
    return out


def downsample_average_pooling(array: np.ndarray, n: int) -> np.ndarray:
    """decomposition.py:192-232: n x n mean pool, divisor = true in-bounds window count."""
    downsampled = reduce_window_sum_same(array, n)
    count = np.ones((array.shape[0], array.shape[1], 1), dtype=array.dtype)
    divisors = reduce_window_sum_same(count, n)
    return downsampled / divisors


# --------------------------------------------------------------------------------------
# evaluation.py:84-222
# --------------------------------------------------------------------------------------
def spatial_roughness_stat(u: np.ndarray) -> np.floating:
    """evaluation.py:84-111."""
    vert = np.abs(u[1:, :] - u[:-1, :])
    horiz = np.abs(u[:, :-1] - u[:, 1:])
    avg_diff = (np.sum(vert, dtype=u.dtype) + np.sum(horiz, dtype=u.dtype)) / u.dtype.type(
        vert.shape[0] * vert.shape[1] + horiz.shape[0] * horiz.shape[1]
    )
    avg_elem = np.mean(np.abs(u), dtype=u.dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        return avg_diff / avg_elem


def temporal_roughness_stat(v: np.ndarray) -> np.floating:
    """evaluation.py:114-126."""
    left = v[:-2]
    right = v[2:]
    mid = v[1:-1]
    num = np.mean(np.abs(left + right - v.dtype.type(2) * mid), dtype=v.dtype)
    den = np.mean(np.abs(v), dtype=v.dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        return num / den


def spatial_roughness_stat_vmap(u3: np.ndarray) -> np.ndarray:
    """evaluation.py:129 (vmap over axis 2)."""
    return np.array([spatial_roughness_stat(u3[:, :, c]) for c in range(u3.shape[2])], dtype=u3.dtype)


def temporal_roughness_stat_vmap(v2: np.ndarray) -> np.ndarray:
    """evaluation.py:130 (vmap over axis 0)."""
    return np.array([temporal_roughness_stat(v2[c]) for c in range(v2.shape[0])], dtype=v2.dtype)


def construct_final_fitness_decision(images, traces, spatial_threshold, temporal_threshold):
    """evaluation.py:133-192.  images (d1,d2,r), traces (t,r) -> int32 (r,1).
    Also returns the two statistic vectors (diagnostics for the parity margin report)."""
    sp = spatial_roughness_stat_vmap(images)
    tp = temporal_roughness_stat_vmap(np.ascontiguousarray(traces.T))
    good = (sp < F32(spatial_threshold)) & (tp < F32(temporal_threshold))
    return good.astype(np.int32)[:, None], sp, tp


def filter_by_failures(decisions: np.ndarray, max_consecutive_failures: int) -> np.ndarray:
    """evaluation.py:195-222 (in place; a failing component is KEPT until the budget is hit)."""
    number_of_failures = 0
    all_fails = False
    for k in range(decisions.shape[0]):
        if all_fails:
            decisions[k] = False
        elif not decisions[k]:
            number_of_failures += 1
            decisions[k] = 1
            if number_of_failures == max_consecutive_failures:
                all_fails = True
        else:
            number_of_failures = 0
    return decisions


# --------------------------------------------------------------------------------------
# preprocessing_utils.py:10-40
# --------------------------------------------------------------------------------------
def _hann_periodic(n: int) -> np.ndarray:
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(F32)


def welch_psd(traces: np.ndarray, nperseg: int = 256, noverlap: int = 128) -> np.ndarray:
    """jax.scipy.signal.welch(x, noverlap=128) defaults restated: fs=1, periodic Hann,
    nperseg=256, detrend='constant' per segment, density scaling, one-sided, mean average.
    traces (n, T) float32 -> Pxx (n, nperseg//2+1) float32."""
    n, T = traces.shape
    nperseg = min(nperseg, T)
    step = nperseg - noverlap
    nseg = (T - noverlap) // step
    win = _hann_periodic(nperseg)
    scale = F32(1.0) / (F32(1.0) * np.sum(win * win, dtype=F32))
    acc = np.zeros((n, nperseg // 2 + 1), dtype=F32)
    for sidx in range(nseg):
        seg = traces[:, sidx * step : sidx * step + nperseg].astype(F32)
        seg = seg - np.mean(seg, axis=1, keepdims=True, dtype=F32)
        seg = seg * win[None, :]
        spec = np.fft.rfft(seg, axis=1).astype(np.complex64)
        p = (spec.real * spec.real + spec.imag * spec.imag).astype(F32) * scale
        if nperseg % 2 == 0:
            p[:, 1:-1] *= F32(2.0)
        else:
            p[:, 1:] *= F32(2.0)
        acc += p
    return acc / F32(nseg)


def get_noise_estimate_vmap(traces: np.ndarray) -> np.ndarray:
    """preprocessing_utils.py:28-40: sqrt(mean(0.5*Pxx[65:129]))."""
    pxx = welch_psd(traces)
    start = int(256 / 4 + 1)
    end = int(256 / 2 + 1)
    values = pxx[:, start:end] * F32(0.5)
    return np.sqrt(np.sum(values, axis=1, dtype=F32) / F32(end - start))


def get_mean_and_noise(movie: np.ndarray, mean_divisor) -> tuple:
    """preprocessing_utils.py:10-20.  movie (d1,d2,T) float32."""
    sum_val = np.sum(movie, axis=2, dtype=F32) / F32(mean_divisor)
    d1, d2, T = movie.shape
    movie_2d = np.reshape(movie, (d1 * d2, T), order="F")
    noise_1d = get_noise_estimate_vmap(movie_2d)
    return sum_val, np.reshape(noise_1d, (d1, d2), order="F")


def get_mean_chunk(movie: np.ndarray, mean_divisor) -> np.ndarray:
    """preprocessing_utils.py:23-25."""
    return np.sum(movie, axis=2, dtype=F32) / F32(mean_divisor)


# --------------------------------------------------------------------------------------
# rSVD: decomposition.py:37-73 and pmd_loader.py:46-68
# --------------------------------------------------------------------------------------
def truncated_random_svd(input_matrix: np.ndarray, random_data: np.ndarray, rank: int):
    """decomposition.py:59-73.  random_data is the injected (t, rank+10) Gaussian matrix."""
    a = input_matrix.astype(F32, copy=False)
    projected = a @ random_data.astype(F32, copy=False)
    q, _ = _qr(projected)
    b = q.T @ a
    u, s, v = _svd(b)
    u_final = q @ u
    return u_final[:, :rank], s[:rank], v[:rank, :]


def loader_truncated_random_svd(input_matrix, random_data, rank: int):
    """pmd_loader.py:46-68 (returns U and s*V)."""
    a = input_matrix.astype(F32, copy=False)
    projected = a @ random_data.astype(F32, copy=False)
    q, _ = _qr(projected)
    b = q.T @ a
    u, s, v = _svd(b)
    u_final = q @ u
    v = s[:, None] * v
    return u_final[:, :rank], v[:rank, :]


# --------------------------------------------------------------------------------------
# threshold simulation: decomposition.py:76-189
# --------------------------------------------------------------------------------------
def decomposition_no_normalize_approx(block: np.ndarray, random_data: np.ndarray, rank: int):
    """decomposition.py:76-99."""
    d1, d2, t = block.shape
    block_2d = np.reshape(block, (d1 * d2, t), order="F")
    u_mat, s_mat, v_mat = truncated_random_svd(block_2d, random_data, rank)
    v_mat = s_mat[:, None] * v_mat
    u_mat = np.reshape(u_mat, (d1, d2, u_mat.shape[1]), order="F")
    return spatial_roughness_stat_vmap(u_mat), temporal_roughness_stat_vmap(v_mat)


def threshold_heuristic(dimensions, rng, num_comps=1, iters=250, percentile_threshold=5):
    """decomposition.py:147-189.  rng supplies noise(index,d1,d2,t) and omega(stream,index,t,l).
    Returns (spatial_threshold, temporal_threshold, spatial_list, temporal_list)."""
    d1, d2, t = dimensions
    spatial_list, temporal_list = [], []
    for k in range(iters):
        noise = rng.noise(k, d1, d2, t)
        omega = rng.omega(philox.STREAM_SIM_OMEGA, k, t, num_comps + 10)
        x, y = decomposition_no_normalize_approx(noise, omega, num_comps)
        spatial_list.append(x)
        temporal_list.append(y)
    sp = np.array(spatial_list).flatten()
    tp = np.array(temporal_list).flatten()
    return (
        np.percentile(sp, percentile_threshold),
        np.percentile(tp, percentile_threshold),
        sp,
        tp,
    )


# --------------------------------------------------------------------------------------
# per-tile decomposition: decomposition.py:235-525
# --------------------------------------------------------------------------------------
def single_block_md(block, random_data, rank, temporal_avg_factor, spatial_average_factor,
                    spatial_threshold, temporal_threshold, spatial_denoiser=None,
                    temporal_denoiser=None):
    """decomposition.py:276-330.  block (b1,b2,t) float32, t % temporal_avg_factor == 0.
    Returns (u_final (b1,b2,r), good (r,1) int32, v_final (r,t), stats dict)."""
    order = "F"
    block = block.astype(F32, copy=False)
    d1, d2, t = block.shape
    block_downsample = downsample_average_pooling(block, spatial_average_factor)
    d1n, d2n = block_downsample.shape[0], block_downsample.shape[1]
    bdta = np.mean(
        np.reshape(block_downsample, (d1n * d2n, temporal_avg_factor, t // temporal_avg_factor), order=order),
        axis=1, dtype=F32,
    )
    u_mat_downsample = truncated_random_svd(bdta, random_data, rank)[0]
    v_ds = u_mat_downsample.T @ np.reshape(block_downsample, (d1n * d2n, t), order=order)
    if temporal_denoiser is not None:
        v_ds = temporal_denoiser(v_ds)
    v_mat_basis = _svd(v_ds)[2]

    block_2d = np.reshape(block, (d1 * d2, t), order=order)
    spf = block_2d @ v_mat_basis.T
    if spatial_denoiser is not None:
        spf = np.reshape(spf, (d1, d2, v_mat_basis.shape[0]), order=order).transpose(2, 0, 1)
        spf = spatial_denoiser(spf)
        spf = spf.transpose(1, 2, 0).reshape((d1 * d2, v_mat_basis.shape[0]), order=order)

    u_final, _, _ = _svd(spf)
    v_new = u_final.T @ block_2d
    v_left, v_sing, v_right = _svd(v_new)
    u_final = u_final @ v_left
    v_final = v_sing[:, None] * v_right
    u_final = np.reshape(u_final, (d1, d2, u_final.shape[1]), order=order)
    good, sp, tp = construct_final_fitness_decision(u_final, v_final.T, spatial_threshold, temporal_threshold)
    return u_final, good, v_final, {"spatial": sp, "temporal": tp}


def single_residual_block_md(block, existing, random_data, rank, temporal_avg_factor,
                             spatial_threshold, temporal_threshold):
    """decomposition.py:364-387."""
    order = "F"
    block = block.astype(F32, copy=False)
    d1, d2, t = block.shape
    net_comps = existing.shape[2]
    block_2d = np.reshape(block, (d1 * d2, t), order=order)
    existing_2d = np.reshape(existing, (d1 * d2, net_comps), order=order).astype(F32)
    projection = existing_2d @ (existing_2d.T @ block_2d)
    block_2d = block_2d - projection
    block_r = np.reshape(block_2d, (d1 * d2, temporal_avg_factor, t // temporal_avg_factor), order=order)
    block_r_avg = np.mean(block_r, axis=1, dtype=F32)
    u_mat = truncated_random_svd(block_r_avg, random_data, rank)[0]
    v_mat = u_mat.T @ block_2d
    u_mat = np.reshape(u_mat, (d1, d2, u_mat.shape[1]), order=order)
    good, sp, tp = construct_final_fitness_decision(u_mat, v_mat.T, spatial_threshold, temporal_threshold)
    return u_mat, good, v_mat, {"spatial": sp, "temporal": tp}


def get_temporal_projector(spatial_decomposition, block):
    """decomposition.py:390-407 (fp32: jit converts the float64 host array to float32)."""
    d1, d2, r = spatial_decomposition.shape
    t = block.shape[2]
    sdr = np.reshape(spatial_decomposition, (d1 * d2, r), order="F").astype(F32)
    block_r = np.reshape(block, (d1 * d2, t), order="F").astype(F32)
    return sdr.T @ block_r


def windowed_pmd(window_length, block, max_rank, spatial_threshold, temporal_threshold,
                 max_consecutive_failures, temporal_avg_factor, spatial_avg_factor,
                 omega_provider, spatial_denoiser=None, temporal_denoiser=None, diagnostics=None):
    """decomposition.py:455-525.  omega_provider(window_index, rows, cols) -> Gaussian matrix
    (replaces make_jax_random_key at :475)."""
    d1, d2 = block.shape[0], block.shape[1]
    window_range = block.shape[2]
    if window_length > window_range:
        window_length = window_range
    start_points = list(range(0, window_range, window_length))
    if len(start_points) > 0 and start_points[-1] + window_length > window_range:
        start_points[-1] = window_range - window_length

    final_spatial_decomposition = np.zeros((d1, d2, max_rank))
    remaining_components = max_rank
    component_counter = 0

    for widx, k in enumerate(start_points):
        start_value = k
        end_value = start_value + window_length
        subset = block[:, :, start_value:end_value]
        t_w = subset.shape[2]
        if k == 0 or component_counter == 0:
            omega = omega_provider(widx, t_w // temporal_avg_factor, max_rank + 10)
            spatial_comps, decisions, _, st = single_block_md(
                subset, omega, max_rank, temporal_avg_factor, spatial_avg_factor,
                spatial_threshold, temporal_threshold, spatial_denoiser, temporal_denoiser)
        else:
            omega = omega_provider(widx, t_w // temporal_avg_factor, max_rank + 10)
            spatial_comps, decisions, _, st = single_residual_block_md(
                subset, final_spatial_decomposition, omega, max_rank, temporal_avg_factor,
                spatial_threshold, temporal_threshold)
        raw_good = np.array(decisions).flatten() > 0
        decisions = filter_by_failures(raw_good.copy(), max_consecutive_failures)
        if diagnostics is not None:
            diagnostics.append({"window": widx, "good": raw_good, "kept": decisions.copy(),
                                "spatial": st["spatial"], "temporal": st["temporal"]})
        spatial_cropped = spatial_comps[:, :, decisions]
        final_filter_index = min(spatial_cropped.shape[2], remaining_components)
        spatial_cropped = spatial_cropped[:, :, :final_filter_index]
        final_spatial_decomposition[:, :, component_counter:component_counter + spatial_cropped.shape[2]] = spatial_cropped
        component_counter += spatial_cropped.shape[2]
        if component_counter == max_rank:
            break
        else:
            remaining_components = max_rank - component_counter

    final_temporal_decomposition = np.array(get_temporal_projector(final_spatial_decomposition, block))
    final_spatial_decomposition = final_spatial_decomposition[:, :, :component_counter]
    final_temporal_decomposition = final_temporal_decomposition[:component_counter, :]
    return final_spatial_decomposition, final_temporal_decomposition


# --------------------------------------------------------------------------------------
# frame sampling / validation: decomposition.py:528-635
# --------------------------------------------------------------------------------------
def identify_window_chunks(frame_range: int, total_frames: int, window_chunks: int) -> list:
    """decomposition.py:546-569 (uses the global np.random state, like the reference)."""
    if frame_range > total_frames:
        raise ValueError("Requested more frames than available")
    if window_chunks > frame_range:
        raise ValueError("The size of each temporal chunk is bigger than frame range")
    num_intervals = math.ceil(frame_range / window_chunks)
    available_intervals = np.arange(0, total_frames, window_chunks)
    if available_intervals[-1] > total_frames - window_chunks:
        available_intervals[-1] = total_frames - window_chunks
    starting_points = np.random.choice(available_intervals, size=num_intervals, replace=False)
    starting_points = np.sort(starting_points)
    display("sampled from the following regions: {}".format(starting_points))
    net_frames = []
    for k in starting_points:
        net_frames.extend(range(int(k), int(min(k + window_chunks, total_frames))))
    return net_frames


def update_block_sizes(blocks, fov_shape, min_block_value: int = 10) -> list:
    """decomposition.py:589-613."""
    if blocks[0] < min_block_value or blocks[1] < min_block_value:
        raise ValueError(
            "One of the block dimensions was less than min allowed value of {}, "
            "set to a larger value".format(min_block_value))
    return [min(blocks[0], fov_shape[0]), min(blocks[1], fov_shape[1])]


def check_fov_size(fov_dims, min_allowed_value: int = 10) -> None:
    """decomposition.py:630-635."""
    for k in fov_dims:
        if k < min_allowed_value:
            raise ValueError("At least one FOV dimension is lower than {}, "
                             "too small to process".format(min_allowed_value))


def tile_grid(fov, block_sizes):
    """decomposition.py:698, :723-739: tile origins per dimension."""
    overlap = [math.ceil(block_sizes[0] / 2), math.ceil(block_sizes[1] / 2)]
    iters = []
    for dim in (0, 1):
        it = list(range(0, fov[dim] - block_sizes[dim] + 1, block_sizes[dim] - overlap[dim]))
        if it[-1] != fov[dim] - block_sizes[dim] and fov[dim] - block_sizes[dim] != 0:
            it.append(fov[dim] - block_sizes[dim])
        iters.append(it)
    return iters[0], iters[1]


def block_weight_matrix(block_sizes, dtype=F32) -> np.ndarray:
    """decomposition.py:742-750 (pyramid weights; odd sizes raise ValueError in NumPy broadcasting)."""
    bw = np.ones((block_sizes[0], block_sizes[1]), dtype=dtype)
    hbh = block_sizes[0] // 2
    hbw = block_sizes[1] // 2
    bw[:hbh, :hbw] += np.minimum(np.tile(np.arange(0, hbw), (hbh, 1)), np.tile(np.arange(0, hbh), (hbw, 1)).T)
    bw[:hbh, hbw:] = np.fliplr(bw[:hbh, :hbw])
    bw[hbh:, :] = np.flipud(bw[:hbh, :])
    return bw


# --------------------------------------------------------------------------------------
# pmd_loader.py:71-414
# --------------------------------------------------------------------------------------
class FrameDataloader:
    """pmd_loader.py:71-108."""

    def __init__(self, dataset, batch_size: int, dtype="float32"):
        self.dataset = dataset
        self.shape = dataset.shape
        self.chunks = math.ceil(self.shape[0] / batch_size)
        self.batch_size = batch_size
        self.dtype = dtype

    def __len__(self):
        return max(1, self.chunks - 1)

    def __getitem__(self, index: int) -> np.ndarray:
        start = index * self.batch_size
        if index == max(0, self.chunks - 2):
            keys = list(range(start, self.shape[0]))
        elif index < self.chunks - 2:
            keys = list(range(start, start + self.batch_size))
        else:
            raise ValueError
        return self.dataset[keys].astype(self.dtype).transpose(1, 2, 0)


class PMDLoader:
    """pmd_loader.py:111-371.  rng supplies omega(STREAM_BG_OMEGA, 0, n, K+10)."""

    def __init__(self, dataset, rng, dtype="float32", background_rank=15, batch_size=2000,
                 pixel_batch_size=5000, order="F", compute_normalizer=True):
        self.order = order
        self.dataset = dataset
        self.dtype = dtype
        self.shape = dataset.shape
        self.batch_size = batch_size
        self.pixel_batch_size = pixel_batch_size
        self._compute_normalizer = compute_normalizer
        self.curr_dataloader = FrameDataloader(dataset, batch_size, dtype=dtype)
        self.background_rank = background_rank
        self.frame_constant = 1024
        self.rng = rng
        self.mean_img, self.std_img = self._calculate_mean_and_normalizer()
        self.spatial_basis = self._calculate_background_filter()

    def temporal_crop(self, frames):
        """pmd_loader.py:179-188."""
        return self.dataset[frames].astype(self.dtype).transpose(1, 2, 0)

    def _calculate_mean_and_normalizer(self, min_allowed_frames: int = 256):
        """pmd_loader.py:203-291."""
        normalizer_flag = self._compute_normalizer
        if self.shape[0] < min_allowed_frames:
            normalizer_flag = False
        overall_mean = np.zeros((self.shape[1], self.shape[2]), dtype=self.dtype)
        if normalizer_flag:
            overall_normalizer = np.zeros((self.shape[1], self.shape[2]), dtype=self.dtype)
        else:
            overall_normalizer = np.ones((self.shape[1], self.shape[2]), dtype=self.dtype)

        divisor = math.ceil(math.sqrt(self.pixel_batch_size))

        def starts(size):
            if size - divisor <= 0:
                return np.arange(1)
            pts = np.arange(0, size - divisor, divisor)
            return np.concatenate([pts, [size - divisor]], axis=0)

        dim1_pts = starts(self.shape[1])
        dim2_pts = starts(self.shape[2])
        elts_used = list(range(0, self.shape[0], self.frame_constant))
        elts_for_var_est = 0
        for i in elts_used:
            end_pt_frame = min(i + self.frame_constant, self.shape[0])
            data = np.array(self.temporal_crop(list(range(i, end_pt_frame))))
            mean_value_net = np.zeros((self.shape[1], self.shape[2]))
            normalizer_net = np.zeros((self.shape[1], self.shape[2]))
            if data.shape[2] >= min_allowed_frames:
                elts_for_var_est += 1
            for s1 in dim1_pts:
                for s2 in dim2_pts:
                    crop = data[s1:s1 + divisor, s2:s2 + divisor, :]
                    if crop.shape[2] >= min_allowed_frames and normalizer_flag:
                        mv, ne = get_mean_and_noise(crop, self.shape[0])
                        mean_value_net[s1:s1 + divisor, s2:s2 + divisor] = mv
                        normalizer_net[s1:s1 + divisor, s2:s2 + divisor] = ne
                    else:
                        mean_value_net[s1:s1 + divisor, s2:s2 + divisor] = get_mean_chunk(crop, self.shape[0])
            overall_mean += mean_value_net
            if normalizer_flag:
                overall_normalizer += normalizer_net / len(elts_used)
        if normalizer_flag and elts_for_var_est != 0:
            overall_normalizer *= len(elts_used) / elts_for_var_est
            overall_normalizer[overall_normalizer == 0] = 1
        return overall_mean, overall_normalizer

    def temporal_crop_standardized(self, frames):
        """pmd_loader.py:293-298."""
        crop = self.temporal_crop(frames)
        crop -= self.mean_img[:, :, None]
        crop /= self.std_img[:, :, None]
        return crop.astype(self.dtype)

    def _calculate_background_filter(self, n_samples=1000):
        """pmd_loader.py:300-314."""
        if self.background_rank <= 0:
            return np.zeros((self.shape[1] * self.shape[2], 1)).astype(self.dtype)
        sample_list = list(range(0, self.shape[0]))
        random_data = np.random.choice(sample_list, replace=False, size=min(n_samples, self.shape[0])).tolist()
        self.background_frames = random_data
        crop = self.temporal_crop_standardized(random_data)
        omega = self.rng.omega(philox.STREAM_BG_OMEGA, 0, crop.shape[-1], self.background_rank + 10)
        spatial_basis, _ = loader_truncated_random_svd(
            crop.reshape((-1, crop.shape[-1]), order=self.order), omega, self.background_rank)
        return np.array(spatial_basis).astype(self.dtype)

    def temporal_crop_with_filter(self, frames):
        """pmd_loader.py:348-371 + standardize_and_filter :374-389.  Returns float64 arrays."""
        crop = self.temporal_crop(frames)
        basis_r = self.spatial_basis.reshape((self.shape[1], self.shape[2], -1), order=self.order)
        out = np.zeros(crop.shape)
        temporal_basis = np.zeros((basis_r.shape[2], crop.shape[2]))
        num_iters = math.ceil(out.shape[2] / self.batch_size)
        start = 0
        for _ in range(num_iters):
            end_pt = min(crop.shape[2], start + self.batch_size)
            fd, tb = standardize_and_filter(crop[:, :, start:end_pt], self.mean_img, self.std_img, basis_r)
            out[:, :, start:end_pt] = fd
            temporal_basis[:, start:end_pt] = tb
            start += self.batch_size
        return out, temporal_basis

    def v_projection(self, u, spatial_mixing_matrix):
        """pmd_loader.py:316-346 + v_projection_routine :392-414."""
        ut = scipy.sparse.csr_matrix(u.T).astype(F32)
        dense = spatial_mixing_matrix.T.astype(F32)
        mean_r = self.mean_img.reshape((-1, 1), order=self.order).astype(F32)
        std_r = self.std_img.reshape((-1, 1), order=self.order).astype(F32)
        results = []
        for i in range(len(self.curr_dataloader)):
            data = self.curr_dataloader[i]
            data = np.reshape(data, (-1, data.shape[2]), order=self.order).astype(F32)
            centered = (data - mean_r) / std_r
            out = ut @ centered
            results.append(dense @ out)
        return np.concatenate(results, axis=1)


def standardize_and_filter(new_data, mean_img, std_img, spatial_basis):
    """pmd_loader.py:374-389 (fp32)."""
    x = new_data.astype(F32)
    x = x - mean_img.astype(F32)[:, :, None]
    x = x / std_img.astype(F32)[:, :, None]
    d1, d2, t = x.shape
    x2 = np.reshape(x, (d1 * d2, t), order="F")
    sb = np.reshape(spatial_basis.astype(F32), (d1 * d2, spatial_basis.shape[2]), order="F")
    tp = sb.T @ x2
    x2 = x2 - sb @ tp
    return np.reshape(x2, (d1, d2, t), order="F"), tp


# --------------------------------------------------------------------------------------
# global recombination: decomposition.py:912-1137
# --------------------------------------------------------------------------------------
def aggregate_local_and_global_decomposition(u, v, spatial_basis, temporal_basis):
    """decomposition.py:929-933."""
    u_net = hstack([u, coo_matrix(spatial_basis)])
    v_net = np.concatenate([v, temporal_basis], axis=0)
    return u_net, v_net


def fewer_rows_svd_routine(data):
    """decomposition.py:1089-1099."""
    data = data.astype(F32, copy=False)
    v_vt = data @ data.T
    left, vals, _ = svd_hermitian(v_vt)
    sing = np.sqrt(vals)
    divisor = np.where(sing == 0, F32(1), sing)
    right = (left.T @ data) / divisor[:, None]
    return left, sing, right


def fewer_columns_svd_routine(data):
    """decomposition.py:1128-1137."""
    data = data.astype(F32, copy=False)
    vt_v = data.T @ data
    right_t, vals, _ = svd_hermitian(vt_v)
    sing = np.sqrt(vals)
    divisor = np.where(sing == 0, F32(1), sing)
    left = data @ (right_t / divisor[None, :])
    return left, sing, right_t.T


def projected_svd(projection, data):
    """decomposition.py:1042-1060."""
    d1, d2 = data.shape
    if d1 <= d2:
        left, sing, right = fewer_rows_svd_routine(data)
    else:
        left, sing, right = fewer_columns_svd_routine(data)
    left = projection.astype(F32) @ left
    return left, sing, right


def compute_lowrank_factorized_svd(u, v, only_left: bool = False):
    """decomposition.py:974-1010."""
    ut_u = u.T.dot(u)
    if u.shape[1] > v.shape[1]:
        right_mat = v
    else:
        right_mat = np.eye(u.shape[1])
    ut_ur = ut_u.dot(right_mat)
    rtut_ur = (right_mat.T.astype(F32) @ np.asarray(ut_ur).astype(F32))
    eig_vecs, eig_vals, _ = svd_hermitian(rtut_ur)
    good = eig_vals > 0
    eig_vecs = eig_vecs[:, good]
    eig_vals = eig_vals[good]
    smm = np.array(right_mat.astype(F32) @ eig_vecs)
    sing = np.sqrt(eig_vals)
    smm /= sing[None, :]
    if only_left:
        return smm
    new_temporal = smm.T @ np.asarray(ut_u.dot(v)).astype(F32)
    return projected_svd(smm, new_temporal)


# --------------------------------------------------------------------------------------
# driver: decomposition.py:643-909
# --------------------------------------------------------------------------------------
class OracleResult:
    """Fields of the reference's PMDArray (pmdarray.py:44-58) plus per-stage diagnostics."""

    def __init__(self):
        self.diag = {}


def localmd_decomposition(dataset_obj, block_sizes, frame_range, max_components=50,
                          background_rank=15, sim_conf=5, frame_batch_size=10000, dtype="float32",
                          num_workers=0, pixel_batch_size=5000, max_consecutive_failures=1,
                          rank_prune=False, rank_prune_factor=0.33, temporal_avg_factor=10,
                          spatial_avg_factor=2, order="F", window_chunks=None,
                          compute_normalizer=True, pixel_weighting=None, spatial_denoiser=None,
                          temporal_denoiser=None, rng=None, thresholds=None, sim_iters=250):
    """decomposition.py:643-909.  Extra arguments (not in the reference): ``rng`` random source
    (default PhiloxSource(0)), ``thresholds`` (spatial, temporal) to skip the simulation,
    ``sim_iters``."""
    if rng is None:
        rng = philox.PhiloxSource(0)
    res = OracleResult()
    check_fov_size((dataset_obj.shape[1], dataset_obj.shape[2]))
    load_obj = PMDLoader(dataset_obj, rng, dtype=dtype, background_rank=background_rank,
                         batch_size=frame_batch_size, pixel_batch_size=pixel_batch_size, order=order,
                         compute_normalizer=compute_normalizer)
    if window_chunks is None:
        window_chunks = frame_range
    if load_obj.shape[0] < frame_range:
        display("WARNING: Specified using more frames than there are in the dataset.")
        frame_range = load_obj.shape[0]
        frames = list(range(0, load_obj.shape[0]))
        if frame_range <= window_chunks:
            window_chunks = frame_range
    else:
        if frame_range <= window_chunks:
            window_chunks = frame_range
        frames = identify_window_chunks(frame_range, load_obj.shape[0], window_chunks)
    res.diag["frames"] = list(frames)

    block_sizes = update_block_sizes(block_sizes, (dataset_obj.shape[1], dataset_obj.shape[2]))

    if thresholds is None:
        spatial_threshold, temporal_threshold, sp_l, tp_l = threshold_heuristic(
            [block_sizes[0], block_sizes[1], window_chunks], rng, num_comps=1, iters=sim_iters,
            percentile_threshold=sim_conf)
        res.diag["sim_spatial"] = sp_l
        res.diag["sim_temporal"] = tp_l
    else:
        spatial_threshold, temporal_threshold = thresholds
    res.diag["thresholds"] = (float(spatial_threshold), float(temporal_threshold))

    data, temporal_basis_crop = load_obj.temporal_crop_with_filter(frames)
    if pixel_weighting is not None:
        data *= pixel_weighting[:, :, None]

    dim_1_iters, dim_2_iters = tile_grid((data.shape[0], data.shape[1]), block_sizes)
    block_weights = block_weight_matrix(block_sizes, dtype=dtype)

    sparse_indices = np.arange(data.shape[0] * data.shape[1]).reshape(
        (data.shape[0], data.shape[1]), order=load_obj.order)
    column_number = 0
    rows_l, cols_l, vals_l = [], [], []
    cumulative_weights = np.zeros((data.shape[0], data.shape[1]))
    total_temporal_fit = []

    if temporal_avg_factor >= data.shape[2]:
        raise ValueError("Need at least {} frames".format(temporal_avg_factor))
    if data.shape[2] // temporal_avg_factor <= max_components:
        max_components = int(data.shape[2] // temporal_avg_factor)
    crop_avg_constant = (data.shape[2] // temporal_avg_factor) * temporal_avg_factor
    temporal_basis_crop = temporal_basis_crop[:, :crop_avg_constant]

    pairs, tile_ranks, tile_diag, tile_u = [], [], [], []
    n_windows_max = max(1, math.ceil(crop_avg_constant / max(1, min(window_chunks, crop_avg_constant))))
    tile_index = 0
    for k in dim_1_iters:
        for j in dim_2_iters:
            pairs.append((k, j))
            subset = data[k:k + block_sizes[0], j:j + block_sizes[1], :].astype(dtype)
            subset = subset[:, :, :crop_avg_constant]
            diag_list = []

            def omega_provider(widx, rows, cols, _ti=tile_index):
                return rng.omega(philox.STREAM_TILE_OMEGA, _ti * n_windows_max + widx, rows, cols)

            spatial_cropped, temporal_cropped = windowed_pmd(
                window_chunks, subset, max_components, spatial_threshold, temporal_threshold,
                max_consecutive_failures, temporal_avg_factor, spatial_avg_factor,
                omega_provider, spatial_denoiser, temporal_denoiser, diagnostics=diag_list)
            total_temporal_fit.append(temporal_cropped)
            tile_ranks.append(spatial_cropped.shape[2])
            tile_diag.append(diag_list)
            tile_u.append(spatial_cropped.copy())

            spatial_cropped = spatial_cropped * block_weights[:, :, None]
            cumulative_weights[k:k + block_sizes[0], j:j + block_sizes[1]] += block_weights
            r_b = spatial_cropped.shape[2]
            ridx = sparse_indices[k:k + block_sizes[0], j:j + block_sizes[1]][:, :, None] + np.zeros((1, 1, r_b))
            cidx = np.zeros_like(ridx) + np.arange(column_number, column_number + r_b)[None, None, :]
            rows_l.append(ridx.flatten())
            cols_l.append(cidx.flatten())
            vals_l.append(spatial_cropped.flatten())
            column_number += r_b
            tile_index += 1

    res.diag["pairs"] = pairs
    res.diag["tile_ranks"] = np.array(tile_ranks, dtype=np.int32)
    res.diag["tile_diag"] = tile_diag
    res.diag["tile_u"] = tile_u
    res.diag["block_weights"] = block_weights
    res.diag["max_components"] = max_components

    u_r = coo_matrix((np.concatenate(vals_l), (np.concatenate(rows_l), np.concatenate(cols_l))),
                     shape=(data.shape[0] * data.shape[1], column_number))
    v_cropped = np.concatenate(total_temporal_fit, axis=0)

    weight_normalization_diag = np.zeros((data.shape[0] * data.shape[1],))
    weight_normalization_diag[sparse_indices.flatten(order=load_obj.order)] = cumulative_weights.flatten(order=load_obj.order)
    normalizing_weights = diags([(1 / weight_normalization_diag).ravel()], [0])
    u_r = normalizing_weights.dot(u_r)

    u_r, v_cropped = aggregate_local_and_global_decomposition(u_r, v_cropped, load_obj.spatial_basis, temporal_basis_crop)
    res.diag["rank_before"] = u_r.shape[1]
    res.diag["v_cropped"] = v_cropped

    if rank_prune:
        if rank_prune_factor <= 0 or rank_prune_factor > 1:
            raise ValueError("Rank prune factor should be a value in the interval (0, 1]")
        min_dimension = min(u_r.shape[1], v_cropped.shape[1])
        random_mat = rng.omega(philox.STREAM_PRUNE, 0, v_cropped.shape[1], int(min_dimension * rank_prune_factor))
        temporal_mat = np.array(v_cropped.astype(F32) @ random_mat)
        p = compute_lowrank_factorized_svd(u_r, temporal_mat, only_left=True)
    else:
        p = compute_lowrank_factorized_svd(u_r, v_cropped, only_left=True)
    res.diag["p"] = p

    v = load_obj.v_projection(u_r, p)
    res.diag["v_proj"] = v
    r, s, vt = projected_svd(p, v)
    r, s, vt = np.array(r), np.array(s), np.array(vt)
    good_components = s != 0
    r = r[:, good_components]
    s = s[good_components]
    vt = vt[good_components, :]

    res.u = u_r.tocsr()
    res.r, res.s, res.v = r, s, vt
    res.shape = load_obj.shape
    res.order = load_obj.order
    res.mean_img = load_obj.mean_img
    res.std_img = load_obj.std_img
    res.spatial_basis = load_obj.spatial_basis
    return res
