"""
Synthetic calcium-imaging-like movies of the BASELINE shapes:

    Y[t,i,j] = 100 + sum_n A_n(i,j) C_n(t) + 0.5 sin(2 pi t / 1000) ramp(i,j) + N(0,1)

Gaussian footprints (sigma = 3 px, peak 5..20, ~2 sources per 20x20 px), traces = Poisson
spikes (rate 0.01 / frame) convolved with exp(-t/20).  ``make_movie`` is the NumPy
generator (tests, CPU baseline); ``make_movie_torch`` evaluates the same formula on a torch
device in pixel strips so the 10 GB headline movie never exists on the host.
"""
import math

import numpy as np


def _sources(d1, d2, T, rng, density=2.0 / 400.0):
    n = max(1, int(round(d1 * d2 * density)))
    ci = rng.uniform(0, d1, n)
    cj = rng.uniform(0, d2, n)
    peak = rng.uniform(5.0, 20.0, n)
    spikes = (rng.random((n, T)) < 0.01).astype(np.float32)
    # exp(-t/20) kernel via a first-order recursion
    decay = math.exp(-1.0 / 20.0)
    traces = np.empty((n, T), dtype=np.float32)
    acc = np.zeros(n, dtype=np.float32)
    for t in range(T):
        acc = acc * decay + spikes[:, t]
        traces[:, t] = acc
    return ci, cj, peak, traces


def _ladder_sources(d1, d2, T, n, seed, top=57.0, ratio=0.93, smooth=0.0):
    """n bright, spatially separated sources whose singular values form a geometric ladder (ratio between neighbours):
    footprints of sigma 3 px on a jittered grid, zero-mean spike traces scaled to unit RMS, peaks top * ratio^k.  Gives the
    decomposition of a synthetic movie well-separated leading components (parity fixtures: SVD vectors are only
    comparable component by component when their singular values are separated)."""
    rng = np.random.Generator(np.random.PCG64(1000003 * (seed + 1) + n))
    g2 = int(math.ceil(math.sqrt(n * d2 / float(d1))))
    g1 = int(math.ceil(n / float(g2)))
    cells = rng.permutation(g1 * g2)[:n]
    ci = ((cells // g2) + 0.5 + rng.uniform(-0.2, 0.2, n)) * (d1 / float(g1))
    cj = ((cells % g2) + 0.5 + rng.uniform(-0.2, 0.2, n)) * (d2 / float(g2))
    spikes = (rng.random((n, T)) < 0.02).astype(np.float64)
    decay = math.exp(-1.0 / 20.0)
    traces = np.empty((n, T))
    acc = np.zeros(n)
    for t in range(T):
        acc = acc * decay + spikes[:, t]
        traces[:, t] = acc
    if smooth > 0:
        # band-limited traces instead (white noise smoothed by a Gaussian of `smooth` frames, plain np.convolve): nothing
        # above ~0.1 cycles / frame, so the Welch noise estimate of the bright pixels stays at the noise level and the
        # singular values of the standardised movie follow the geometric ladder of the peaks
        half = int(4 * smooth)
        ker = np.exp(-0.5 * (np.arange(-half, half + 1) / smooth) ** 2)
        white = rng.standard_normal((n, T + 2 * half))
        traces = np.stack([np.convolve(white[k], ker, mode="valid") for k in range(n)])
    traces -= traces.mean(axis=1, keepdims=True)
    traces /= np.sqrt((traces ** 2).mean(axis=1, keepdims=True))
    peak = top * ratio ** np.arange(n)
    return ci, cj, peak, traces.astype(np.float32)


def make_movie(T, d1, d2, seed=0, noise=1.0, dtype=np.float32, ladder=0, ladder_top=57.0, ladder_ratio=0.93, ladder_smooth=0.0):
    """ladder = n > 0 adds n bright sources with a geometric ladder of singular values (see _ladder_sources); the rest of
    the movie (sources, background, noise draws) is unchanged by it."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ci, cj, peak, traces = _sources(d1, d2, T, rng)
    ii = np.arange(d1, dtype=np.float32)[:, None]
    jj = np.arange(d2, dtype=np.float32)[None, :]
    foot = np.empty((len(ci), d1, d2), dtype=np.float32)
    for n in range(len(ci)):
        foot[n] = peak[n] * np.exp(-((ii - ci[n]) ** 2 + (jj - cj[n]) ** 2) / (2 * 3.0 ** 2))
    foot[foot < 1e-3 * peak[:, None, None]] = 0
    movie = np.tensordot(traces.T, foot, axes=(1, 0))  # (T, d1, d2)
    ramp = (ii / max(d1 - 1, 1) + jj / max(d2 - 1, 1)).astype(np.float32)
    slow = (0.5 * np.sin(2 * np.pi * np.arange(T) / 1000.0)).astype(np.float32)
    movie += slow[:, None, None] * ramp[None]
    movie += 100.0
    movie += noise * rng.standard_normal((T, d1, d2), dtype=np.float32)
    if ladder > 0:
        li, lj, lpeak, ltr = _ladder_sources(d1, d2, T, int(ladder), seed, ladder_top, ladder_ratio, ladder_smooth)
        lfoot = np.empty((len(li), d1, d2), dtype=np.float32)
        for n in range(len(li)):
            lfoot[n] = lpeak[n] * np.exp(-((ii - li[n]) ** 2 + (jj - lj[n]) ** 2) / (2 * 3.0 ** 2))
        lfoot[lfoot < 1e-3 * lpeak[:, None, None]] = 0
        movie += np.tensordot(ltr.T, lfoot, axes=(1, 0))
    return movie.astype(dtype)


def make_movie_torch(T, d1, d2, device, seed=0, noise=1.0, strip=64, rows=None):
    """Same model on ``device`` (float32, (T, d1, d2)); sources drawn on host with the same
    generator as ``make_movie``; noise from torch's device generator, re-seeded per strip of 64 FOV rows so
    that any band of rows can be generated on its own: rows=(i_lo, i_hi) returns exactly
    make_movie_torch(...)[:, i_lo:i_hi, :] without building the rest."""
    import torch

    rng = np.random.Generator(np.random.PCG64(seed))
    ci, cj, peak, traces = _sources(d1, d2, T, rng)
    g = torch.Generator(device=device)
    r_lo, r_hi = (0, d1) if rows is None else (int(rows[0]), int(rows[1]))
    tr = torch.from_numpy(traces).to(device)  # (n, T)
    ci_t = torch.from_numpy(ci.astype(np.float32)).to(device)
    cj_t = torch.from_numpy(cj.astype(np.float32)).to(device)
    pk_t = torch.from_numpy(peak.astype(np.float32)).to(device)
    slow = 0.5 * torch.sin(2 * math.pi * torch.arange(T, device=device, dtype=torch.float32) / 1000.0)
    jj = torch.arange(d2, device=device, dtype=torch.float32)
    movie = torch.empty((T, r_hi - r_lo, d2), device=device, dtype=torch.float32)
    for i0 in range(0, d1, strip):
        i1 = min(d1, i0 + strip)
        if i1 <= r_lo or i0 >= r_hi:
            continue
        g.manual_seed(int(seed) * 1000003 + i0)
        ii = torch.arange(i0, i1, device=device, dtype=torch.float32)
        near = ((ci_t > i0 - 12) & (ci_t < i1 + 12)).nonzero().flatten()
        di = ii[None, :, None] - ci_t[near, None, None]
        dj = jj[None, None, :] - cj_t[near, None, None]
        foot = pk_t[near, None, None] * torch.exp(-(di * di + dj * dj) / (2 * 3.0 ** 2))
        foot = torch.where(foot < 1e-3 * pk_t[near, None, None], torch.zeros_like(foot), foot)
        block = torch.matmul(tr[near].T, foot.reshape(len(near), -1)).reshape(T, i1 - i0, d2)
        ramp = ii[:, None] / max(d1 - 1, 1) + jj[None, :] / max(d2 - 1, 1)
        block += slow[:, None, None] * ramp[None]
        block += 100.0
        block += noise * torch.randn(block.shape, device=device, dtype=torch.float32, generator=g)
        a, b = max(i0, r_lo), min(i1, r_hi)
        movie[:, a - r_lo:b - r_lo, :] = block[:, a - i0:b - i0, :]
        del block, foot, di, dj
    return movie


class SyntheticSlabSource:
    """The synthetic movie as a dataset that never exists as a whole: ``shape`` plus ``slab(i_lo, i_hi)``,
    which builds the FOV rows [i_lo, i_hi) on the device.  localmd_decomposition(distributed=True) asks every
    rank only for the slab of its band of tile rows (BASELINE configs 4 and 5 do not fit one GPU otherwise)."""

    def __init__(self, T, d1, d2, device, seed=0, noise=1.0):
        self.shape = (int(T), int(d1), int(d2))
        self.device, self.seed, self.noise = device, seed, noise
        self._cache = None

    def slab(self, i_lo, i_hi):
        """The rows stay resident after the first request (the data set of a benchmark lives in HBM)."""
        key = (int(i_lo), int(i_hi))
        if self._cache is None or self._cache[0] != key:
            T, d1, d2 = self.shape
            self._cache = (key, make_movie_torch(T, d1, d2, self.device, seed=self.seed, noise=self.noise, rows=key))
        return self._cache[1]
