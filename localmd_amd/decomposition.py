"""
localmd_decomposition / compute_lowrank_factorized_svd / projected_svd on MI355X.

Host side of the blockwise-PMD hot path: same call signature, defaults, printed phase
messages and error behaviour as /root/reference/localmd/decomposition.py:643-909, with every
numeric stage executed by libpmd_hip.so (include/pmd_hip.h) on the HIP device.  PyTorch is
used for device memory, streams and H2D/D2H copies only.  There is no CPU fallback.

Additions to the reference signature (keyword-only): ``seed`` (counter-based device RNG seed;
default: one draw from np.random), ``device`` (HIP device index), ``thresholds`` (inject the
two roughness cut-offs instead of simulating them), ``sim_iters``, ``return_diagnostics``,
``orthogonalizer``: how the mixing matrix P with (UP)^T(UP) = I is built when R > frames --
"eigh" is the reference's eigendecomposition (decomposition.py:984-996); "cholesky" uses the
Cholesky factor of the same Gram matrix (P differs by an orthogonal factor that the final SVD
absorbs, so R, s, Vt are the same); "auto" (default) = cholesky with eigh as fallback.
``distributed=True`` (one process per GPU, torch.distributed initialised): every rank owns a band of tile
rows and keeps only the pixel slab those tiles touch (statistics, standardisation, tile fits and the movie
projection run on the slab); sums over pixels (background projection) and over component rows (the two
Gram matrices of the Cholesky route) are all-reduced, per-tile results are gathered, R is collected on rank 0.
Rank 0 returns the PMDArray; on the row-sharded route the other ranks return None.  Random draws (seed,
frame sample, frame windows) are rank 0's.
"""
import datetime
import math
import sys
import os
import time
from typing import Callable, Optional

import numpy as np
import scipy.sparse
from scipy.sparse import coo_matrix

from . import grid
from ._lib import Context, ptr, PMDLibraryError, c_i, C
from .pmdarray import PMDArray
from .parallel import Dist, tile_partition

STREAM_PRUNE = 5
QUIET = False
# Cholesky route, kept null direction: the rotated-in constant vector is carried next to the joint SVD only when its pivot is
# below this fraction of the mean diagonal of C (pure rounding: 1e-10 ... 1e-8 measured); a resolved direction goes through
# the joint SVD like any other.
NULL_PIVOT_REL = 1e-6


def display(msg):
    """Timestamped, flushed phase message (decomposition.py:28-34)."""
    if QUIET:
        return
    tag = "[" + datetime.datetime.today().strftime("%y-%m-%d %H:%M:%S") + "]: "
    sys.stdout.write(tag + msg + "\n")
    sys.stdout.flush()


def _torch():
    import torch

    return torch


def _upload(ctx, arr, np_dtype):
    """Host table -> device tensor.  Large tables go through a page-locked staging buffer of the context and an
    asynchronous copy (released with ctx.release_pinned() after the next synchronisation)."""
    torch = _torch()
    a = np.ascontiguousarray(arr, dtype=np_dtype)
    t = torch.from_numpy(a)
    if a.nbytes < (1 << 18) or not hasattr(ctx, "pinned") or ctx.device.type != "cuda":
        return t.to(ctx.device)
    stage = ctx.pinned(a.nbytes)[:a.nbytes].view(t.dtype).view(t.shape)
    stage.copy_(t)
    return stage.to(ctx.device, non_blocking=True)


def _i32(ctx, arr):
    return _upload(ctx, arr, np.int32)


def _f32(ctx, arr):
    return _upload(ctx, arr, np.float32)


def _round_up(x, m):
    return (x + m - 1) // m * m


def _dbg(name, t):
    """PMD_DEBUG=1: synchronise and report range / finiteness of an intermediate tensor."""
    import os

    if not os.environ.get("PMD_DEBUG"):
        return
    torch = _torch()
    torch.cuda.synchronize()
    tf = t.float() if t.dtype != torch.float64 else t
    finite = bool(torch.isfinite(tf).all())
    print(f"[pmd-debug] {name}: shape={tuple(t.shape)} finite={finite} absmax={float(tf.abs().max()):.4g}", flush=True)


class _Movie:
    """The (T, D) float32 movie resident in HBM, plus the pixel-major standardised copies."""

    def __init__(self, ctx, dataset_obj, frame_batch_size, rows=None, num_workers=0):
        """rows = (i_lo, i_hi): keep only these FOV rows (first spatial axis) - the pixel slab of one rank.
        self.D is the number of resident pixels; pixel c of the slab is FOV pixel c + i_lo * d2 (C order)."""
        torch = _torch()
        self.ctx = ctx
        shape = tuple(int(x) for x in dataset_obj.shape)
        self.T, self.d1, self.d2 = shape
        i_lo, i_hi = (0, self.d1) if rows is None else rows
        self.D = (i_hi - i_lo) * self.d2
        self.owned = True    # False: self.dev aliases the caller's tensor (releasing it frees nothing)
        if hasattr(dataset_obj, "slab"):   # a source that can build a band of FOV rows on the device
            mv = dataset_obj.slab(i_lo, i_hi).to(device=ctx.device, dtype=torch.float32)
            self.dev = mv.reshape(self.T, self.D).contiguous()
        elif isinstance(dataset_obj, torch.Tensor):
            mv = dataset_obj[:, i_lo:i_hi, :].to(device=ctx.device, dtype=torch.float32)
            self.dev = mv.reshape(self.T, self.D).contiguous()
            self.owned = self.dev.data_ptr() != dataset_obj.data_ptr()
        else:
            self.dev = torch.empty((self.T, self.D), dtype=torch.float32, device=ctx.device)
            self._stream_in(dataset_obj, frame_batch_size, i_lo, i_hi, num_workers)
        # one extra block of zero rows: kernels that walk the rows in 1024-blocks may start at any owned offset
        self.rows_alloc = _round_up(self.D, 1024) + 1024

    # Host -> HBM ingestion (reference: FrameDataloader + torch DataLoader workers, pmd_loader.py:71-108,
    # :151-168, reading a lazy_data_loader such as TiffArray, dataset.py:131-181).  Frame batches are read / decoded by
    # worker threads straight into page-locked staging buffers (a ring of STAGE_BUFFERS) and leave by asynchronous DMA
    # on a copy stream, so decoding batch k + 1 overlaps the transfer of batch k; a buffer is reused once the event
    # recorded behind its transfer has completed.
    STAGE_BUFFERS = 3
    STAGE_BYTES = 64 << 20
    _stage_cache = {}     # (frames per buffer, D, pinned) -> ring of staging buffers, kept between calls (page-locking is slow)

    def _stream_in(self, dataset_obj, frame_batch_size, i_lo, i_hi, num_workers):
        torch = _torch()
        from concurrent.futures import ThreadPoolExecutor

        on_gpu = self.ctx.device.type == "cuda"
        frame_bytes = 4 * self.D
        step = max(1, min(int(frame_batch_size), self.STAGE_BYTES // max(frame_bytes, 1), self.T))
        is_array = isinstance(dataset_obj, np.ndarray)
        # num_workers = 0 is in-process, single-threaded loading in the reference (pmd_loader.py:161-168; workers > 0 are
        # separate processes with a dataset copy each).  A user's lazy_data_loader may share a file handle or a decoder
        # that is not re-entrant, so its __getitem__ is called from ONE reader thread unless num_workers > 0 is passed,
        # which then promises a thread-safe __getitem__ (as does a true `thread_safe` attribute of the loader, which the
        # built-in TiffArray has).  Slicing a NumPy array is thread-safe: many copy threads.
        if num_workers and num_workers > 0:
            n_threads = int(num_workers)
        else:
            n_threads = min(32, os.cpu_count() or 1) if (is_array or getattr(dataset_obj, "thread_safe", False)) else 1
        full_rows = (i_lo == 0 and i_hi == self.d1)
        key = (step, self.D, on_gpu)
        if key not in _Movie._stage_cache:
            _Movie._stage_cache.clear()
            _Movie._stage_cache[key] = [torch.empty((step, self.D), dtype=torch.float32, pin_memory=on_gpu) for _ in range(self.STAGE_BUFFERS)]
        stage = _Movie._stage_cache[key]
        stage_np = [b.numpy() for b in stage]
        done = [None] * self.STAGE_BUFFERS
        copy_stream = _side_stream(self.ctx.device) if on_gpu else None
        if on_gpu:
            copy_stream.synchronize()     # the staging ring is kept between calls: no transfer of an earlier call may still read it
            copy_stream.wait_stream(torch.cuda.current_stream(self.ctx.device))   # self.dev may reuse a block still in use

        def fill(buf, t0, t1):
            """frames [t0, t1) -> rows [t0 - base, t1 - base) of staging buffer `buf` (runs on a worker thread)."""
            base = fill.base
            if is_array:
                src = dataset_obj[t0:t1] if full_rows else dataset_obj[t0:t1, i_lo:i_hi, :]
            else:
                # lazy_data_loader.__getitem__ ends with .squeeze() (dataset.py:114): a one-frame batch comes back 2-D
                src = np.asarray(dataset_obj[list(range(t0, t1))]).reshape(t1 - t0, self.d1, self.d2)
                if not full_rows:
                    src = src[:, i_lo:i_hi, :]
            np.copyto(stage_np[buf][t0 - base:t1 - base].reshape(t1 - t0, i_hi - i_lo, self.d2), src, casting="unsafe")

        with ThreadPoolExecutor(max_workers=n_threads) as pool:
            for k, base in enumerate(range(0, self.T, step)):
                buf = k % self.STAGE_BUFFERS
                n = min(step, self.T - base)
                if done[buf] is not None:
                    done[buf].synchronize()          # the transfer that last read this buffer has finished
                fill.base = base
                parts = max(1, min(n_threads, n))
                edges = [base + (n * j) // parts for j in range(parts + 1)]
                for f in [pool.submit(fill, buf, edges[j], edges[j + 1]) for j in range(parts)]:
                    f.result()
                if on_gpu:
                    with torch.cuda.stream(copy_stream):
                        self.dev[base:base + n].copy_(stage[buf][:n], non_blocking=True)
                        done[buf] = torch.cuda.Event()
                        done[buf].record(copy_stream)
                else:
                    self.dev[base:base + n].copy_(stage[buf][:n])
        if on_gpu:
            torch.cuda.current_stream(self.ctx.device).wait_stream(copy_stream)

    def release(self):
        """Drop the raw frames-first copy once every standardised copy has been written (statistics, the background
        sample and the standardised movies are its only readers): 1/3 of the resident bytes of a large movie."""
        self.dev = None

    def standardized(self, frames, mean, std):
        """Pixel-major (rows_alloc x ld) standardised frames; frames=None means all, in order."""
        torch = _torch()
        ctx = self.ctx
        nf = self.T if frames is None else len(frames)
        ld = ctx.lib.pmd_time_ld(nf)
        # the kernel writes every column of the D pixel rows (zeros beyond nf); only the row padding needs a fill
        out = torch.empty((self.rows_alloc, ld), dtype=torch.float32, device=ctx.device)
        if self.rows_alloc > self.D:
            out[self.D:].zero_()
        fr = None if frames is None else _i32(ctx, frames)
        ctx.call("pmd_standardize_transpose", ptr(self.dev), self.D, ptr(fr), nf, ptr(mean), ptr(std), ptr(out), ld)
        return out, ld


def _to_host(t):
    """Device tensor -> NumPy through page-locked memory (PCIe rate instead of pageable-copy rate).
    The array owns its pinned block until it is garbage collected."""
    torch = _torch()
    out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    out.copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return out.numpy()


_SIDE_STREAMS = {}


def _side_stream(device):
    """One extra HIP stream per device for result downloads that overlap compute."""
    torch = _torch()
    key = str(device)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def _sparse_u(ut_host, ranks, pix_f, block_weights, inv_cumw_rows, n_rows):
    """Sparse assembly of decomposition.py:812-853 (COO triplets, weights, row normalisation),
    vectorised.  ut_host: (n_tiles, 64, dpad) float32; pix_f: (n_tiles, d) row ids of the output
    matrix.  Values are float64: (float64(U) * w) * (1/cumw), like the reference."""
    n_tiles, _, _ = ut_host.shape
    d = pix_f.shape[1]
    offsets = np.concatenate([[0], np.cumsum(ranks)]).astype(np.int64)
    total = int(offsets[-1])
    w_flat = block_weights.reshape(-1, order="F").astype(np.float64)
    rows_l, cols_l, vals_l = [], [], []
    for rk in np.unique(ranks):
        if rk == 0:
            continue
        sel = np.nonzero(ranks == rk)[0]
        u = ut_host[sel, :rk, :d].astype(np.float64)            # (m, rk, d)
        vals = u * w_flat[None, None, :]
        rows = np.broadcast_to(pix_f[sel][:, None, :], vals.shape)
        vals = vals * inv_cumw_rows[rows]
        cols = offsets[sel][:, None, None] + np.arange(rk)[None, :, None]
        cols = np.broadcast_to(cols, vals.shape)
        rows_l.append(rows.reshape(-1))
        cols_l.append(cols.reshape(-1))
        vals_l.append(vals.reshape(-1))
    if vals_l:
        rows = np.concatenate(rows_l)
        cols = np.concatenate(cols_l)
        vals = np.concatenate(vals_l)
        keep = vals != 0  # the reference's dia.dot(coo) drops exact zeros
        u_r = coo_matrix((vals[keep], (rows[keep], cols[keep])), shape=(n_rows, total))
    else:
        u_r = coo_matrix((n_rows, total), dtype=np.float64)
    return u_r, offsets


def _virtual_ranks(ranks_dev, nvt):
    """ranks of the blocks of 64 component rows of every tile: tile t with rank k -> min(max(k - 64 h, 0), 64), h < nvt."""
    torch = _torch()
    if nvt == 1:
        return ranks_dev
    h = 64 * torch.arange(nvt, device=ranks_dev.device, dtype=ranks_dev.dtype)
    return torch.clamp(ranks_dev[:, None] - h[None, :], 0, 64).reshape(-1).contiguous()


def _apply_hook(fn, arr):
    """arr: (n_tiles, ...) device view.  The reference calls the denoiser once per block (decomposition.py:300, :310);
    a callable with a true ``batched`` attribute receives all tiles at once instead."""
    torch = _torch()
    if getattr(fn, "batched", False):
        out = fn(arr.contiguous())
        if tuple(out.shape) != tuple(arr.shape):
            raise ValueError("denoiser returned shape {} for input {}".format(tuple(out.shape), tuple(arr.shape)))
        arr.copy_(torch.as_tensor(out, dtype=torch.float32, device=arr.device))
        return
    for i in range(arr.shape[0]):
        out = fn(arr[i].contiguous())
        if tuple(out.shape) != tuple(arr[i].shape):
            raise ValueError("denoiser returned shape {} for input {}".format(tuple(out.shape), tuple(arr[i].shape)))
        arr[i].copy_(torch.as_tensor(out, dtype=torch.float32, device=arr.device))


def _tiles_decompose(ctx, args, ws, hook_geom, temporal_denoiser, spatial_denoiser):
    """pmd_tiles_decompose, or its three resumable parts around the denoiser hooks of single_block_md
    (decomposition.py:300 temporal_denoiser on V_ds (r, t); :304-313 spatial_denoiser on S as (r, b1, b2)).
    The hooks get float32 device tensors (torch) and must return tensors of the same shape."""
    torch = _torch()
    if temporal_denoiser is None and spatial_denoiser is None:
        ctx.call("pmd_tiles_decompose", *args, ptr(ws), ws.numel())
        return
    n, b1, b2, P_pool, r, a_f, t_crop, ldv, n_rows, dpad = hook_geom
    rp = ctx.lib.pmd_tile_rpad(r)
    import ctypes as C

    off_v, off_s = C.c_size_t(0), C.c_size_t(0)
    rc = ctx.lib.pmd_tiles_hook_offsets(n, b1, b2, P_pool, r, a_f, t_crop, ldv, n_rows, C.byref(off_v), C.byref(off_s))
    if rc != 0:
        raise ValueError("pmd_tiles_hook_offsets failed ({})".format(rc))
    flat = ws.view(torch.uint8).reshape(-1)
    r = min(r, t_crop // a_f, P_pool, r + 10)   # the components that exist (pmd_tiles_decompose: bins / pooled pixels may be fewer)
    vds = flat[off_v.value:off_v.value + n * rp * ldv * 4].view(torch.float32).view(n, rp, ldv)[:, :r, :t_crop]
    s_arr = flat[off_s.value:off_s.value + n * rp * dpad * 4].view(torch.float32).view(n, rp, dpad)[:, :r, :b1 * b2]
    ctx.call("pmd_tiles_decompose_staged", *args, ptr(ws), ws.numel(), 1)
    if temporal_denoiser is not None:
        _apply_hook(temporal_denoiser, vds)
    ctx.call("pmd_tiles_decompose_staged", *args, ptr(ws), ws.numel(), 2)
    if spatial_denoiser is not None:
        # tile pixel q = il + b1 * jl  ->  (r, b1, b2) as the reference's F-order reshape + transpose(2, 0, 1)
        _apply_hook(spatial_denoiser, s_arr.view(n, r, b2, b1).transpose(2, 3))
    ctx.call("pmd_tiles_decompose_staged", *args, ptr(ws), ws.numel(), 4)


def _orthogonalize(ctx, G, R, M, m, ldm):
    """Device A15.  Returns (P tensor (R, ldp), R')."""
    torch = _torch()
    mm = m if M is not None else R
    ldp = mm
    P = torch.empty((R, ldp), dtype=torch.float32, device=ctx.device)
    ws_bytes = ctx.lib.pmd_orthogonalize_workspace_bytes(R, mm, 1 if M is not None else 0)
    ws = ctx.workspace(ws_bytes)
    rp = c_i(0)
    ctx.call("pmd_orthogonalize", ptr(G), R, ptr(M), mm, ldm, ptr(P), ldp, C.byref(rp), ptr(ws), ws.numel())
    return P, int(rp.value)


def _projected_svd_dev(ctx, P, rows_p, ldp, V, n1, n2, ldv):
    torch = _torch()
    nk = min(n1, n2)
    R_out = torch.empty((rows_p, nk), dtype=torch.float32, device=ctx.device)
    s_out = torch.empty((nk,), dtype=torch.float32, device=ctx.device)
    Vt_out = torch.empty((nk, n2), dtype=torch.float32, device=ctx.device)
    ws_bytes = ctx.lib.pmd_projected_svd_workspace_bytes(rows_p, n1, n2)
    ws = ctx.workspace(ws_bytes)
    ctx.call("pmd_projected_svd", ptr(P), rows_p, ldp, ptr(V), n1, n2, ldv, ptr(R_out), nk, ptr(s_out), ptr(Vt_out), n2,
             ptr(ws), ws.numel())
    return R_out, s_out, Vt_out


def projected_svd(projection, data, device=None):
    """(projection @ left, s, right) for the SVD of ``data`` via its Gram matrix
    (decomposition.py:1013-1060).  Host arrays in, host arrays out; computed on the device."""
    torch = _torch()
    ctx = Context(0 if device is None else device)
    try:
        d1, d2 = data.shape
        if d1 <= d2:
            display("Short matrix, using leftward SVD routine")
        else:
            display("Tall matrix, using rightward SVD routine")
        P = _f32(ctx, np.asarray(projection))
        V = _f32(ctx, np.asarray(data))
        R_out, s_out, Vt_out = _projected_svd_dev(ctx, P, P.shape[0], P.shape[1], V, d1, d2, d2)
        ctx.sync()
        return R_out.cpu().numpy(), s_out.cpu().numpy(), Vt_out.cpu().numpy()
    finally:
        ctx.close()


def compute_lowrank_factorized_svd(u: coo_matrix, v: np.ndarray, only_left: bool = False, device=None):
    """Orthonormalising mixing matrix of the factorisation u @ v (decomposition.py:936-1010).
    ``u`` is a general scipy sparse matrix here, so U^T U is formed with scipy (as the reference
    does, :974) and uploaded; the eigendecomposition and all dense products run on the device."""
    torch = _torch()
    ctx = Context(0 if device is None else device)
    try:
        u = scipy.sparse.csr_matrix(u)
        R = u.shape[1]
        G = _f32(ctx, np.asarray((u.T.dot(u)).todense()))
        v32 = np.ascontiguousarray(v, dtype=np.float32)
        M = _f32(ctx, v32) if R > v.shape[1] else None
        P, rp = _orthogonalize(ctx, G, R, M, v.shape[1], v.shape[1])
        P = P[:, :rp].contiguous()
        if only_left:
            ctx.sync()
            return P.cpu().numpy()
        utuv = _f32(ctx, np.asarray(u.T.dot(u).dot(v)))
        nt = torch.empty((rp, v.shape[1]), dtype=torch.float32, device=ctx.device)
        ctx.call("pmd_gemm", 1, 0, rp, v.shape[1], R, 1.0, ptr(P), rp, ptr(utuv), v.shape[1], 0.0, ptr(nt), v.shape[1])
        R_out, s_out, Vt_out = _projected_svd_dev(ctx, P, R, rp, nt, rp, v.shape[1], v.shape[1])
        ctx.sync()
        return R_out.cpu().numpy(), s_out.cpu().numpy(), Vt_out.cpu().numpy()
    finally:
        ctx.close()


def localmd_decomposition(
    dataset_obj,
    block_sizes: tuple,
    frame_range: int,
    max_components: int = 50,
    background_rank: int = 15,
    sim_conf: int = 5,
    frame_batch_size: int = 10000,
    dtype: str = "float32",
    num_workers: int = 0,
    pixel_batch_size: int = 5000,
    max_consecutive_failures=1,
    rank_prune: bool = False,
    rank_prune_factor: float = 0.33,
    temporal_avg_factor: int = 10,
    spatial_avg_factor: int = 2,
    order: str = "F",
    window_chunks: Optional[int] = None,
    compute_normalizer: bool = True,
    pixel_weighting: Optional[np.ndarray] = None,
    spatial_denoiser: Optional[Callable] = None,
    temporal_denoiser: Optional[Callable] = None,
    *,
    seed: Optional[int] = None,
    device: Optional[int] = None,
    thresholds=None,
    sim_iters: int = 250,
    orthogonalizer: str = "auto",
    null_directions: str = "keep",
    null_cutoff: float = 0.0,
    single_copy: Optional[bool] = None,
    tile_batch_bytes: int = 24 << 30,
    distributed: bool = False,
    return_diagnostics: bool = False,
    ctx: Optional[Context] = None,
):
    torch = _torch()
    if np.dtype(dtype) != np.float32:
        raise ValueError("only dtype='float32' is supported (the reference computes in float32 throughout)")
    for hook in (spatial_denoiser, temporal_denoiser):
        if hook is not None and not callable(hook):
            raise TypeError("spatial_denoiser / temporal_denoiser must be callables on torch device tensors")
    if order not in ("F", "C"):
        raise ValueError("order must be 'F' or 'C'")
    if orthogonalizer not in ("auto", "eigh", "cholesky"):
        raise ValueError("orthogonalizer must be 'auto', 'eigh' or 'cholesky'")
    if null_directions not in ("keep", "drop"):
        raise ValueError("null_directions must be 'keep' (the reference's rule) or 'drop'")
    timings = {}
    t_start = time.perf_counter()

    def lap(name, t0):
        if return_diagnostics:
            ctx.sync()
        timings[name] = timings.get(name, 0.0) + time.perf_counter() - t0

    T, d1, d2 = (int(x) for x in dataset_obj.shape)
    grid.check_fov_size((d1, d2))
    own_ctx = ctx is None
    if own_ctx:
        ctx = Context(0 if device is None else device)
    try:
        lib = ctx.lib
        # decomposition.py:984-996 keeps every direction of the orthogonalising eigendecomposition whose eigenvalue is
        # not exactly zero ("keep"); "drop" removes lambda <= null_cutoff * lambda_max and the known null direction
        ctx.call("pmd_ctx_set_null_cutoff", -1.0 if null_directions == "keep" else float(null_cutoff))
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1))
        seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        D = d1 * d2

        # ---- multi-GPU ownership (distributed=True): rank r decomposes a band of tile rows and keeps only the
        # FOV rows those tiles touch (its pixel slab, P_lo <= c < P_hi in C-order pixel ids); every pixel is OWNED
        # by exactly one rank (O_lo <= c < O_hi, the slab minus the halo shared with the next rank) for the sums
        # over pixels.  Single process: slab = owned range = the whole field of view.
        dist = Dist(distributed)
        seed = dist.broadcast_object(seed)   # drawn from np.random when not given: rank 0's draw counts
        geo_blocks = grid.update_block_sizes(list(block_sizes), (d1, d2), display=None)
        geo_o1, geo_o2 = grid.tile_origins((d1, d2), geo_blocks)
        n1_geo, n2_geo = len(geo_o1), len(geo_o2)
        if dist.enabled and dist.world > n1_geo:
            raise ValueError("distributed=True needs at least one tile row per rank ({} rows, {} ranks)".format(n1_geo, dist.world))
        row_runs = tile_partition(n1_geo, dist.world)
        runs = [(a * n2_geo, b * n2_geo) for a, b in row_runs]
        a_lo, a_hi = row_runs[dist.rank]
        i_lo, i_hi = (int(geo_o1[a_lo]), int(geo_o1[a_hi - 1]) + int(geo_blocks[0])) if dist.enabled else (0, d1)
        own_starts = [int(geo_o1[a]) for a, _ in row_runs] + [d1]
        own_starts[0] = 0
        owned = [(own_starts[k] * d2, own_starts[k + 1] * d2) for k in range(dist.world)]
        P_lo, P_hi = i_lo * d2, i_hi * d2
        O_lo, O_hi = owned[dist.rank] if dist.enabled else (0, D)

        # ---- PMDLoader.__init__ (pmd_loader.py:112-173): movie to HBM, statistics, background basis
        t0 = time.perf_counter()
        movie = _Movie(ctx, dataset_obj, frame_batch_size, rows=(i_lo, i_hi) if dist.enabled else None, num_workers=num_workers)
        Dl = movie.D   # resident pixels (= D unless distributed)
        lap("upload", t0)
        display("Computing Video Statistics")
        if compute_normalizer:
            display("We are normalizing each pixel by a noise variance estimate")
        else:
            display("We are not normalizing each pixel by a noise variance estimate")
        display("Calculating mean and noise variance")
        t0 = time.perf_counter()
        mean_dev = torch.empty(Dl, dtype=torch.float32, device=ctx.device)
        std_dev = torch.empty(Dl, dtype=torch.float32, device=ctx.device)
        ws = ctx.workspace(lib.pmd_stats_workspace_bytes(T, Dl, 1024))
        ctx.call("pmd_stats", ptr(movie.dev), T, Dl, 1024, 1 if compute_normalizer else 0, ptr(mean_dev), ptr(std_dev),
                 ptr(ws), ws.numel())
        display("Finished mean and noise variance")
        lap("stats", t0)

        t0 = time.perf_counter()
        K = int(background_rank)
        basis_dev = None
        if K > 0:
            sample = dist.broadcast_object(np.random.choice(list(range(T)), replace=False, size=min(1000, T)).tolist())
            xs_s, ld_s = movie.standardized(sample, mean_dev, std_dev)
            if dist.enabled:
                # the (<= 1000 frame) sample is small: every rank contributes its owned pixels, all get the whole
                # sample and compute the same basis (same counter-based test matrix)
                xs_all = torch.zeros((_round_up(D, 1024) + 1024, ld_s), dtype=torch.float32, device=ctx.device)
                xs_all[O_lo:O_hi] = xs_s[O_lo - P_lo:O_hi - P_lo]
                dist.gather_runs(xs_all, owned)
                xs_s = xs_all
            basis_dev = torch.empty((D, K), dtype=torch.float32, device=ctx.device)
            ws = ctx.workspace(lib.pmd_background_rsvd_workspace_bytes(D, len(sample), K))
            ctx.call("pmd_background_rsvd", ptr(xs_s), D, len(sample), ld_s, K, seed, ptr(basis_dev), ptr(ws), ws.numel())
            del xs_s
        lap("background", t0)

        # ---- frames to fit on (decomposition.py:678-693)
        if window_chunks is None:
            window_chunks = frame_range
        if T < frame_range:
            display("WARNING: Specified using more frames than there are in the dataset.")
            frame_range = T
            frames = list(range(T))
            if frame_range <= window_chunks:
                window_chunks = frame_range
        else:
            if frame_range <= window_chunks:
                window_chunks = frame_range
            frames = dist.broadcast_object(grid.identify_window_chunks(frame_range, T, window_chunks, display=display))
        display("We are initializing on a total of {} frames".format(len(frames)))
        Tf = len(frames)

        block_sizes = grid.update_block_sizes(block_sizes, (d1, d2), display=display)
        b1, b2 = block_sizes
        d = b1 * b2
        block_weights = grid.block_weight_matrix(block_sizes, dtype=np.float32)

        # ---- roughness thresholds (decomposition.py:700-711)
        display("Running Simulations, block dimensions are {} x {} x {} ".format(b1, b2, window_chunks))
        t0 = time.perf_counter()
        sim_stats = None
        sim_pending = None
        if thresholds is None:
            # The simulation depends on nothing but the block shape: it runs on the second stream, next to the
            # HBM-bound standardise / filter passes below; its result is fetched right before the tile stage.
            sc = ctx.side()
            with torch.cuda.stream(sc.stream):
                ws_sim = sc.workspace(lib.pmd_threshold_sim_workspace_bytes(b1, b2, int(window_chunks), int(sim_iters)))
                sim_dev = torch.empty((sim_iters, 2), dtype=torch.float32, device=ctx.device)
                sc.call("pmd_threshold_sim", b1, b2, int(window_chunks), int(sim_iters), seed, ptr(sim_dev), ptr(ws_sim),
                        ws_sim.numel())
            sim_pending = (sc, sim_dev)
        else:
            spatial_threshold, temporal_threshold = thresholds
        lap("simulation", t0)

        # ---- standardise, background-filter, transpose (pmd_loader.py:348-389)
        display("Loading Data")
        t0 = time.perf_counter()
        all_frames = frames == list(range(T))
        xs_full, ld_T = movie.standardized(None, mean_dev, std_dev)
        if all_frames:
            xs_init, ld_f = xs_full, ld_T
        else:
            xs_init, ld_f = movie.standardized(frames, mean_dev, std_dev)
        pj_dev = None
        movie.release()      # the standardised copies are written: the raw frames are not read again
        # Memory plan for movies that fill the HBM (BASELINE configs 4 / 5: 84 GB): with every frame fitted the filtered
        # copy X_f = X - B (B^T X) replaces the standardised one in place, and the movie projection of the last stage is
        # formed as (UW)^T X = (UW)^T X_f + ((UW)^T B) (B^T X) - the second term is a rank-K product of quantities the
        # global stage holds anyway (the background strip of U^T U, the background traces).  One movie-sized array
        # instead of two (three with the raw copy).  `single_copy=None`: automatic, above 1/8 of the device memory.
        Tf_, Ta_ = len(frames), (len(frames) // int(temporal_avg_factor)) * int(temporal_avg_factor)
        if single_copy is None:
            single_copy = 4.0 * Dl * T > torch.cuda.get_device_properties(ctx.device).total_memory / 8.0
        single_copy = bool(single_copy) and all_frames and K > 0 and pixel_weighting is None and Ta_ == Tf_
        if K > 0:
            pj_dev = torch.zeros((K, ld_f), dtype=torch.float32, device=ctx.device)
            # projection on the background basis: a sum over pixels - owned pixels here, summed over the ranks
            ws = ctx.workspace(lib.pmd_bg_project_workspace_bytes(O_hi - O_lo, Tf))
            ctx.call("pmd_bg_project", ptr(xs_init[O_lo - P_lo:]), O_hi - O_lo, Tf, ld_f, ptr(basis_dev[O_lo:]), K, ptr(pj_dev),
                     ld_f, ptr(ws), ws.numel())
            dist.all_reduce(pj_dev)
            if single_copy:
                xf = xs_init                     # filtered in place (the kernel is element-wise)
                xs_full = None
            else:
                xf = torch.empty_like(xs_init)   # bg_filter writes all ld columns of the resident pixel rows
                if xf.shape[0] > Dl:
                    xf[Dl:].zero_()
            ctx.call("pmd_bg_filter", ptr(xs_init), ptr(xf), Dl, Tf, ld_f, ptr(basis_dev[P_lo:]), K, ptr(pj_dev), ld_f)
            if single_copy:
                xs_init = None
        else:
            xf = xs_init.clone() if pixel_weighting is not None else xs_init
        if pixel_weighting is not None:
            pw = _f32(ctx, np.asarray(pixel_weighting, dtype=np.float32).reshape(-1)[P_lo:P_hi])
            ctx.call("pmd_scale_rows", ptr(xf), Dl, Tf, ld_f, ptr(pw))
        lap("standardize_filter", t0)

        # ---- tile grid (decomposition.py:721-773)
        display("Obtaining blocks and running local SVD")
        dim_1_iters, dim_2_iters = grid.tile_origins((d1, d2), block_sizes)
        if temporal_avg_factor >= Tf:
            raise ValueError("Need at least {} frames".format(temporal_avg_factor))
        if Tf // temporal_avg_factor <= max_components:
            display(
                f"WARNING: temporal avg factor is too big, max rank per block adjusted to {Tf // temporal_avg_factor}.\n"
                "To avoid this, initialize with more frames or reduce temporal avg factor")
            max_components = int(Tf // temporal_avg_factor)
        crop = (Tf // temporal_avg_factor) * temporal_avg_factor
        # temporal windows of windowed_pmd (decomposition.py:455-463)
        win_len = int(min(window_chunks, crop))
        win_starts = list(range(0, crop, win_len))
        if win_starts[-1] + win_len > crop:
            win_starts[-1] = crop - win_len
        n_win = len(win_starts)
        if n_win > 1 and win_len % temporal_avg_factor != 0:
            raise ValueError("window_chunks must be a multiple of temporal_avg_factor")
        if crop < Tf:
            # The tile fits use the first `crop` fitted frames only (decomposition.py:773-774, :796), and the tile
            # kernels walk the time axis in 32-frame chunks relying on zeros behind the last frame they are given:
            # frames crop .. Tf - 1 of the filtered copy must not leak into the contractions.  (Found by the seeded
            # fuzz of round 2: every draw with frames % temporal_avg_factor != 0 was off by ~1e-3 in its tile bases.)
            if xs_init is not None and xf.data_ptr() == xs_init.data_ptr():
                xf = xs_init.clone()    # xs_init may be the standardised movie the V projection still needs
            xf[:, crop:Tf].zero_()
        pix_c, origins = grid.tile_pixel_lists((d1, d2), block_sizes, dim_1_iters, dim_2_iters)
        n_tiles = pix_c.shape[0]
        pool_q, pool_idx, pool_w, pooled_shape = grid.pooling_maps(block_sizes, int(spatial_avg_factor))
        P_pool = pool_q.shape[0]
        dpad = lib.pmd_tile_dpad(d)
        if dpad < 0 or lib.pmd_tile_dpad(P_pool) < 0:
            raise ValueError("block of {} x {} pixels is larger than the supported maximum (65536 pixels)".format(b1, b2))
        # Component rows of the per-tile arrays: 64 while max_components + 10 <= 64 (the MFMA-tiled kernels), a larger multiple
        # of 64 beyond (generic-width kernels, csrc/wide.hip) - max_components is unbounded, as in the reference
        # (decomposition.py:643-665).  The global stage works on blocks of 64 component rows: an array with rpad = 64 nvt rows is
        # handed to it as nvt "virtual tiles" per tile, [n * nvt][64][x], the same memory (see below, after the assembly).
        rpad = int(lib.pmd_tile_rpad(int(max_components)))
        nvt = rpad // 64
        pix_dev = _i32(ctx, pix_c)
        pool_q_dev, pool_idx_dev, pool_w_dev = _i32(ctx, pool_q), _i32(ctx, pool_idx), _f32(ctx, pool_w)

        t0 = time.perf_counter()
        r = int(max_components)
        ldv = ld_f
        ut_dev = torch.empty((n_tiles, rpad, dpad), dtype=torch.float32, device=ctx.device)
        stats_dev = torch.zeros((n_tiles, rpad, 2), dtype=torch.float32, device=ctx.device)
        good_dev = torch.zeros((n_tiles, rpad), dtype=torch.int32, device=ctx.device)
        keep_dev = torch.zeros((n_tiles, rpad), dtype=torch.int32, device=ctx.device)
        ranks_dev = torch.zeros((n_tiles,), dtype=torch.int32, device=ctx.device)
        lam_dev = torch.zeros((n_tiles, rpad), dtype=torch.float64, device=ctx.device)
        assert n_tiles == n1_geo * n2_geo
        t_lo, t_hi = runs[dist.rank] if dist.enabled else (0, n_tiles)
        if not dist.enabled:
            runs = [(0, n_tiles)]
        n_loc = t_hi - t_lo
        # pixel lists of this rank's tiles relative to its slab (the kernels that read the movie copies)
        pix_loc_dev = _i32(ctx, pix_c[t_lo:t_hi] - P_lo) if dist.enabled else pix_dev
        if sim_pending is not None:
            sim_pending[0].sync()
            sim_stats = sim_pending[1].cpu().numpy()
            spatial_threshold = np.percentile(sim_stats[:, 0], sim_conf)
            temporal_threshold = np.percentile(sim_stats[:, 1], sim_conf)
            sim_pending = None
        thr_s32, thr_t32 = float(np.float32(spatial_threshold)), float(np.float32(temporal_threshold))
        a_f = int(temporal_avg_factor)
        # Tile batches.  The per-tile temporaries ([tile][64][frames] traces and the kernels' workspace, ~3 such arrays)
        # grow with tiles x frames: 84 GB each at BASELINE config 5 (65 025 tiles x 5 000 frames).  Above
        # `tile_batch_bytes` the tiles are fitted (and later projected) in batches whose traces are compacted to the
        # kept components right away; one batch (every workload up to config 3's size) is the single-launch path.
        per_tile_bytes = 3 * rpad * int(max(ldv, ld_T)) * 4 + 8 * rpad * dpad * 4 + (6 * rpad * rpad * 8 if nvt > 1 else 0)
        tile_batch = max(8, int(tile_batch_bytes // per_tile_bytes))
        batches = [(b0, min(n_loc, b0 + tile_batch)) for b0 in range(0, max(n_loc, 1), tile_batch)] if n_win == 1 else [(0, n_loc)]
        batched = len(batches) > 1
        v_dev = None if batched else torch.empty((n_tiles, rpad, ldv), dtype=torch.float32, device=ctx.device)
        v_pieces = []

        def tiles_args(g0, nb_, v_out, x_rows=None, n_rows=None, pix_rows=None):
            # g0: global index of the batch's first tile (the Gaussian matrix of a tile is keyed by it); a batch may hand over
            # only the band of pixel rows its tiles touch (x_rows, n_rows) with pixel lists relative to it (pix_rows)
            return (ptr(xf if x_rows is None else x_rows), ld_f, Dl if n_rows is None else n_rows, crop,
                    ptr(pix_loc_dev[g0 - t_lo:] if pix_rows is None else pix_rows), nb_, b1, b2, ptr(pool_q_dev), pool_q.shape[1], P_pool,
                    ptr(pool_idx_dev), ptr(pool_w_dev), r, a_f, thr_s32, thr_t32, int(max_consecutive_failures), seed, g0, 1,
                    ptr(ut_dev[g0:]), ptr(v_out), ldv, ptr(stats_dev[g0:]), ptr(good_dev[g0:]), ptr(keep_dev[g0:]),
                    ptr(ranks_dev[g0:]), ptr(lam_dev[g0:]))

        if n_loc > 0 and n_win == 1 and not batched:
            ws = ctx.workspace(lib.pmd_tiles_workspace_bytes(n_loc, b1, b2, P_pool, r, a_f, crop, ldv, Dl))
            _tiles_decompose(ctx, tiles_args(t_lo, n_loc, v_dev[t_lo:]), ws, (n_loc, b1, b2, P_pool, r, a_f, crop, ldv, Dl, dpad),
                             temporal_denoiser, spatial_denoiser)
        elif n_loc > 0 and n_win == 1:
            for b0, b1_ in batches:
                nb_, g0 = b1_ - b0, t_lo + b0
                vb = torch.empty((nb_, rpad, ldv), dtype=torch.float32, device=ctx.device)
                # the band of resident pixels the batch touches (tiles are in tile-row order): the temporal binning of the
                # sketch walks only these rows instead of the whole movie once per batch (409 ms of 1.46 s at config 5)
                lo_pix = int(pix_c[g0:g0 + nb_].min()) - P_lo
                n_band = int(pix_c[g0:g0 + nb_].max()) + 1 - P_lo - lo_pix
                pix_b = (pix_loc_dev[b0:b1_] - lo_pix).contiguous()
                ws = ctx.workspace(lib.pmd_tiles_workspace_bytes(nb_, b1, b2, P_pool, r, a_f, crop, ldv, n_band))
                _tiles_decompose(ctx, tiles_args(g0, nb_, vb, xf[lo_pix:], n_band, pix_b), ws,
                                 (nb_, b1, b2, P_pool, r, a_f, crop, ldv, n_band, dpad), temporal_denoiser, spatial_denoiser)
                rk = _virtual_ranks(ranks_dev[g0:g0 + nb_], nvt)     # blocks of 64 component rows
                off_b = (torch.cumsum(rk, 0) - rk).to(torch.int32)
                rows_b = int(rk.sum().item())
                piece = torch.zeros((max(rows_b, 1), crop), dtype=torch.float32, device=ctx.device)
                ctx.call("pmd_compact_rows", ptr(vb), ldv, ptr(off_b), ptr(rk), crop, ptr(piece), crop, nb_ * nvt)
                v_pieces.append(piece[:rows_b])
                del vb
            ctx.release_workspace()     # the batch workspace (GBs) is not needed again
        elif n_loc > 0:
            # several windows: first window = single_block_md, later ones fit the residual (decomposition.py:471-515);
            # the Gaussian matrix of (tile, window) is logical array tile * n_win + window
            ld_w = lib.pmd_time_ld(win_len)
            xw = torch.zeros((movie.rows_alloc, ld_w), dtype=torch.float32, device=ctx.device)
            vw = torch.empty((n_loc, rpad, ld_w), dtype=torch.float32, device=ctx.device)
            st_w = torch.zeros((n_loc, rpad, 2), dtype=torch.float32, device=ctx.device)
            gd_w = torch.zeros((n_loc, rpad), dtype=torch.int32, device=ctx.device)
            kp_w = torch.zeros((n_loc, rpad), dtype=torch.int32, device=ctx.device)
            for widx, w0 in enumerate(win_starts):
                xw[:Dl, :win_len] = xf[:Dl, w0:w0 + win_len]
                if widx == 0:
                    ws = ctx.workspace(lib.pmd_tiles_workspace_bytes(n_loc, b1, b2, P_pool, r, a_f, win_len, ld_w, Dl))
                    # the denoiser hooks act in single_block_md only, i.e. in the first window (decomposition.py:476-488)
                    _tiles_decompose(ctx, (ptr(xw), ld_w, Dl, win_len, ptr(pix_loc_dev), n_loc, b1, b2,
                                           ptr(pool_q_dev), pool_q.shape[1], P_pool, ptr(pool_idx_dev), ptr(pool_w_dev), r, a_f,
                                           thr_s32, thr_t32, int(max_consecutive_failures), seed, t_lo * n_win, n_win,
                                           ptr(ut_dev[t_lo:]), ptr(vw), ld_w, ptr(stats_dev[t_lo:]), ptr(good_dev[t_lo:]),
                                           ptr(keep_dev[t_lo:]), ptr(ranks_dev[t_lo:]), ptr(lam_dev[t_lo:])), ws,
                                     (n_loc, b1, b2, P_pool, r, a_f, win_len, ld_w, Dl, dpad), temporal_denoiser,
                                     spatial_denoiser)
                    ctx.call("pmd_tiles_truncate", ptr(ut_dev[t_lo:]), dpad, ptr(ranks_dev[t_lo:]), n_loc, rpad)
                else:
                    ws = ctx.workspace(lib.pmd_tiles_residual_workspace_bytes(n_loc, b1, b2, r, a_f, win_len, Dl))
                    ctx.call("pmd_tiles_residual", ptr(xw), ld_w, Dl, win_len, ptr(pix_loc_dev), n_loc, b1, b2, r, a_f,
                             thr_s32, thr_t32, int(max_consecutive_failures), seed, t_lo * n_win + widx, n_win,
                             ptr(ut_dev[t_lo:]), ptr(ranks_dev[t_lo:]), ptr(st_w), ptr(gd_w), ptr(kp_w), ptr(ws), ws.numel())
            del xw, vw
            # temporal traces over all fitted frames: U_b^T X (get_temporal_projector, decomposition.py:518-523)
            # (blocks of 64 component rows: a tile with rpad = 64 nvt rows is nvt such blocks with the same pixels)
            pix_blocks = pix_loc_dev if nvt == 1 else pix_loc_dev.repeat_interleave(nvt, dim=0)
            ctx.call("pmd_tiles_project", ptr(xf), ld_f, crop, ptr(pix_blocks), n_loc * nvt, d, ptr(ut_dev[t_lo:]), dpad,
                     ptr(v_dev[t_lo:]), ldv, 2)
        for tns in (ut_dev, stats_dev, good_dev, keep_dev, ranks_dev, lam_dev):
            dist.gather_runs(tns, runs)
        # geometry-only host tables of the assembly and of U^T U, built while the tile kernels run
        fov_ids = np.arange(D).reshape((d1, d2), order=order)
        cumw = grid.cumulative_weights((d1, d2), block_sizes, origins, block_weights)
        cover1, cover2 = grid.cover_tables((d1, d2), block_sizes, dim_1_iters, dim_2_iters)
        pairs = grid.overlap_pairs(origins, block_sizes)
        inv_cumw_host = np.ascontiguousarray(1.0 / cumw.reshape(-1))
        ctx.sync()
        lap("tiles", t0)
        _dbg("ut", ut_dev); _dbg("tile_lambda", lam_dev)

        # ---- sparse assembly (decomposition.py:752-857): CSR arrays built on the device
        t0 = time.perf_counter()
        ranks = ranks_dev.cpu().numpy().astype(np.int64)
        offsets = np.concatenate([[0], np.cumsum(ranks)]).astype(np.int64)
        Rt = int(offsets[-1])
        K_cols = K if K > 0 else 1  # K <= 0: one empty placeholder column (pmd_loader.py:301-302)
        R = Rt + K_cols
        Rc = Rt + max(K, 0)         # columns with content
        col_off_dev = _i32(ctx, offsets[:-1])
        w_dev = _f32(ctx, block_weights.reshape(-1, order="F"))
        inv_cumw_dev = torch.from_numpy(inv_cumw_host).to(ctx.device)
        cov1_dev, cov2_dev = _i32(ctx, cover1), _i32(ctx, cover2)
        o1_dev, o2_dev = _i32(ctx, dim_1_iters), _i32(ctx, dim_2_iters)
        row_nnz = torch.empty(D, dtype=torch.int64, device=ctx.device)
        n2 = len(dim_2_iters)
        order_f = 1 if order == "F" else 0
        ctx.call("pmd_csr_count", d1, d2, order_f, ptr(cov1_dev), ptr(cov2_dev), n2, ptr(ranks_dev), max(K, 0), ptr(row_nnz))
        indptr_dev = torch.zeros(D + 1, dtype=torch.int64, device=ctx.device)
        indptr_dev[1:] = torch.cumsum(row_nnz, 0)
        nnz = int(indptr_dev[-1].item())
        data_dev = torch.empty(max(nnz, 1), dtype=torch.float64, device=ctx.device)
        idx_dev = torch.empty(max(nnz, 1), dtype=torch.int32, device=ctx.device)
        zero_dev = torch.zeros(1, dtype=torch.int32, device=ctx.device)
        ctx.call("pmd_csr_fill", d1, d2, order_f, b1, ptr(cov1_dev), ptr(cov2_dev), ptr(o1_dev), ptr(o2_dev), n2,
                 ptr(ranks_dev), ptr(col_off_dev), ptr(ut_dev), dpad, ptr(w_dev), ptr(inv_cumw_dev), ptr(basis_dev),
                 max(K, 0), Rt, ptr(indptr_dev), ptr(data_dev), ptr(idx_dev), ptr(zero_dev), rpad)
        ut_host = None
        basis_rows = None
        u_pending = None
        if int(zero_dev.item()) == 0 and nnz < 2 ** 31:
            # the CSR arrays go to the host on the side stream while the global stage runs
            side = _side_stream(ctx.device)
            u_host = (torch.empty(nnz, dtype=torch.float64, pin_memory=True),
                      torch.empty(nnz, dtype=torch.int32, pin_memory=True),
                      torch.empty(D + 1, dtype=torch.int64, pin_memory=True))
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(ctx.device))
            side.wait_event(ev)
            with torch.cuda.stream(side):
                u_host[0].copy_(data_dev[:nnz], non_blocking=True)
                u_host[1].copy_(idx_dev[:nnz], non_blocking=True)
                u_host[2].copy_(indptr_dev, non_blocking=True)
            u_pending = (u_host, data_dev, idx_dev, indptr_dev)  # keeps the device arrays alive until the copy is done
            u_r = None
        else:
            # exact zeros present (the reference drops them, decomposition.py:853): host construction
            ut_host = ut_dev.cpu().numpy()
            pix_f = fov_ids.reshape(-1)[pix_c]
            inv_rows = np.zeros(D)
            inv_rows[fov_ids.reshape(-1)] = 1.0 / cumw.reshape(-1)
            u_local, _ = _sparse_u(ut_host, ranks, pix_f, block_weights, inv_rows, D)
            if K > 0:
                basis_rows = np.empty((D, K), dtype=np.float32)
                basis_rows[fov_ids.reshape(-1)] = basis_dev.cpu().numpy()
                u_r = scipy.sparse.hstack([u_local, coo_matrix(basis_rows)])
            else:
                u_r = scipy.sparse.hstack([u_local, coo_matrix((D, 1), dtype=np.float32)])
        display("Normalizing by weights")
        display("The total rank before pruning is {}".format(R))
        lap("assembly", t0)
        tile_ranks_out = ranks.astype(np.int32)
        if nvt > 1:
            # From here on a "tile" is a block of 64 component rows: tile t with rpad = 64 nvt rows becomes the nvt virtual tiles
            # t nvt + h (same pixels, components 64 h .. 64 h + 63), in the same memory.  Column order, and with it every row
            # index of the global stage, is unchanged; two virtual tiles of one tile overlap on all of its pixels.
            ranks = np.clip(ranks[:, None] - 64 * np.arange(nvt)[None, :], 0, 64).reshape(-1)
            offsets = np.concatenate([[0], np.cumsum(ranks)]).astype(np.int64)
            ranks_dev = _virtual_ranks(ranks_dev, nvt)
            col_off_dev = _i32(ctx, offsets[:-1])
            ut_dev = ut_dev.view(n_tiles * nvt, 64, dpad)
            if v_dev is not None:
                v_dev = v_dev.view(n_tiles * nvt, 64, ldv)
            pix_c = np.repeat(pix_c, nvt, axis=0)
            origins = np.repeat(np.asarray(origins), nvt, axis=0)
            pix_dev = pix_dev.repeat_interleave(nvt, dim=0)
            pix_loc_dev = pix_loc_dev.repeat_interleave(nvt, dim=0) if dist.enabled else pix_dev
            pairs = grid.virtual_pairs(pairs, nvt)
            runs = [(lo * nvt, hi * nvt) for lo, hi in runs]
            batches = [(lo * nvt, hi * nvt) for lo, hi in batches]
            n_tiles, t_lo, t_hi, n_loc = n_tiles * nvt, t_lo * nvt, t_hi * nvt, n_loc * nvt

        # ---- orthogonalisation (decomposition.py:860-881)
        display("Performing rank pruning and orthogonalization for fast sparse regression.")
        t0 = time.perf_counter()
        cumw_dev = _f32(ctx, cumw.reshape(-1))
        uw_dev = torch.empty_like(ut_dev)
        ctx.call("pmd_weight_tiles", ptr(ut_dev), dpad, ptr(pix_dev), d, ptr(w_dev), ptr(cumw_dev), ptr(ranks_dev),
                 ptr(uw_dev), n_tiles)
        pairs_dev, origins_dev = _i32(ctx, pairs), _i32(ctx, origins)

        # v_cropped = [tile traces ; background temporal basis] (decomposition.py:844, :932)
        m_cols = crop
        vc = torch.zeros((Rc, m_cols), dtype=torch.float32, device=ctx.device)
        if n_loc > 0 and batched:
            row = int(offsets[t_lo])
            for piece in v_pieces:
                vc[row:row + piece.shape[0]] = piece
                row += piece.shape[0]
            v_pieces = None
        elif n_loc > 0:
            ctx.call("pmd_compact_rows", ptr(v_dev[t_lo:]), ldv, ptr(col_off_dev[t_lo:]), ptr(ranks_dev[t_lo:]), crop, ptr(vc),
                     m_cols, n_loc)
        row_bounds = [(int(offsets[lo]), int(offsets[hi])) for lo, hi in runs]
        if K > 0:
            vc[Rt:Rt + K, :] = pj_dev[:, :crop]
        right = vc
        col_sigma = vc.norm(dim=1).cpu().numpy() if return_diagnostics else None   # norm of every trace (tile sigma)
        if rank_prune:
            if rank_prune_factor <= 0 or rank_prune_factor > 1:
                raise ValueError("Rank prune factor should be a value in the interval (0, 1]")
            min_dimension = min(R, crop)
            n_rand = int(min_dimension * rank_prune_factor)
            rand = torch.empty((crop, max(n_rand, 1)), dtype=torch.float32, device=ctx.device)
            ctx.call("pmd_rng_normal", seed, STREAM_PRUNE, 0, 0, 1, crop, n_rand, 0, ptr(rand), n_rand, 0)
            right = torch.empty((Rc, n_rand), dtype=torch.float32, device=ctx.device)
            ctx.call("pmd_gemm", 0, 0, Rc, n_rand, crop, 1.0, ptr(vc), m_cols, ptr(rand), n_rand, 0.0, ptr(right), n_rand)
            m_cols = n_rand
        use_right = R > m_cols  # decomposition.py:976 (R counts the placeholder column too)
        # Rows of `right` that live on other ranks.  Row-sharded Cholesky route: a rank applies the block-sparse Gram
        # matrix to its own tiles only and needs, besides its own rows, the rows of the foreign tiles that overlap
        # them (a halo of one or two tile rows on either side: tens of MB instead of the whole 2.2 GB matrix at
        # config 3).  Every other route is replicated and collects all rows.
        will_shard = dist.enabled and use_right and orthogonalizer in ("auto", "cholesky")
        if will_shard:
            tile_owner = np.searchsorted(np.asarray([hi for _, hi in runs]), np.arange(n_tiles), side="right")
            needs = [[] for _ in range(dist.world)]
            pa, pb = pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)
            for a_t, b_t in ((pa, pb), (pb, pa)):
                cross = tile_owner[a_t] != tile_owner[b_t]
                for rr in range(dist.world):
                    for t in np.unique(b_t[cross & (tile_owner[a_t] == rr)]):
                        needs[rr].append((int(offsets[t]), int(offsets[t + 1])))
            dist.exchange_rows(right, row_bounds, [sorted(set(x for x in nd if x[1] > x[0])) for nd in needs])
        elif dist.enabled:
            dist.gather_runs(right, row_bounds)
        _dbg("v_cropped", vc)
        P_dev = Et_dev = None
        chol_ok = False
        null_tail = False
        null_info = {}              # kept null direction of the Cholesky route: pivot level, coupling left out (diagnostics)
        shard = False               # rows of right / GM / Z / R split over the ranks (Cholesky route only)
        row_lo, row_hi = 0, Rc
        Z = W1 = None
        bg_strip = None             # (K x Rc) background rows of U^T U (single-copy plan: the rank-K term of the projection)

        def build_z(everywhere):
            """Z = (U W)^T ((Y - mean) / std) over the whole movie (pmd_loader.py:316-346).  A rank projects the tiles
            of its own run (it holds no other pixels); everywhere=True then collects all rows on every rank (the
            replicated global stage), otherwise a rank keeps only its own rows (row-sharded stage).  The K
            background rows are filled on every rank."""
            src = xf if xs_full is None else xs_full   # single-copy plan: the filtered movie + the rank-K term below
            z = torch.zeros((Rc, T), dtype=torch.float32, device=ctx.device)
            if n_loc > 0 and batched:
                for b0, b1_ in batches:
                    nb_, g0 = b1_ - b0, t_lo + b0
                    proj = torch.empty((nb_, 64, ld_T), dtype=torch.float32, device=ctx.device)
                    ctx.call("pmd_tiles_project_ranked", ptr(src), ld_T, T, ptr(pix_loc_dev[b0:]), nb_, d, ptr(uw_dev[g0:]), dpad,
                             ptr(proj), ld_T, 2, ptr(ranks_dev[g0:]))
                    ctx.call("pmd_compact_rows", ptr(proj), ld_T, ptr(col_off_dev[g0:]), ptr(ranks_dev[g0:]), T, ptr(z), T, nb_)
                    del proj
            elif n_loc > 0:
                if all_frames and ldv == ld_T:
                    proj = v_dev     # same shape; the fit-frame traces are already compacted into v_cropped
                else:
                    proj = torch.empty((n_tiles, 64, ld_T), dtype=torch.float32, device=ctx.device)
                ctx.call("pmd_tiles_project_ranked", ptr(src), ld_T, T, ptr(pix_loc_dev), n_loc, d, ptr(uw_dev[t_lo:]), dpad,
                         ptr(proj[t_lo:]), ld_T, 2, ptr(ranks_dev[t_lo:]))
                ctx.call("pmd_compact_rows", ptr(proj[t_lo:]), ld_T, ptr(col_off_dev[t_lo:]), ptr(ranks_dev[t_lo:]), T, ptr(z), T,
                         n_loc)
            if xs_full is None and K > 0:
                # (UW)^T X = (UW)^T X_f + ((UW)^T B) (B^T X): bg_strip[k][c] = (B^T U W)[k][c], pj_dev = B^T X
                lo_, hi_ = int(offsets[t_lo]), int(offsets[t_hi])
                if hi_ > lo_:
                    ctx.call("pmd_gemm", 1, 0, hi_ - lo_, T, K, 1.0, ptr(bg_strip[:, lo_:]), bg_strip.shape[1], ptr(pj_dev), ld_f, 1.0,
                             ptr(z[lo_:]), T)
            if everywhere:
                dist.gather_runs(z, [(int(offsets[lo]), int(offsets[hi])) for lo, hi in runs])
            if K > 0:
                if all_frames:
                    z[Rt:Rt + K, :] = pj_dev[:, :T]
                else:
                    pj_full = torch.zeros((K, ld_T), dtype=torch.float32, device=ctx.device)
                    ws_ = ctx.workspace(lib.pmd_bg_project_workspace_bytes(O_hi - O_lo, T))
                    ctx.call("pmd_bg_project", ptr(xs_full[O_lo - P_lo:]), O_hi - O_lo, T, ld_T, ptr(basis_dev[O_lo:]), K,
                             ptr(pj_full), ld_T, ptr(ws_), ws_.numel())
                    dist.all_reduce(pj_full)
                    z[Rt:Rt + K, :] = pj_full[:, :T]
            return z

        if use_right:
            # P = right E / sqrt(lambda) stays factored; G = U^T U stays block-sparse
            n_pairs = pairs.shape[0]
            gblk = torch.empty((max(n_pairs, 1), 64, 64), dtype=torch.float32, device=ctx.device)
            gbg = torch.zeros((max(1, (max(K, 0) + 63) // 64) * n_tiles, 64, 64), dtype=torch.float32, device=ctx.device)
            gstrip = torch.zeros((max(K, 1), Rc), dtype=torch.float32, device=ctx.device)
            ctx.call("pmd_gram_blocks", ptr(uw_dev), dpad, b1, b2, ptr(pix_dev), ptr(pairs_dev), n_pairs, ptr(origins_dev),
                     ptr(col_off_dev), ptr(ranks_dev), n_tiles, Rt, ptr(basis_dev), D, max(K, 0), ptr(gblk), ptr(gbg),
                     ptr(gstrip), Rc)
            bg_strip = gstrip
            nbr_ptr, nbr = grid.neighbour_lists(pairs, ranks, offsets[:-1], n_tiles, Rt, max(K, 0))
            nbr_ptr_dev, nbr_dev = _i32(ctx, nbr_ptr), _i32(ctx, nbr)
            ld_right = m_cols
            GM = torch.empty((Rc, m_cols), dtype=torch.float32, device=ctx.device)
            Et_dev = torch.empty((m_cols, m_cols), dtype=torch.float32, device=ctx.device)

            # Rows of `right`, GM, Z and R owned by this rank: the components of its tile run; the last rank
            # also owns the K background rows.  Single process: everything.
            row_lo, row_hi = int(offsets[t_lo]), int(offsets[t_hi])
            if dist.rank == dist.world - 1:
                row_hi = Rc
            shard = dist.enabled

            def gram_apply(ncols):
                # block rows of G M for the tiles of this rank (all tiles when not distributed); the K background
                # rows come from one small GEMM inside the call (every rank forms them, only their owner uses them)
                if n_loc > 0 or not shard:
                    a0, an = (t_lo, n_loc) if shard else (0, n_tiles)
                    ctx.call("pmd_gram_apply", ptr(gblk), ptr(gbg), ptr(gstrip), Rc, ptr(nbr_ptr_dev[a0:]), ptr(nbr_dev),
                             ptr(col_off_dev[a0:]), ptr(ranks_dev[a0:]), an, Rt, max(K, 0),
                             int(ranks.max()) if n_tiles else 0, ptr(right), ld_right, ncols, ptr(GM), m_cols)
                if shard and K > 0:
                    # The background rows of G M are sums over ALL rows of `right`; a rank holds its own rows and a halo
                    # only.  Every rank contributes the product over the rows it owns (the last one also the background
                    # columns), the K x ncols partials are all-reduced, their owner (the last rank) keeps the result.
                    part = torch.zeros((K, ncols), dtype=torch.float32, device=ctx.device)
                    lo_, hi_ = row_bounds[dist.rank][0], (Rc if dist.rank == dist.world - 1 else row_bounds[dist.rank][1])
                    if hi_ > lo_:
                        ctx.call("pmd_gemm", 0, 0, K, ncols, hi_ - lo_, 1.0, ptr(gstrip[:, lo_:]), Rc, ptr(right[lo_:]), ld_right, 0.0,
                                 ptr(part), ncols)
                    dist.all_reduce(part)
                    GM[Rt:Rt + K, :ncols] = part

            chol_ok = False
            hv_dev = None
            if orthogonalizer in ("auto", "cholesky"):
                m_eff = m_cols
                abs_last = 0
                if all_frames and crop == T and not rank_prune and pixel_weighting is None:
                    # Every standardised trace sums to ~0 over the frames it was centred on (all of them: crop == T),
                    # so the constant vector is a numerically null right vector of v_cropped.  Rotate it into the
                    # last column (Householder H, H e_m = 1/sqrt(m)): the leading block of C is then well
                    # conditioned and the null direction sits in the last pivot.  "keep" (the reference's rule:
                    # |lambda| of a numerically null direction is kept, decomposition.py:984-988): the last pivot
                    # enters through its absolute value; "drop": the column is dropped (one component fewer).
                    nhat = np.full(m_cols, 1.0 / math.sqrt(m_cols))
                    hv = -nhat
                    hv[-1] += 1.0
                    hv /= np.linalg.norm(hv)
                    hv_dev = _f32(ctx, hv)
                    y_dev = torch.empty((Rc, 1), dtype=torch.float32, device=ctx.device)
                    ctx.call("pmd_gemm", 0, 0, Rc, 1, m_cols, 1.0, ptr(right), ld_right, ptr(hv_dev), 1, 0.0, ptr(y_dev), 1)
                    ctx.call("pmd_gemm", 0, 0, Rc, m_cols, 1, -2.0, ptr(y_dev), 1, ptr(hv_dev), m_cols, 1.0, ptr(right), ld_right)
                    if null_directions == "keep":
                        abs_last = 1
                    else:
                        m_eff = m_cols - 1
                gram_apply(m_eff)
                ok_c = c_i(0)
                # C = right[rows]^T GM[rows] (summed over the ranks when the rows are sharded)
                if shard:
                    Et_dev.zero_()
                nrow = row_hi - row_lo
                Mt_buf = None
                if nrow > 0:
                    # pmd_gram_mtgm writes the transposed copy of its M rows at the start of its workspace; handing it a
                    # buffer of our own keeps that copy for the M^T Z product below (one transpose and, at BASELINE
                    # config 5, 20 GB fewer than a second copy)
                    Mt_buf = torch.empty((lib.pmd_gram_mtgm_workspace_bytes(nrow, m_eff) + 3) // 4, dtype=torch.float32, device=ctx.device)
                    ctx.call("pmd_gram_mtgm", ptr(right[row_lo:]), nrow, m_eff, ld_right, ptr(GM[row_lo:]), m_cols,
                             ptr(Et_dev), m_cols, ptr(Mt_buf), Mt_buf.numel() * 4)
                dist.all_reduce(Et_dev)
                tr_c_dev = Et_dev.diagonal()[:m_eff].sum() if abs_last else None   # scale of C for the null-pivot test below
                # The Cholesky step (C -> Et) is a chain of small latency-bound launches; the V projection
                # Z = (UW)^T Y and the large product M^T Z do not depend on it.  Those are enqueued on the main
                # stream first, then the Cholesky step runs on a second stream next to them.
                main = torch.cuda.current_stream(ctx.device)
                ev_c = torch.cuda.Event()
                ev_c.record(main)
                lap("orthogonalize", t0)
                t0 = time.perf_counter()
                Z = build_z(everywhere=not shard)
                W1 = torch.zeros((m_eff, T), dtype=torch.float32, device=ctx.device)
                if nrow > 0:
                    ld_mt = int(lib.pmd_gram_mtgm_ld(nrow))   # (rows of the transposed copy are padded to 64 floats)
                    ctx.call("pmd_gemm", 0, 0, m_eff, T, nrow, 1.0, ptr(Mt_buf), ld_mt, ptr(Z[row_lo:]), T, 0.0, ptr(W1), T)
                lap("v_projection", t0)
                t0 = time.perf_counter()
                # (driving the side stream from a helper thread, so that the chain is enqueued before the blocking read-back
                # inside the M^T Z product, was tried: no gain at config 3 - the chain is bound by CU availability next to
                # the product, not by when it is enqueued - and erratic 40-70 ms waits at config 2; not kept)
                sc = ctx.side()
                sc.stream.wait_event(ev_c)
                with torch.cuda.stream(sc.stream):
                    ws2 = sc.workspace(lib.pmd_chol_inverse_workspace_bytes(m_eff))
                    sc.call("pmd_chol_inverse", ptr(Et_dev), m_eff, m_cols, abs_last, C.byref(ok_c), ptr(ws2), ws2.numel())
                ev_e = torch.cuda.Event()
                ev_e.record(sc.stream)
                main.wait_event(ev_e)
                ctx.sync()
                Mt = Mt_buf = None
                chol_ok = bool(ok_c.value)
                if chol_ok:
                    dist.all_reduce(W1)
                else:
                    Z = W1 = None   # eigenvector route below: full Z, M^T Z inside pmd_projected_svd_factored
                null_tail = False
                if chol_ok:
                    rp = m_eff
                    m_used = m_eff
                    if abs_last:
                        # Is the last pivot really at rounding level?  (with a large mean / noise ratio the
                        # fp32 centring leaves the constant vector a small but resolved direction; splitting it off would
                        # then drop real signal.)  Et[m-1][m-1] = 1 / sqrt(|pivot|); compared with the mean diagonal of C.
                        piv = 1.0 / max(float(Et_dev[m_eff - 1, m_eff - 1].item()) ** 2, 1e-300)
                        null_pivot_rel = piv / max(float(tr_c_dev.item()) / m_eff, 1e-300)
                        null_info["pivot_rel"] = null_pivot_rel
                        # The pivot of a null direction is rounding noise and can come out accidentally tiny (3e-12 of the mean
                        # diagonal seen: the kept direction was then scaled up 5e5 times and led the spectrum with 18 x s_1).
                        # The reference's fp32 eigensolver returns |lambda| ~ eps32 lambda_max for such a direction, never less:
                        # the pivot is floored at eps32 trace(C) / 4 (lambda_max <= trace), which bounds the amplification the
                        # same way (the last row of Et is linear in 1 / sqrt(pivot)).
                        floor_rel = 1.1920929e-07 * m_eff / 4.0
                        null_info["pivot_floor_rel"] = floor_rel
                        if null_pivot_rel < floor_rel:
                            Et_dev[m_eff - 1, :m_eff] *= math.sqrt(null_pivot_rel / floor_rel)
                    if abs_last and null_pivot_rel <= max(NULL_PIVOT_REL, 8.0 * floor_rel):
                        # The last row of Et is the kept numerically null direction (scale 1 / sqrt(|last pivot|), as the
                        # reference's 1 / sqrt(|lambda|), decomposition.py:984-996).  Its coupling to the other directions is
                        # rounding noise, so it is carried as one extra component next to the SVD of the leading
                        # m - 1 directions (whose Et block is exactly the Cholesky inverse of the deflated matrix)
                        # instead of through it, where its huge coefficients would amplify rounding errors into every
                        # other component.
                        null_tail = True
                        rp = m_used = m_eff - 1
                elif orthogonalizer == "cholesky":
                    raise PMDLibraryError("orthogonalizer='cholesky': U^T U restricted to the right matrix is not positive definite")
            if not chol_ok:
                if hv_dev is not None:
                    # Undo the rotation (H is an involution).  With the null direction isolated in the last column, the
                    # eigensolver returns an eigenvalue for it that is orders of magnitude below the rounding level of
                    # the unrotated matrix, and the reference's 1 / sqrt(|lambda|) rule then scales that direction up
                    # until it swamps every other component (seen in the seeded fuzz: mean squared residual 1400
                    # instead of 1.3).  Unrotated, the route is exactly the eigenvector route the caller can also ask for.
                    ctx.call("pmd_gemm", 0, 0, Rc, 1, m_cols, 1.0, ptr(right), ld_right, ptr(hv_dev), 1, 0.0, ptr(y_dev), 1)
                    ctx.call("pmd_gemm", 0, 0, Rc, m_cols, 1, -2.0, ptr(y_dev), 1, ptr(hv_dev), m_cols, 1.0, ptr(right), ld_right)
                if shard:
                    dist.gather_runs(right, row_bounds)   # only the halo rows were exchanged so far
                shard = False  # eigenvector route: replicated on every rank
                row_lo, row_hi = 0, Rc
                gram_apply(m_cols)
                ws = ctx.workspace(lib.pmd_orthogonalize_factored_workspace_bytes(m_cols))
                rp_c = c_i(0)
                ctx.call("pmd_orthogonalize_factored", ptr(right), Rc, m_cols, ld_right, ptr(GM), m_cols, ptr(Et_dev), m_cols,
                         C.byref(rp_c), ptr(ws), ws.numel())
                rp = int(rp_c.value)
                m_used = m_cols
            del GM, gblk, gbg
        else:
            G = torch.empty((Rc, Rc), dtype=torch.float32, device=ctx.device)
            ctx.call("pmd_gram_u", ptr(uw_dev), dpad, b1, b2, ptr(pix_dev), ptr(pairs_dev), pairs.shape[0], ptr(origins_dev),
                     ptr(col_off_dev), ptr(ranks_dev), n_tiles, Rt, ptr(basis_dev), D, max(K, 0), ptr(G), Rc)
            _dbg("G", G)
            if K > 0 and xs_full is None:
                bg_strip = G[Rt:Rt + K, :].clone()   # G is overwritten by the eigendecomposition
            P_dev, rp = _orthogonalize(ctx, G, Rc, None, m_cols, m_cols)
            del G
        display("After performing rank reduction, the updated rank is {}".format(rp + (1 if null_tail else 0)))
        lap("orthogonalize", t0)

        # ---- V = P^T U^T X over the whole movie (pmd_loader.py:316-346); already enqueued on the Cholesky route
        display("Running sparse regression")
        t0 = time.perf_counter()
        shard = shard and use_right and chol_ok and rp <= T
        if Z is None:
            Z = build_z(everywhere=True)
        _dbg("Z", Z)
        lap("v_projection", t0)

        # ---- V = P^T Z and the final SVD (decomposition.py:885, :894-904)
        display("Final reformat of data into complete SVD")
        t0 = time.perf_counter()
        if rp <= T:
            display("Short matrix, using leftward SVD routine")
        else:
            display("Tall matrix, using rightward SVD routine")
        Vp = torch.empty((rp, T), dtype=torch.float32, device=ctx.device) if return_diagnostics else None
        extra_row = 1 if K_cols != max(K, 0) else 0  # the reference's placeholder background column (a zero row of R)
        hosts = None
        if use_right and rp <= T:
            nk = rp
            nko = nk + (1 if null_tail else 0)   # components in the outputs
            # device copy of R only where rows are exchanged between ranks; rank 0's own rows go straight to the host
            # Zero copy only for the large case it was measured on: for small outputs rocBLAS may pick split-K kernels
            # that read-modify-write C, which is ruinous across PCIe (58 ms instead of 1 ms at 5015 x 1999).
            # (not when the product runs as fp16-piece products, gemm_f16x2.hip: its three passes accumulate into C, which
            # must then live in HBM; R is formed in row blocks there, each downloaded while the next is computed)
            split_gemm = bool(lib.pmd_gemm_split_active(ctx.handle, min(Rc, 16384), nk, m_used))
            zero_copy = (not shard or dist.rank == 0) and m_used >= 8192 and Rc * nk * 4 >= 2 ** 30 and not split_gemm
            R_out = torch.empty((Rc, nko), dtype=torch.float32, device=ctx.device) if (shard or not zero_copy) else None
            s_out = torch.empty((nko,), dtype=torch.float32, device=ctx.device)
            Vt_out = torch.empty((nko, T), dtype=torch.float32, device=ctx.device)
            X1 = torch.empty((m_used, rp), dtype=torch.float32, device=ctx.device)
            # W1 = M^T Z was formed next to the Cholesky step (summed over the ranks when the rows are sharded);
            # None on the eigenvector route (formed inside the call)
            if shard:
                # The frames x frames products of the last stage by FRAME COLUMNS (round 3): rank r forms its T / N columns
                # of Vp = Et W1 and their Gram matrix; the partial Gram matrices are all-reduced (one more m x m exchange);
                # the eigensolver runs replicated; every rank then forms its columns of Vt, which travel to rank 0.
                cparts = tile_partition(T, dist.world)
                c0, c1 = cparts[dist.rank]
                nc = c1 - c0
                ldc = _round_up(rp, 4)
                Vp_r = torch.empty((rp, max(nc, 1)), dtype=torch.float32, device=ctx.device)
                Cg = torch.zeros((rp, ldc), dtype=torch.float32, device=ctx.device)
                ctx.call("pmd_psvd_vp_gram", ptr(Et_dev), rp, m_used, m_cols, ptr(W1.view(-1)[c0:]), nc, T, 1 if chol_ok else 0,
                         ptr(Vp_r), max(nc, 1), ptr(Cg), ldc)
                dist.all_reduce(Cg)
                Wmat = torch.empty((rp, rp), dtype=torch.float32, device=ctx.device)
                Vt_r = torch.empty((rp, max(nc, 1)), dtype=torch.float32, device=ctx.device)
                ws = ctx.workspace(lib.pmd_psvd_finish_workspace_bytes(rp))
                ctx.call("pmd_psvd_finish", ptr(Cg), ldc, rp, ptr(Vp_r), nc, max(nc, 1), ptr(Wmat), rp, ptr(s_out), ptr(Vt_r),
                         max(nc, 1), ptr(ws), ws.numel())
                ctx.call("pmd_gemm", 1, 0, m_used, rp, rp, 1.0, ptr(Et_dev), m_cols, ptr(Wmat), rp, 0.0, ptr(X1), rp)
                ctx.sync()
                blocks = dist.gather_blocks_to_root(Vt_r[:, :nc].contiguous(), [(rp, b_ - a_) for a_, b_ in cparts])
                if blocks is not None:
                    for (a_, b_), blk in zip(cparts, blocks):
                        if b_ > a_:
                            Vt_out[:rp, a_:b_] = blk
                if Vp is not None:      # diagnostics only: every rank gets the whole V = P^T Z (a sum of disjoint column blocks)
                    Vp.zero_()
                    if nc > 0:
                        Vp[:, c0:c1] = Vp_r[:, :nc]
                    dist.all_reduce(Vp)
                del Vp_r, Cg, Vt_r, Wmat
            else:
                ws = ctx.workspace(lib.pmd_projected_svd_factored_workspace_bytes(Rc, m_used, rp, T))
                ctx.call("pmd_projected_svd_factored", ptr(right), Rc, m_used, m_cols, ptr(Et_dev), rp, m_cols, ptr(Z), T, T,
                         None, nk, ptr(s_out), ptr(Vt_out), T, ptr(Vp), T, ptr(X1), ptr(W1), 1 if chol_ok else 0, ptr(ws),
                         ws.numel())
            p_null = None
            if null_tail:
                # the kept null direction: P column = right Et[m-1, :]^T, V row = Et[m-1, :] (right^T Z), its own
                # singular triple (s = |V row|, Vt = V row / s, R column = P column) appended as the last component
                m_full = m_used + 1
                et_row = Et_dev[m_full - 1, :m_full]
                v_null = Vt_out[nk:nk + 1]
                ctx.call("pmd_gemm", 0, 0, 1, T, m_full, 1.0, ptr(et_row), m_full, ptr(W1), T, 0.0, ptr(v_null), T)
                # the joint SVD of the reference would deflate this row against the other right vectors: do the same
                # (two passes of classical Gram-Schmidt against the orthonormal rows of Vt), so that Vt stays orthonormal
                coef = torch.empty((nk,), dtype=torch.float32, device=ctx.device)
                for gs_pass in range(2):
                    ctx.call("pmd_gemm", 0, 0, nk, 1, T, 1.0, ptr(Vt_out), T, ptr(v_null), 1, 0.0, ptr(coef), 1)
                    if gs_pass == 0 and return_diagnostics:
                        coef_norm = coef.norm()
                    ctx.call("pmd_gemm", 0, 0, 1, T, nk, -1.0, ptr(coef), nk, ptr(Vt_out), T, 1.0, ptr(v_null), T)
                s_null = v_null.norm()
                if return_diagnostics:
                    # what the split-off leaves out of R diag(s) Vt: p_null (coef^T Vt), relative to the null row itself
                    null_info["coupling_rel"] = float((coef_norm / torch.clamp(s_null, min=1e-30)).item())
                s_out[nk:nk + 1] = s_null
                v_null.div_(torch.where(s_null == 0, torch.ones_like(s_null), s_null))
            # R = right X1 in row blocks; s, Vt and every finished block go to the host on a side stream
            # while the next block is computed (2.6 GB of results, ~45 ms of PCIe time otherwise serial)
            main = torch.cuda.current_stream(ctx.device)
            side = _side_stream(ctx.device)
            root = dist.rank == 0
            if root:
                r_host = torch.empty((Rc + extra_row, nko), dtype=torch.float32, pin_memory=True)
                s_host = torch.empty((nko,), dtype=torch.float32, pin_memory=True)
                vt_host = torch.empty((nko, T), dtype=torch.float32, pin_memory=True)
                if extra_row:
                    r_host[Rc:].zero_()
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    s_host.copy_(s_out, non_blocking=True)
                    vt_host.copy_(Vt_out, non_blocking=True)

            def download(lo, hi):
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    r_host[lo:hi].copy_(R_out[lo:hi], non_blocking=True)

            # R = right X1.  On rank 0 the GEMM writes its rows straight into the pinned host array (zero copy: the
            # tile stores go over PCIe while the product is still being computed - 83 ms for the 2.2 GB at config 3
            # against 75 ms into HBM plus 39 ms of download, scripts/gemm_probe3.hip).
            r_lo, r_hi = (row_lo, row_hi) if shard else (0, Rc)
            if r_hi > r_lo:
                dst = r_host[r_lo:] if (root and zero_copy) else R_out[r_lo:]
                if null_tail:
                    # last column first (a small product and, on the zero-copy path, a blocking copy: before the large
                    # asynchronous GEMM is enqueued, which writes the other columns of the same rows)
                    p_null = torch.empty((r_hi - r_lo, 1), dtype=torch.float32, device=ctx.device)
                    ctx.call("pmd_gemm", 0, 0, r_hi - r_lo, 1, m_full, 1.0, ptr(right[r_lo:]), m_cols, ptr(et_row), 1, 0.0,
                             ptr(p_null), 1)
                    if root and zero_copy:
                        r_host[r_lo:r_hi, nk:nko].copy_(p_null)
                    else:
                        R_out[r_lo:r_hi, nk:nko] = p_null
                if root and not zero_copy and split_gemm and (r_hi - r_lo) * nk * 4 >= 2 ** 30:
                    n_blk = int(os.environ.get("PMD_R_BLOCKS", "4"))
                    step_r = max(4096, -(-(r_hi - r_lo) // n_blk // 256) * 256)
                    for b_lo in range(r_lo, r_hi, step_r):
                        b_hi = min(r_hi, b_lo + step_r)
                        ctx.call("pmd_gemm", 0, 0, b_hi - b_lo, nk, m_used, 1.0, ptr(right[b_lo:]), m_cols, ptr(X1), rp, 0.0,
                                 ptr(R_out[b_lo:]), nko)
                        download(b_lo, b_hi)
                else:
                    ctx.call("pmd_gemm", 0, 0, r_hi - r_lo, nk, m_used, 1.0, ptr(right[r_lo:]), m_cols, ptr(X1), rp, 0.0,
                             ptr(dst), nko)
                    if root and not zero_copy:
                        download(r_lo, r_hi)
            if shard:
                # the other ranks' row blocks of R travel to rank 0 and are downloaded as they arrive
                ctx.sync()
                bounds = [(int(offsets[lo]), Rc if rr == dist.world - 1 else int(offsets[hi])) for rr, (lo, hi) in enumerate(runs)]
                dist.gather_rows_to_root(R_out, bounds, on_block=download if root else None)
            side.synchronize()
            if root:
                hosts = (r_host.numpy(), s_host.numpy(), vt_host.numpy())
        else:
            if P_dev is None:  # factored P with R' > T (rank_prune corner): materialise P = right Et^T
                P_dev = torch.empty((Rc, m_cols), dtype=torch.float32, device=ctx.device)
                ctx.call("pmd_gemm", 0, 1, Rc, rp, m_used, 1.0, ptr(right), m_cols, ptr(Et_dev), m_cols, 0.0, ptr(P_dev), m_cols)
            ldp = P_dev.shape[1]
            if Vp is None:
                Vp = torch.empty((rp, T), dtype=torch.float32, device=ctx.device)
            ctx.call("pmd_gemm", 1, 0, rp, T, Rc, 1.0, ptr(P_dev), ldp, ptr(Z), T, 0.0, ptr(Vp), T)
            R_out, s_out, Vt_out = _projected_svd_dev(ctx, P_dev, Rc, ldp, Vp, rp, T, T)
        ctx.sync()
        lap("final_svd", t0)
        t0 = time.perf_counter()
        if u_pending is not None:
            _side_stream(ctx.device).synchronize()
            u_host = u_pending[0]
            u_r = scipy.sparse.csr_matrix((u_host[0].numpy(), u_host[1].numpy(), u_host[2].numpy().astype(np.int32)),
                                          shape=(D, R))
            u_pending = None
        root_only = shard and dist.rank != 0   # sharded global stage: the results live on rank 0
        if root_only:
            r_mat = s = vt = None
        elif hosts is not None:
            r_mat, s, vt = hosts
        else:
            r_mat = _to_host(R_out)
            s = _to_host(s_out)
            vt = _to_host(Vt_out)
            if extra_row:
                r_mat = np.concatenate([r_mat, np.zeros((1, r_mat.shape[1]), dtype=r_mat.dtype)], axis=0)
        if not root_only:
            if null_tail and len(s) > 1 and s[-1] > s[-2]:
                # the appended component is normally the smallest; keep `s` descending in any case (as svd returns it)
                order_s = np.argsort(-s, kind="stable")
                r_mat, s, vt = r_mat[:, order_s], s[order_s], vt[order_s]
            good_components = s != 0
            if not np.all(good_components):
                r_mat = r_mat[:, good_components]
                s = s[good_components]
                vt = vt[good_components, :]
        lap("d2h_results", t0)
        display("Matrix decomposition completed")

        if dist.enabled:
            mean_all = torch.zeros(D, dtype=torch.float32, device=ctx.device)
            std_all = torch.zeros(D, dtype=torch.float32, device=ctx.device)
            mean_all[O_lo:O_hi] = mean_dev[O_lo - P_lo:O_hi - P_lo]
            std_all[O_lo:O_hi] = std_dev[O_lo - P_lo:O_hi - P_lo]
            dist.gather_runs(mean_all, owned)
            dist.gather_runs(std_all, owned)
            mean_dev, std_dev = mean_all, std_all
        mean_img = mean_dev.cpu().numpy().reshape(d1, d2)
        std_img = std_dev.cpu().numpy().reshape(d1, d2)
        final_movie = None if root_only else PMDArray(u_r, r_mat, s, vt, (T, d1, d2), order, mean_img, std_img)
        timings["total"] = time.perf_counter() - t_start
        if not return_diagnostics:
            return final_movie
        diag = {
            "seed": seed, "frames": list(frames), "thresholds": (float(spatial_threshold), float(temporal_threshold)),
            "sim_stats": sim_stats, "tile_ranks": tile_ranks_out, "tile_stats": stats_dev.cpu().numpy(),
            "tile_good": good_dev.cpu().numpy(), "tile_keep": keep_dev.cpu().numpy(), "tile_lambda": lam_dev.cpu().numpy(),
            "tile_ut": ut_dev.cpu().numpy().reshape(len(tile_ranks_out), rpad, dpad), "origins": origins[::nvt], "pix": pix_c[::nvt], "block_weights": block_weights,
            "max_components": r, "rank_before": R, "rank_after": rp + (1 if null_tail else 0), "timings": timings,
            "orthogonalizer": ("cholesky" if (use_right and chol_ok) else "eigh"),
            "crop": crop, "dpad": dpad, "v_proj": Vp.cpu().numpy(), "eig_order": min(rp, T), "col_sigma": col_sigma, "n_tile_cols": Rt,
            "null_direction": dict(null_info, split_off=bool(null_tail)),
        }
        return final_movie, diag
    finally:
        if hasattr(ctx, "release_pinned"):
            try:
                ctx.sync()
            except Exception:
                pass
            ctx.release_pinned()
            try:
                # scratch of the large products: kept between calls up to 8 GiB, larger ones are returned to the device
                ctx.call("pmd_scratch_trim", 8 << 30)
            except Exception:
                pass
        if own_ctx:
            ctx.release_workspace()
            ctx.close()
