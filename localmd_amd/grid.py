"""
Integer bookkeeping of the tile grid (host side, exact): tile origins, pyramid weights,
pixel lists, pooling windows, overlap pairs, frame sampling, argument validation.
Mirrors /root/reference/localmd/decomposition.py:528-635 and :695-754.
"""
import math

import numpy as np


def check_fov_size(fov_dims, min_allowed_value: int = 10) -> None:
    """ValueError when a FOV dimension is below 10 (decomposition.py:616-635)."""
    for k in fov_dims:
        if k < min_allowed_value:
            raise ValueError(
                "At least one FOV dimension is lower than {}, too small to process".format(min_allowed_value))


def update_block_sizes(blocks, fov_shape, min_block_value: int = 10, display=None) -> list:
    """Clamp block sizes to the FOV; ValueError below 10 (decomposition.py:572-613)."""
    if blocks[0] < min_block_value or blocks[1] < min_block_value:
        raise ValueError(
            "One of the block dimensions was less than min allowed value of {}, "
            "set to a larger value".format(min_block_value))
    out = []
    for dim in (0, 1):
        if blocks[dim] > fov_shape[dim]:
            if display is not None:
                display("Height blocksize was set to {} but corresponding dimension has size {}. Truncating to {}".format(
                    blocks[dim], fov_shape[dim], fov_shape[dim]))
            out.append(int(fov_shape[dim]))
        else:
            out.append(int(blocks[dim]))
    return out


def identify_window_chunks(frame_range: int, total_frames: int, window_chunks: int, display=None) -> list:
    """Contiguous chunks of frames to fit on; draws from the global np.random state exactly like
    the reference (decomposition.py:528-569)."""
    if frame_range > total_frames:
        raise ValueError("Requested more frames than available")
    if window_chunks > frame_range:
        raise ValueError("The size of each temporal chunk is bigger than frame range")
    num_intervals = math.ceil(frame_range / window_chunks)
    available = np.arange(0, total_frames, window_chunks)
    if available[-1] > total_frames - window_chunks:
        available[-1] = total_frames - window_chunks
    starts = np.sort(np.random.choice(available, size=num_intervals, replace=False))
    if display is not None:
        display("sampled from the following regions: {}".format(starts))
    frames = []
    for k in starts:
        frames.extend(range(int(k), int(min(k + window_chunks, total_frames))))
    return frames


def tile_origins(fov, block_sizes):
    """Origins per dimension: stride b - ceil(b/2), last tile snapped to D - b
    (decomposition.py:698, :723-739)."""
    out = []
    for dim in (0, 1):
        overlap = math.ceil(block_sizes[dim] / 2)
        it = list(range(0, fov[dim] - block_sizes[dim] + 1, block_sizes[dim] - overlap))
        if it[-1] != fov[dim] - block_sizes[dim] and fov[dim] - block_sizes[dim] != 0:
            it.append(fov[dim] - block_sizes[dim])
        out.append(it)
    return out[0], out[1]


def block_weight_matrix(block_sizes, dtype=np.float32) -> np.ndarray:
    """Pyramid weights (decomposition.py:742-750).  Odd block sizes are rejected with a
    ValueError (the reference fails there with a NumPy broadcast ValueError)."""
    b1, b2 = int(block_sizes[0]), int(block_sizes[1])
    if b1 % 2 or b2 % 2:
        raise ValueError("block sizes must be even (got {} x {})".format(b1, b2))
    hbh, hbw = b1 // 2, b2 // 2
    w = np.ones((b1, b2), dtype=dtype)
    w[:hbh, :hbw] += np.minimum(np.arange(hbw)[None, :], np.arange(hbh)[:, None])
    w[:hbh, hbw:] = np.fliplr(w[:hbh, :hbw])
    w[hbh:, :] = np.flipud(w[:hbh, :])
    return w


def tile_pixel_lists(fov, block_sizes, dim_1_iters, dim_2_iters):
    """pix[tile][q]: C-order FOV pixel (i*d2 + j) of tile-local pixel q = il + b1*jl.
    Tile order is k-outer / j-inner (decomposition.py:790-792).  Also returns origins[tile].
    (One broadcast instead of a Python loop over the tiles: 52 ms of host time at 16 129 tiles, on the critical path.)"""
    d1, d2 = fov
    b1, b2 = block_sizes
    k = np.asarray(dim_1_iters, dtype=np.int64)
    j = np.asarray(dim_2_iters, dtype=np.int64)
    il = np.arange(b1, dtype=np.int64)
    jl = np.arange(b2, dtype=np.int64)
    # axes (tile row, tile column, jl, il): flattening the last two gives q = jl * b1 + il
    c = (k[:, None, None, None] + il[None, None, None, :]) * d2 + (j[None, :, None, None] + jl[None, None, :, None])
    pix = np.ascontiguousarray(c.reshape(len(k) * len(j), b1 * b2), dtype=np.int32)
    origins = np.stack(np.meshgrid(k, j, indexing="ij"), axis=-1).reshape(-1, 2).astype(np.int32)
    return pix, origins


def pooling_maps(block_sizes, n):
    """n x n mean pooling with XLA SAME padding (decomposition.py:192-232).
    Returns pool_q (P, n*n) local pixel ids (F order, -1 = out of bounds), pool_idx (d,) window
    of each pixel (windows in F order), pool_w (d,) = 1/|window|, and the pooled shape."""
    b1, b2 = block_sizes

    def windows(size):
        out = -(-size // n)
        total = max((out - 1) * n + n - size, 0)
        lo = total // 2
        return out, lo

    o1, lo1 = windows(b1)
    o2, lo2 = windows(b2)
    P = o1 * o2
    pool_q = -np.ones((P, n * n), dtype=np.int32)
    pool_idx = np.zeros(b1 * b2, dtype=np.int32)
    pool_w = np.zeros(b1 * b2, dtype=np.float32)
    for pj in range(o2):
        for pi in range(o1):
            p = pi + o1 * pj
            members = []
            for a in range(n):
                for b in range(n):
                    i = pi * n + a - lo1
                    j = pj * n + b - lo2
                    if 0 <= i < b1 and 0 <= j < b2:
                        members.append(i + b1 * j)
            pool_q[p, : len(members)] = members
            for q in members:
                pool_idx[q] = p
                pool_w[q] = 1.0 / len(members)
    return pool_q, pool_idx, pool_w, (o1, o2)


def overlap_pairs(origins, block_sizes):
    """All (a <= b) tile pairs whose rectangles intersect, with the intersection rectangle
    (i0, i1, j0, j1) in FOV coordinates, ordered by (a, b).  int32 (n_pairs, 6).
    `origins` is the row-major product grid of the tile-row and tile-column origins (tile_origins)."""
    b1, b2 = block_sizes
    origins = np.asarray(origins)
    ks = np.unique(origins[:, 0]).astype(np.int64)
    js = np.unique(origins[:, 1]).astype(np.int64)
    n2 = len(js)
    if len(ks) * n2 != len(origins):
        raise ValueError("origins must be the full product grid of tile-row and tile-column origins")

    def near(o, b):
        # index pairs (p, q) of origins closer than one block: their tiles share pixels along this axis
        p, q = np.nonzero(np.abs(o[:, None] - o[None, :]) < b)
        return p, q

    ra, rb = near(ks, b1)
    ca, cb = near(js, b2)
    ta = (ra[:, None] * n2 + ca[None, :]).reshape(-1)
    tb = (rb[:, None] * n2 + cb[None, :]).reshape(-1)
    i0 = np.broadcast_to(np.maximum(ks[ra], ks[rb])[:, None], (len(ra), len(ca))).reshape(-1)
    i1 = np.broadcast_to(np.minimum(ks[ra], ks[rb])[:, None] + b1, (len(ra), len(ca))).reshape(-1)
    j0 = np.broadcast_to(np.maximum(js[ca], js[cb])[None, :], (len(ra), len(ca))).reshape(-1)
    j1 = np.broadcast_to(np.minimum(js[ca], js[cb])[None, :] + b2, (len(ra), len(ca))).reshape(-1)
    keep = tb >= ta
    out = np.stack([ta, tb, i0, i1, j0, j1], axis=1)[keep]
    out = out[np.lexsort((out[:, 1], out[:, 0]))]
    return np.ascontiguousarray(out, dtype=np.int32).reshape(-1, 6)


def virtual_pairs(pairs, vt):
    """overlap_pairs for tiles split into vt blocks of 64 component rows each ("virtual tiles" t * vt + h, all with the
    pixels of tile t): every pair (a, b) becomes the vt x vt pairs of their blocks, a pair (a, a) the pairs (h <= g) of its
    own blocks; same intersection rectangles, ordered by (a, b)."""
    if vt == 1:
        return pairs
    pairs = np.asarray(pairs, dtype=np.int64)
    h, g = np.meshgrid(np.arange(vt), np.arange(vt), indexing="ij")
    h, g = h.reshape(-1), g.reshape(-1)
    ta = (pairs[:, 0:1] * vt + h[None, :]).reshape(-1)
    tb = (pairs[:, 1:2] * vt + g[None, :]).reshape(-1)
    rect = np.repeat(pairs[:, 2:], vt * vt, axis=0)
    keep = tb >= ta
    out = np.concatenate([ta[:, None], tb[:, None], rect], axis=1)[keep]
    out = out[np.lexsort((out[:, 1], out[:, 0]))]
    return np.ascontiguousarray(out, dtype=np.int32).reshape(-1, 6)


def cumulative_weights(fov, block_sizes, origins, block_weights):
    """Sum of the tile weights covering each pixel (decomposition.py:813-816), float64 (d1, d2).  The origins form a product
    grid, so the sum over the tile columns is the same strip for every tile row: n1 + n2 placements instead of n1 * n2
    (the weights are small integers: the float64 sums are exact in any order)."""
    b1, b2 = block_sizes
    origins = np.asarray(origins)
    ks = np.unique(origins[:, 0])
    js = np.unique(origins[:, 1])
    if len(ks) * len(js) != len(origins):
        cw = np.zeros(fov, dtype=np.float64)
        for k, j in origins:
            cw[k : k + b1, j : j + b2] += block_weights
        return cw
    strip = np.zeros((b1, fov[1]), dtype=np.float64)
    for j in js:
        strip[:, j : j + b2] += block_weights
    cw = np.zeros(fov, dtype=np.float64)
    for k in ks:
        cw[k : k + b1, :] += strip
    return cw


def cover_tables(fov, block_sizes, dim_1_iters, dim_2_iters):
    """cover1[i][0..3]: indices into dim_1_iters of the tile rows covering FOV row i (ascending,
    -1 padded); cover2 likewise for columns.  int32 (d1, 4), (d2, 4)."""
    out = []
    for size, b, its in ((fov[0], block_sizes[0], dim_1_iters), (fov[1], block_sizes[1], dim_2_iters)):
        cov = -np.ones((size, 4), dtype=np.int32)
        fill = np.zeros(size, dtype=np.int64)
        for idx, o in enumerate(its):
            rows = np.arange(o, o + b)
            if np.any(fill[rows] >= 4):
                raise ValueError("a pixel is covered by more than 4 tile origins along one dimension")
            cov[rows, fill[rows]] = idx
            fill[rows] += 1
        out.append(cov)
    return out[0], out[1]


def neighbour_lists(pairs, ranks, col_off, n_tiles, Rt, K):
    """CSR-like lists for pmd_gram_apply: for tile a, every block of G in its block row:
    (first M row of the block, rows in it, block index, flags) with flags bit0 = stored block is
    (b, a) so it must be used transposed, bit1 = background block."""
    a = pairs[:, 0].astype(np.int64)
    b = pairs[:, 1].astype(np.int64)
    idx = np.arange(pairs.shape[0], dtype=np.int64)
    off_diag = a != b
    rows = np.concatenate([a, b[off_diag]])
    cols = np.concatenate([b, a[off_diag]])
    blk = np.concatenate([idx, idx[off_diag]])
    flg = np.concatenate([np.zeros(len(a), dtype=np.int64), np.ones(int(off_diag.sum()), dtype=np.int64)])
    # background blocks: one per tile and per block of 64 background columns (column index -1 - kb keeps them apart from the
    # tile columns and in block order); block index = kb * n_tiles + tile, as pmd_gram_blocks lays them out
    for kb in range((max(K, 0) + 63) // 64):
        t = np.arange(n_tiles, dtype=np.int64)
        rows = np.concatenate([rows, t])
        cols = np.concatenate([cols, np.full(n_tiles, -1 - kb, dtype=np.int64)])
        blk = np.concatenate([blk, kb * n_tiles + t])
        flg = np.concatenate([flg, np.full(n_tiles, 2, dtype=np.int64)])
    order = np.lexsort((cols, rows))
    rows, cols, blk, flg = rows[order], cols[order], blk[order], flg[order]
    kb_of = np.maximum(-1 - cols, 0)
    row0 = np.where(cols >= 0, np.asarray(col_off)[np.maximum(cols, 0)], Rt + 64 * kb_of)
    nrow = np.where(cols >= 0, np.asarray(ranks)[np.maximum(cols, 0)], np.minimum(64, K - 64 * kb_of))
    nbr = np.stack([row0, nrow, blk, flg], axis=1).astype(np.int32)
    ptr = np.zeros(n_tiles + 1, dtype=np.int64)
    np.add.at(ptr, rows + 1, 1)
    return np.cumsum(ptr).astype(np.int32), nbr
