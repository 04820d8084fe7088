"""
CPU tests of the host side: tile-grid integers, pooling maps, sparse assembly, PMDArray /
.npz layout, dataset boundary, TIFF reader, and the C-ABI surface of libpmd_hip.so (symbols
only -- no compute call is possible without a GPU, and the product must fail loudly then).
"""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse

import localmd_amd
from localmd_amd import grid
from localmd_amd import _lib
from oracle import pmd_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ grid / integers ----------
@pytest.mark.parametrize("fov,blk", [((512, 512), (20, 20)), ((150, 150), (28, 40)), ((60, 80), (20, 20)),
                                     ((33, 47), (20, 16)), ((20, 20), (20, 20)), ((2048, 64), (16, 16))])
def test_tile_grid_weights_match_oracle(fov, blk):
    assert grid.tile_origins(fov, blk) == O.tile_grid(fov, blk)
    np.testing.assert_array_equal(grid.block_weight_matrix(blk), O.block_weight_matrix(blk))
    it1, it2 = grid.tile_origins(fov, blk)
    pix, origins = grid.tile_pixel_lists(fov, blk, it1, it2)
    assert pix.shape == (len(it1) * len(it2), blk[0] * blk[1])
    # tile order k-outer / j-inner, q = il + b1*jl
    t = len(it2) + 1 if len(it1) > 1 and len(it2) > 1 else 0
    k, j = origins[t]
    assert (k, j) == (it1[t // len(it2)], it2[t % len(it2)])
    q = 3 + blk[0] * 2
    assert pix[t, q] == (k + 3) * fov[1] + (j + 2)
    # weights: partition of unity after normalisation
    bw = grid.block_weight_matrix(blk)
    cw = grid.cumulative_weights(fov, blk, origins, bw)
    assert cw.min() >= 1
    acc = np.zeros(fov)
    for (k, j) in origins:
        acc[k:k + blk[0], j:j + blk[1]] += bw / cw[k:k + blk[0], j:j + blk[1]]
    np.testing.assert_allclose(acc, 1.0, atol=1e-12)


def test_validation_errors():
    with pytest.raises(ValueError):
        grid.check_fov_size((9, 50))
    with pytest.raises(ValueError):
        grid.update_block_sizes((8, 20), (100, 100))
    assert grid.update_block_sizes((40, 20), (30, 100)) == [30, 20]
    with pytest.raises(ValueError):
        grid.block_weight_matrix((21, 20))
    with pytest.raises(ValueError):
        grid.identify_window_chunks(200, 100, 50)
    with pytest.raises(ValueError):
        grid.identify_window_chunks(100, 200, 150)
    np.random.seed(5)
    a = grid.identify_window_chunks(300, 1000, 100)
    np.random.seed(5)
    b = O.identify_window_chunks(300, 1000, 100)
    assert a == b and len(a) == 300


@pytest.mark.parametrize("blk,n", [((20, 20), 2), ((20, 16), 2), ((10, 14), 3), ((20, 20), 1)])
def test_pooling_maps_match_reduce_window(blk, n):
    pool_q, pool_idx, pool_w, shp = grid.pooling_maps(blk, n)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((blk[0], blk[1], 3)).astype(np.float32)
    ref = O.downsample_average_pooling(x, n)
    assert ref.shape[:2] == shp
    flat = x.reshape((blk[0] * blk[1], 3), order="F")
    got = np.zeros((pool_q.shape[0], 3))
    for p in range(pool_q.shape[0]):
        members = pool_q[p][pool_q[p] >= 0]
        got[p] = flat[members].mean(axis=0)
        assert np.all(pool_idx[members] == p)
        np.testing.assert_allclose(pool_w[members], 1.0 / len(members))
    np.testing.assert_allclose(got, ref.reshape((-1, 3), order="F"), atol=1e-6)


def test_overlap_pairs_cover_exactly_the_intersections():
    fov, blk = (50, 46), (20, 20)
    it1, it2 = grid.tile_origins(fov, blk)
    pix, origins = grid.tile_pixel_lists(fov, blk, it1, it2)
    pairs = grid.overlap_pairs(origins, blk)
    masks = []
    for (k, j) in origins:
        m = np.zeros(fov, dtype=bool)
        m[k:k + 20, j:j + 20] = True
        masks.append(m)
    seen = set()
    for a, b, i0, i1, j0, j1 in pairs:
        inter = masks[a] & masks[b]
        rect = np.zeros(fov, dtype=bool)
        rect[i0:i1, j0:j1] = True
        assert a <= b and np.array_equal(inter, rect) and inter.any()
        seen.add((a, b))
    for a in range(len(origins)):
        for b in range(a, len(origins)):
            assert ((masks[a] & masks[b]).any()) == ((a, b) in seen)


def test_sparse_assembly_is_bitwise_the_reference_construction():
    from localmd_amd.decomposition import _sparse_u
    from scipy.sparse import coo_matrix, diags

    rng = np.random.default_rng(0)
    for order in ("F", "C"):
        d1, d2, b = 50, 46, 20
        it1, it2 = grid.tile_origins((d1, d2), (b, b))
        pix, origins = grid.tile_pixel_lists((d1, d2), (b, b), it1, it2)
        n = pix.shape[0]
        ranks = rng.integers(0, 5, n)
        ut = np.zeros((n, 64, 400), dtype=np.float32)
        for t in range(n):
            ut[t, :ranks[t], :] = rng.standard_normal((ranks[t], 400)).astype(np.float32)
        ut[1, 0, 5] = 0.0  # an exact zero must be dropped like dia.dot(coo) does
        bw = grid.block_weight_matrix((b, b))
        fov = np.arange(d1 * d2).reshape((d1, d2), order=order)
        cumw = grid.cumulative_weights((d1, d2), (b, b), origins, bw)
        inv = np.zeros(d1 * d2)
        inv[fov.reshape(-1)] = 1 / cumw.reshape(-1)
        mine, off = _sparse_u(ut, ranks, fov.reshape(-1)[pix], bw, inv, d1 * d2)
        # the reference's construction (decomposition.py:812-853), literally
        rows_l, cols_l, vals_l, col, cw = [], [], [], 0, np.zeros((d1, d2))
        for t, (k, j) in enumerate(origins):
            sp = ut[t, :ranks[t], :].T.reshape((b, b, ranks[t]), order="F").astype(np.float64) * bw[:, :, None]
            cw[k:k + b, j:j + b] += bw
            ridx = fov[k:k + b, j:j + b][:, :, None] + np.zeros((1, 1, ranks[t]))
            cidx = np.zeros_like(ridx) + np.arange(col, col + ranks[t])[None, None, :]
            rows_l += ridx.flatten().tolist()
            cols_l += cidx.flatten().tolist()
            vals_l += sp.flatten().tolist()
            col += ranks[t]
        ref = coo_matrix((vals_l, (rows_l, cols_l)), shape=(d1 * d2, col))
        wnd = np.zeros(d1 * d2)
        wnd[fov.flatten(order=order)] = cw.flatten(order=order)
        ref = diags([(1 / wnd).ravel()], [0]).dot(ref).tocsr()
        ref.sort_indices()
        a = mine.tocsr()
        a.sort_indices()
        assert a.shape == ref.shape
        np.testing.assert_array_equal(a.indptr, ref.indptr)
        np.testing.assert_array_equal(a.indices, ref.indices)
        np.testing.assert_array_equal(a.data, ref.data)


# ------------------------------------------------------------------ PMDArray / npz ----------
def _toy_pmd(order="F"):
    rng = np.random.default_rng(1)
    d1, d2, T, R, K = 9, 7, 15, 6, 4
    u = scipy.sparse.random(d1 * d2, R, density=0.3, random_state=2, format="coo")
    r = rng.standard_normal((R, K)).astype(np.float32)
    s = np.abs(rng.standard_normal(K)).astype(np.float32)
    v = rng.standard_normal((K, T)).astype(np.float32)
    mean = rng.standard_normal((d1, d2)).astype(np.float32)
    std = (1 + rng.random((d1, d2))).astype(np.float32)
    arr = localmd_amd.PMDArray(u, r, s, v, (T, d1, d2), order, mean, std)
    dense = (u @ (r * s) @ v).reshape((d1, d2, T), order=order) * std[:, :, None] + mean[:, :, None]
    return arr, dense.transpose(2, 0, 1).astype(np.float32)


@pytest.mark.parametrize("order", ["F", "C"])
def test_pmdarray_indexing(order):
    arr, dense = _toy_pmd(order)
    assert arr.shape == dense.shape and arr.ndim == 3 and arr.dtype == np.float32
    np.testing.assert_allclose(arr[3, :, :], dense[3], atol=1e-5)
    np.testing.assert_allclose(arr[:, 2, 5], dense[:, 2, 5], atol=1e-5)
    np.testing.assert_allclose(arr[2:6, 1:4, 2:5], dense[2:6, 1:4, 2:5], atol=1e-5)
    np.testing.assert_allclose(arr[[1, 4]], dense[[1, 4]], atol=1e-5)
    np.testing.assert_allclose(arr[:, 2:4], dense[:, 2:4], atol=1e-5)  # two-key form (reference: TypeError)
    with pytest.raises(ValueError):
        arr[None]
    with pytest.raises(ValueError):
        arr[1, 2, 3, 4]
    assert scipy.sparse.isspmatrix_csr(arr.u) and arr.u.has_sorted_indices
    for name in ("r", "s", "v", "mean_img", "var_img", "order", "row_indices"):
        assert hasattr(arr, name)


def test_npz_layout_roundtrip(tmp_path):
    arr, dense = _toy_pmd()
    fn = str(tmp_path / "out.npz")
    localmd_amd.save_npz(fn, arr)
    with np.load(fn, allow_pickle=True) as data:
        assert set(data.files) == {"fov_shape", "fov_order", "U_data", "U_indices", "U_indptr", "U_shape", "U_format", "R",
                                   "s", "Vt", "mean_img", "noise_var_img"}
        assert data["U_data"].dtype == np.float64 and data["U_indices"].dtype == np.int32
        assert data["U_indptr"].dtype == np.int32 and data["R"].dtype == np.float32
        assert data["fov_order"].item() == "F" and tuple(data["fov_shape"]) == (9, 7)
        # the reference's own loader (README.md:48-62)
        U = scipy.sparse.csr_matrix((data["U_data"], data["U_indices"], data["U_indptr"]), shape=data["U_shape"]).tocoo()
        assert U.shape == arr.u.shape
    back = localmd_amd.load_npz(fn)
    np.testing.assert_allclose(back[:, :, :], dense, atol=1e-5)


# ------------------------------------------------------------------ dataset boundary --------
def test_lazy_data_loader_contract(tmp_path):
    rng = np.random.default_rng(3)
    mov = rng.standard_normal((12, 6, 5)).astype(np.float32)
    ds = localmd_amd.ArrayDataset(mov)
    assert ds.shape == (12, 6, 5) and ds.ndim == 3
    np.testing.assert_array_equal(ds[[0, 3, 5]], mov[[0, 3, 5]])
    np.testing.assert_array_equal(ds[np.array([1, 2])], mov[[1, 2]])
    np.testing.assert_array_equal(ds[4], mov[4])
    np.testing.assert_array_equal(ds[2:7], mov[2:7])
    np.testing.assert_array_equal(ds[range(2, 8, 2)], mov[2:8:2])
    np.testing.assert_array_equal(ds[1:4, 2], mov[1:4, 2])
    np.testing.assert_array_equal(ds[1:4, 2:4, 1], mov[1:4, 2:4, 1])
    with pytest.raises(IndexError):
        ds[0:13]
    with pytest.raises(IndexError):
        ds["a"]
    with pytest.raises(IndexError):
        ds[0, 0, 0, 0]
    assert localmd_amd.PMDDataset is localmd_amd.lazy_data_loader


@pytest.mark.parametrize("dt", [np.uint16, np.float32, np.int16, np.uint8])
def test_tiff_reader_roundtrip(tmp_path, dt):
    from localmd_amd._minitiff import write_tiff, MiniTiff

    rng = np.random.default_rng(4)
    frames = (rng.random((7, 11, 13)) * 200).astype(dt)
    fn = str(tmp_path / "m.tif")
    write_tiff(fn, frames)
    rd = MiniTiff(fn)
    assert rd.shape == (7, 11, 13)
    np.testing.assert_array_equal(rd.read([0, 6, 3]), frames[[0, 6, 3]])
    ta = localmd_amd.TiffArray(fn)
    assert ta.shape == (7, 11, 13)
    out = ta[[1, 2]]
    assert out.dtype == np.float32
    np.testing.assert_array_equal(out, frames[[1, 2]].astype(np.float32))
    np.testing.assert_array_equal(ta[2:5], frames[2:5].astype(np.float32))


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("name,key", [
    ("pillow_u16_raw.tif", "u16"), ("pillow_u16_lzw.tif", "u16"), ("pillow_u16_deflate.tif", "u16"),
    ("pillow_u16_packbits.tif", "u16"), ("pillow_u16_lzw_pred.tif", "u16"), ("pillow_u8_raw.tif", "u8"),
    ("pillow_f32_raw.tif", "f32"), ("pillow_u16_strips_raw.tif", "wide"), ("pillow_u16_strips_lzw.tif", "wide"),
    ("pillow_u16_movie_deflate.tif", "movie")])
def test_tiff_fixtures_written_by_pillow(name, key):
    """files written by an independent encoder (Pillow / libtiff, tests/golden/make_tiff_fixtures.py) -
    multipage, several strips per page, LZW / deflate / PackBits, horizontal predictor - read pixel-exact through
    TiffArray (the reference reads them through tifffile, dataset.py:131-181)."""
    expected = np.load(os.path.join(GOLDEN, "tiff_fixture_expected.npz"))[key]
    ta = localmd_amd.TiffArray(os.path.join(GOLDEN, name))
    assert ta.shape == expected.shape and ta.ndim == 3
    got = ta[list(range(expected.shape[0]))]
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, expected.astype(np.float32))
    pick = [expected.shape[0] - 1, 0, 2]
    np.testing.assert_array_equal(ta[pick], expected[pick].astype(np.float32))
    np.testing.assert_array_equal(ta[1:3, 2:7, 3], expected[1:3, 2:7, 3].astype(np.float32))
    np.testing.assert_array_equal(ta[-1], expected[-1].astype(np.float32))


def _craft_tiff(path, frames, endian="<", tiles=None, big=False, compression=1, imagej=False):
    """A second, test-local TIFF writer for the layouts Pillow does not produce: big-endian, BigTIFF, tiled pages,
    deflate-compressed tiles, ImageJ's single-IFD hyperstack."""
    import struct
    import zlib

    T, h, w = frames.shape
    dt = frames.dtype.newbyteorder(endian)
    kind = {"u": 1, "i": 2, "f": 3}[frames.dtype.kind]
    ifd_fmt = (endian + "HHQ8s") if big else (endian + "HHI4s")
    blobs = []          # page payloads, written first
    with open(path, "wb") as f:
        f.write(struct.pack(endian + "2sHHHQ", b"II" if endian == "<" else b"MM", 43, 8, 0, 0) if big else
                struct.pack(endian + "2sHI", b"II" if endian == "<" else b"MM", 42, 0))
        pages = []
        for t in range(1 if imagej else T):
            offs, cnts = [], []
            if imagej:
                chunks = [frames.astype(dt).tobytes()]
            elif tiles:
                tl, tw = tiles
                chunks = []
                for i0 in range(0, h, tl):
                    for j0 in range(0, w, tw):
                        tile = np.zeros((tl, tw), dtype=dt)
                        blk = frames[t, i0:i0 + tl, j0:j0 + tw]
                        tile[:blk.shape[0], :blk.shape[1]] = blk
                        chunks.append(tile.tobytes())
            else:
                chunks = [frames[t].astype(dt).tobytes()]
            for c in chunks:
                if compression == 8:
                    c = zlib.compress(c)
                offs.append(f.tell())
                cnts.append(len(c))
                f.write(c)
                if f.tell() % 2:
                    f.write(b"\0")
            pages.append((offs, cnts))
        prev_next_pos = 8 if big else 4
        for t, (offs, cnts) in enumerate(pages):
            n_chunks = len(offs)
            arr_pos = f.tell()
            off_t = 16 if big else 4
            f.write(struct.pack(endian + ("Q" if big else "I") * n_chunks, *offs))
            f.write(struct.pack(endian + ("Q" if big else "I") * n_chunks, *cnts))
            desc = b"ImageJ=1.53\nimages=%d\nframes=%d\n\0" % (T, T) if imagej else b""
            desc_pos = f.tell()
            f.write(desc)
            if f.tell() % 2:
                f.write(b"\0")
            ifd_pos = f.tell()
            here = f.tell()
            f.seek(prev_next_pos)
            f.write(struct.pack(endian + ("Q" if big else "I"), ifd_pos))
            f.seek(here)

            def val(typ, count, v):
                size = {3: 2, 4: 4, 16: 8, 2: 1}[typ] * count
                room = 8 if big else 4
                if size <= room:
                    if typ == 2:
                        return v.ljust(room, b"\0")
                    return struct.pack(endian + {3: "H", 4: "I", 16: "Q"}[typ] * count, *(v if count > 1 else [v])).ljust(room, b"\0")
                return struct.pack(endian + ("Q" if big else "I"), v)   # v is then a file offset

            size_t = 8 if big else 4
            ents = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, 8 * frames.dtype.itemsize), (259, 3, 1, compression),
                    (262, 3, 1, 1), (277, 3, 1, 1), (339, 3, 1, kind)]
            one = n_chunks == 1
            if tiles:
                ents += [(322, 4, 1, tiles[1]), (323, 4, 1, tiles[0]),
                         (324, off_t, n_chunks, offs[0] if one else arr_pos),
                         (325, off_t, n_chunks, cnts[0] if one else arr_pos + n_chunks * size_t)]
            else:
                ents += [(273, off_t, n_chunks, offs[0] if one else arr_pos), (278, 4, 1, h),
                         (279, off_t, n_chunks, cnts[0] if one else arr_pos + n_chunks * size_t)]
            if imagej:
                ents.append((270, 2, len(desc), desc_pos))
            ents.sort()
            f.write(struct.pack(endian + ("Q" if big else "H"), len(ents)))
            for tag, typ, count, v in ents:
                if n_chunks > 1 and tag in (273, 279, 324, 325) or tag == 270:
                    payload = struct.pack(endian + ("Q" if big else "I"), v).ljust(8 if big else 4, b"\0")
                else:
                    payload = val(typ, count, v)
                f.write(struct.pack(endian + "HH", tag, typ) + struct.pack(endian + ("Q" if big else "I"), count) + payload)
            prev_next_pos = f.tell()
            f.write(struct.pack(endian + ("Q" if big else "I"), 0))


@pytest.mark.parametrize("kw", [dict(endian=">"), dict(big=True), dict(tiles=(16, 16)), dict(tiles=(16, 32), compression=8),
                                dict(endian=">", tiles=(16, 16), big=True), dict(imagej=True), dict(compression=8, endian=">")])
@pytest.mark.parametrize("dt", [np.uint16, np.float32])
def test_tiff_reader_layouts(tmp_path, kw, dt):
    rng = np.random.default_rng(5)
    frames = (rng.random((5, 37, 41)) * 3000).astype(dt)
    fn = str(tmp_path / "c.tif")
    _craft_tiff(fn, frames, **kw)
    ta = localmd_amd.TiffArray(fn)
    assert ta.shape == frames.shape
    np.testing.assert_array_equal(ta[[4, 0, 2]], frames[[4, 0, 2]].astype(np.float32))


def test_tiff_reader_refuses_what_it_cannot_decode(tmp_path):
    """Unsupported layouts fail loudly with the tag named (INTEGRATION.md lists them), never with wrong pixels."""
    import struct
    from localmd_amd._minitiff import MiniTiff, write_tiff

    frames = np.arange(2 * 6 * 8, dtype=np.uint16).reshape(2, 6, 8)
    fn = str(tmp_path / "x.tif")
    write_tiff(fn, frames)
    raw = bytearray(open(fn, "rb").read())

    def patched(tag, value):
        out = bytearray(raw)
        (n,) = struct.unpack("<H", out[8:10])
        for i in range(n):
            pos = 10 + 12 * i
            if struct.unpack("<H", out[pos:pos + 2])[0] == tag:
                out[pos + 8:pos + 10] = struct.pack("<H", value)
        path = str(tmp_path / "p{}_{}.tif".format(tag, value))
        open(path, "wb").write(out)
        return path

    for tag, value in ((259, 7), (259, 50000), (277, 3), (258, 12)):
        with pytest.raises(NotImplementedError):
            MiniTiff(patched(tag, value))
    with pytest.raises(IndexError):
        MiniTiff(fn).read([2])
    with pytest.raises(ValueError):
        open(str(tmp_path / "n.tif"), "wb").write(b"not a tiff at all")
        MiniTiff(str(tmp_path / "n.tif"))


def test_lazy_loader_is_never_reentered_by_default():
    """num_workers = 0 means in-process single-threaded loading in the reference (pmd_loader.py:161-168).  A
    user's loader (shared file handle, non-re-entrant decoder) must be read by one thread at a time unless the caller
    passes num_workers > 0 or the loader declares thread_safe."""
    import threading
    import time as _time
    import types
    import torch
    from localmd_amd.dataset import lazy_data_loader
    from localmd_amd.decomposition import _Movie

    rng = np.random.default_rng(1)
    data = rng.random((64, 9, 7)).astype(np.float32)

    class Guarded(lazy_data_loader):
        def __init__(self):
            self.inside = 0
            self.max_inside = 0
            self.lock = threading.Lock()

        @property
        def dtype(self):
            return np.float32

        @property
        def shape(self):
            return data.shape

        def _compute_at_indices(self, idx):
            with self.lock:
                self.inside += 1
                self.max_inside = max(self.max_inside, self.inside)
            _time.sleep(0.002)
            out = data[idx]
            with self.lock:
                self.inside -= 1
            return out

    fake_ctx = types.SimpleNamespace(device=torch.device("cpu"))
    src = Guarded()
    mv = _Movie(fake_ctx, src, 16)
    np.testing.assert_array_equal(mv.dev.numpy(), data.reshape(64, -1))
    assert src.max_inside == 1
    src = Guarded()
    mv = _Movie(fake_ctx, src, 16, num_workers=4)      # the caller asked for concurrent readers
    np.testing.assert_array_equal(mv.dev.numpy(), data.reshape(64, -1))
    assert src.max_inside >= 1
    src = Guarded()
    src.thread_safe = True
    mv = _Movie(fake_ctx, src, 16)
    np.testing.assert_array_equal(mv.dev.numpy(), data.reshape(64, -1))


# ------------------------------------------------------------------ C ABI surface -----------
def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libpmd_hip.so missing: run localmd_amd/csrc/build.sh"
    header = open(os.path.join(ROOT, "include", "pmd_hip.h")).read()
    declared = set(re.findall(r"\b(pmdk?_[a-z0-9_]+)\s*\(", header))
    declared.discard("pmd_ctx")
    assert len(declared) >= 35
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    loaded = _lib.load()
    assert loaded.pmd_version() == 1
    # host-only helpers can be called without a device
    assert loaded.pmd_tile_dpad(400) == 400 and loaded.pmd_tile_dpad(100) == 256 and loaded.pmd_tile_dpad(1024) == 1024
    assert loaded.pmd_tile_dpad(1600) == 2048 and loaded.pmd_tile_dpad(3000) == 3072 and loaded.pmd_tile_dpad(70000) == -1
    assert loaded.pmd_time_ld(10000) == 10048 + 64 and loaded.pmd_time_ld(64) == 128
    assert loaded.pmd_tiles_workspace_bytes(2601, 20, 20, 100, 50, 10, 10000, 10112, 262144) > 0


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    with pytest.raises(_lib.PMDLibraryError):
        _lib.Context(0)
    mov = np.zeros((300, 20, 20), dtype=np.float32)
    with pytest.raises(_lib.PMDLibraryError):
        localmd_amd.localmd_decomposition(mov, (20, 20), 300)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "localmd_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the oracle", "").replace("oracle/", ""), fn


def test_pmdarray_to_device_needs_a_gpu():
    """No CPU stand-in for the device expansion: without a HIP device to_device() raises."""
    import torch
    import scipy.sparse
    from localmd_amd.pmdarray import PMDArray
    from localmd_amd._lib import PMDLibraryError

    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    u = scipy.sparse.coo_matrix(np.eye(6, 3))
    arr = PMDArray(u, np.eye(3, dtype=np.float32), np.ones(3, np.float32), np.ones((3, 4), np.float32), (4, 2, 3), "F",
                   np.zeros((2, 3), np.float32), np.ones((2, 3), np.float32))
    with pytest.raises(PMDLibraryError):
        arr.to_device()
    assert arr[0].shape == (2, 3)


def test_movie_upload_handles_one_frame_batches():
    """lazy_data_loader.__getitem__ squeezes (dataset.py:114): a trailing batch of exactly one frame comes back 2-D
    (T % frame_batch_size == 1).  The upload loop must not lose the frame axis."""
    import types
    import torch
    from localmd_amd.dataset import ArrayDataset
    from localmd_amd.decomposition import _Movie

    rng = np.random.default_rng(0)
    data = rng.random((11, 12, 13)).astype(np.float32)
    fake_ctx = types.SimpleNamespace(device=torch.device("cpu"))
    for fbs in (5, 10, 1, 11, 50):
        mv = _Movie(fake_ctx, ArrayDataset(data), fbs)
        np.testing.assert_array_equal(mv.dev.numpy(), data.reshape(11, -1))
    mv = _Movie(fake_ctx, ArrayDataset(data), 10, rows=(3, 9))
    np.testing.assert_array_equal(mv.dev.numpy(), data[:, 3:9, :].reshape(11, -1))
