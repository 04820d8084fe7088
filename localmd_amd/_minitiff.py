"""
Minimal multipage TIFF / BigTIFF reader and writer for uncompressed, strip-based grayscale
pages (8/16/32-bit unsigned or signed integers, 32/64-bit floats).  Stands in for
``tifffile`` (not installed on the build or GPU image) behind ``TiffArray``
(/root/reference/localmd/dataset.py:131-181).  Compressed or tiled files raise
``NotImplementedError``.
"""
import struct

import numpy as np

_TYPE_SIZES = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8, 17: 8, 18: 8}
_TYPE_FMT = {1: "B", 3: "H", 4: "I", 6: "b", 8: "h", 9: "i", 16: "Q", 17: "q", 18: "Q"}


class MiniTiff:
    def __init__(self, filename):
        self.filename = filename
        with open(filename, "rb") as f:
            head = f.read(16)
            if head[:2] == b"II":
                self._e = "<"
            elif head[:2] == b"MM":
                self._e = ">"
            else:
                raise ValueError("not a TIFF file")
            magic = struct.unpack(self._e + "H", head[2:4])[0]
            if magic == 42:
                self._big = False
                offset = struct.unpack(self._e + "I", head[4:8])[0]
            elif magic == 43:
                self._big = True
                offset = struct.unpack(self._e + "Q", head[8:16])[0]
            else:
                raise ValueError("bad TIFF magic")
            self.pages = []
            while offset:
                tags, offset = self._read_ifd(f, offset)
                self.pages.append(tags)
        if not self.pages:
            raise ValueError("TIFF has no pages")
        p0 = self.pages[0]
        self.shape = (len(self.pages), int(p0[257][0]), int(p0[256][0]))

    def _read_ifd(self, f, offset):
        e = self._e
        f.seek(offset)
        if self._big:
            (n,) = struct.unpack(e + "Q", f.read(8))
            entry_size, cnt_fmt, inline = 20, "Q", 8
        else:
            (n,) = struct.unpack(e + "H", f.read(2))
            entry_size, cnt_fmt, inline = 12, "I", 4
        raw = f.read(n * entry_size)
        nxt = struct.unpack(e + cnt_fmt, f.read(inline))[0]
        tags = {}
        for i in range(n):
            ent = raw[i * entry_size : (i + 1) * entry_size]
            tag, typ = struct.unpack(e + "HH", ent[:4])
            (count,) = struct.unpack(e + cnt_fmt, ent[4 : 4 + inline])
            valraw = ent[4 + inline :]
            size = _TYPE_SIZES.get(typ, 1) * count
            if size > inline:
                (ptr,) = struct.unpack(e + cnt_fmt, valraw)
                here = f.tell()
                f.seek(ptr)
                valraw = f.read(size)
                f.seek(here)
            if typ in _TYPE_FMT:
                tags[tag] = struct.unpack(e + _TYPE_FMT[typ] * count, valraw[:size])
            else:
                tags[tag] = valraw[:size]
        return tags, nxt

    def _page_dtype(self, tags):
        bits = tags.get(258, (1,))[0]
        fmt = tags.get(339, (1,))[0]
        kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
        if kind is None or bits not in (8, 16, 32, 64):
            raise NotImplementedError("unsupported sample format")
        return np.dtype(self._e + kind + str(bits // 8))

    def read_page(self, f, idx):
        tags = self.pages[idx]
        if tags.get(259, (1,))[0] != 1:
            raise NotImplementedError("compressed TIFF pages are not supported by the built-in reader")
        if 322 in tags or tags.get(277, (1,))[0] != 1:
            raise NotImplementedError("tiled or multi-sample TIFF pages are not supported")
        h, w = int(tags[257][0]), int(tags[256][0])
        dt = self._page_dtype(tags)
        buf = bytearray()
        for off, cnt in zip(tags[273], tags[279]):
            f.seek(off)
            buf += f.read(cnt)
        return np.frombuffer(bytes(buf), dtype=dt, count=h * w).reshape(h, w)

    def read(self, keys):
        out = None
        with open(self.filename, "rb") as f:
            for n, k in enumerate(keys):
                page = self.read_page(f, k)
                if out is None:
                    out = np.empty((len(keys),) + page.shape, dtype=page.dtype.newbyteorder("="))
                out[n] = page
        return out


def write_tiff(filename, frames: np.ndarray):
    """Write a (T, h, w) array as an uncompressed little-endian multipage TIFF (BigTIFF when
    the data exceed 4 GiB).  One strip per page."""
    frames = np.ascontiguousarray(frames)
    if frames.ndim != 3:
        raise ValueError("expected (T, h, w)")
    kind = {"u": 1, "i": 2, "f": 3}[frames.dtype.kind]
    bits = frames.dtype.itemsize * 8
    T, h, w = frames.shape
    page_bytes = h * w * frames.dtype.itemsize
    big = T * (page_bytes + 256) > (1 << 32) - (1 << 20)
    data = frames.astype(frames.dtype.newbyteorder("<"), copy=False)
    with open(filename, "wb") as f:
        if big:
            f.write(struct.pack("<2sHHHQ", b"II", 43, 8, 0, 16))
        else:
            f.write(struct.pack("<2sHI", b"II", 42, 8))
        pos = f.tell()
        for t in range(T):
            entries = [
                (256, 4, w), (257, 4, h), (258, 3, bits), (259, 3, 1), (262, 3, 1),
                (273, 16 if big else 4, None), (277, 3, 1), (278, 4, h),
                (279, 16 if big else 4, page_bytes), (339, 3, kind),
            ]
            n = len(entries)
            ifd_size = (8 + n * 20 + 8) if big else (2 + n * 12 + 4)
            data_off = pos + ifd_size
            next_ifd = data_off + page_bytes if t + 1 < T else 0
            if big:
                f.write(struct.pack("<Q", n))
            else:
                f.write(struct.pack("<H", n))
            for tag, typ, val in entries:
                if val is None:
                    val = data_off
                if big:
                    f.write(struct.pack("<HHQ", tag, typ, 1))
                    f.write(struct.pack("<" + {3: "H6x", 4: "I4x", 16: "Q"}[typ], val))
                else:
                    f.write(struct.pack("<HHI", tag, typ, 1))
                    f.write(struct.pack("<" + {3: "H2x", 4: "I"}[typ], val))
            f.write(struct.pack("<Q" if big else "<I", next_ifd))
            f.write(data[t].tobytes())
            pos = data_off + page_bytes
