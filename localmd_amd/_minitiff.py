"""
Multipage TIFF / BigTIFF reader and a minimal writer for grayscale movies.  Stands in for
``tifffile`` (not installed on the build or GPU image) behind ``TiffArray``
(/root/reference/localmd/dataset.py:131-181, which reads through ``tifffile.imread(key=...)``).

Layouts the reader handles (one sample per pixel; 8/16/32/64-bit unsigned / signed integers, 32/64-bit floats;
little- and big-endian; classic and BigTIFF):
  * strip-based pages with any number of strips, and tiled pages (tags 322-325; edge tiles cropped);
  * compression 1 (none), 5 (LZW, both bit orders of the code-width switch that TIFF writers use: "early change"),
    8 / 32946 (zlib deflate) and 32773 (PackBits); predictor 2 (horizontal differencing) for integer samples;
  * ImageJ hyperstacks whose single IFD is followed by all frames as contiguous raw data ("images=N" in the
    ImageDescription; ImageJ writes files above 4 GiB this way).
Refused with ``NotImplementedError`` (message names the tag): several samples per pixel (RGB / planar), bit depths that
are not whole bytes (1-, 4-, 12-bit packed), JPEG / CCITT / ZSTD / LERC compressions, floating-point predictor 3,
pages of differing shape or sample type inside one file.
"""
import re
import struct
import zlib

import numpy as np

_TYPE_SIZES = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8, 17: 8, 18: 8}
_TYPE_FMT = {1: "B", 3: "H", 4: "I", 6: "b", 8: "h", 9: "i", 16: "Q", 17: "q", 18: "Q"}
_COMPRESSIONS = {1: "none", 5: "lzw", 8: "deflate", 32946: "deflate", 32773: "packbits"}


def _unpackbits(data: bytes, expected: int) -> bytes:
    """PackBits (TIFF 6.0 section 9): n in 0..127 -> copy n + 1 literal bytes; n in -127..-1 -> repeat the next byte
    1 - n times; -128 is a no-op."""
    out = bytearray()
    i, n = 0, len(data)
    while i < n and len(out) < expected:
        c = data[i]
        i += 1
        if c < 128:
            out += data[i:i + c + 1]
            i += c + 1
        elif c > 128:
            out += data[i:i + 1] * (257 - c)
            i += 1
    return bytes(out[:expected])


def _lzw_decode(data: bytes, expected: int) -> bytes:
    """TIFF LZW (TIFF 6.0 section 13): MSB-first variable-width codes, 9 bits to start, ClearCode 256, EndOfInformation
    257, first free code 258, width grows one code early (when the table holds 511, 1023, 2047 entries)."""
    out = bytearray()
    table = [bytes((i,)) for i in range(256)] + [b"", b""]
    bits, nbits, width = 0, 0, 9
    prev = None
    pos, n = 0, len(data)
    while len(out) < expected:
        while nbits < width and pos < n:
            bits = (bits << 8) | data[pos]
            pos += 1
            nbits += 8
        if nbits < width:
            break
        code = (bits >> (nbits - width)) & ((1 << width) - 1)
        nbits -= width
        bits &= (1 << nbits) - 1
        if code == 256:
            del table[258:]
            width = 9
            prev = None
            continue
        if code == 257:
            break
        if prev is None:
            entry = table[code]
        elif code < len(table):
            entry = table[code]
            table.append(prev + entry[:1])
        else:
            entry = prev + prev[:1]
            table.append(entry)
        out += entry
        prev = entry
        if len(table) >= (1 << width) - 1 and width < 12:
            width += 1
    return bytes(out[:expected])


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
                if tags.get(254, (0,))[0] & 1:
                    continue     # reduced-resolution copy of another page (thumbnail / pyramid level): not a frame
                self.pages.append(tags)
        if not self.pages:
            raise ValueError("TIFF has no pages")
        p0 = self.pages[0]
        self._h, self._w = int(p0[257][0]), int(p0[256][0])
        self._dtype = self._page_dtype(p0)
        self._check_page(p0)
        # ImageJ hyperstack with one IFD: the frames follow each other as raw data from the first strip offset
        self._contiguous = None
        n_pages = len(self.pages)
        desc = p0.get(270, b"")
        if n_pages == 1 and isinstance(desc, bytes) and desc.startswith(b"ImageJ="):
            m = re.search(rb"images=(\d+)", desc)
            if m and int(m.group(1)) > 1 and p0.get(259, (1,))[0] == 1 and 322 not in p0:
                n_pages = int(m.group(1))
                self._contiguous = int(p0[273][0])
        self.shape = (n_pages, self._h, self._w)
        # uncompressed pages whose strips follow each other in the file are read with one readinto per page
        self._flat = [self._flat_span(t) for t in self.pages] if self._contiguous is None else None

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
        kind = {1: "u", 2: "i", 3: "f", 4: "u"}.get(fmt)
        if kind is None or bits not in (8, 16, 32, 64) or (kind == "f" and bits < 32):
            raise NotImplementedError("unsupported TIFF sample format (BitsPerSample {}, SampleFormat {})".format(bits, fmt))
        return np.dtype(self._e + kind + str(bits // 8))

    def _check_page(self, tags):
        comp = tags.get(259, (1,))[0]
        if comp not in _COMPRESSIONS:
            raise NotImplementedError("TIFF compression {} is not supported by the built-in reader "
                                      "(supported: none, LZW, deflate, PackBits)".format(comp))
        if tags.get(277, (1,))[0] != 1:
            raise NotImplementedError("TIFF pages with several samples per pixel (tag 277) are not supported")
        pred = tags.get(317, (1,))[0]
        if pred not in (1, 2) or (pred == 2 and self._page_dtype(tags).kind == "f"):
            raise NotImplementedError("TIFF predictor {} is not supported for this sample type".format(pred))
        if (int(tags[257][0]), int(tags[256][0])) != (self._h, self._w) or self._page_dtype(tags) != self._dtype:
            raise NotImplementedError("TIFF pages of differing shape or sample type in one file")

    def _flat_span(self, tags):
        """(offset, bytes) when the page is uncompressed, strip-based and contiguous in the file, else None."""
        if tags.get(259, (1,))[0] != 1 or 322 in tags or tags.get(317, (1,))[0] != 1 or 273 not in tags:
            return None
        offs, cnts = tags[273], tags.get(279)
        need = self._h * self._w * self._dtype.itemsize
        if cnts is None:
            return (int(offs[0]), need) if len(offs) == 1 else None
        pos = int(offs[0])
        for o, c in zip(offs, cnts):
            if int(o) != pos:
                return None
            pos += int(c)
        return (int(offs[0]), need) if pos - int(offs[0]) >= need else None

    def _decode(self, tags, raw, n_bytes, rows, width):
        """One strip / tile: decompress to n_bytes, undo the predictor; returns a (rows, width) array."""
        comp = _COMPRESSIONS[tags.get(259, (1,))[0]]
        if comp == "deflate":
            raw = zlib.decompress(raw)
        elif comp == "lzw":
            raw = _lzw_decode(raw, n_bytes)
        elif comp == "packbits":
            raw = _unpackbits(raw, n_bytes)
        if len(raw) < n_bytes:
            raise ValueError("TIFF strip / tile shorter than its declared size ({} < {} bytes)".format(len(raw), n_bytes))
        arr = np.frombuffer(raw, dtype=self._dtype, count=rows * width).reshape(rows, width)
        if tags.get(317, (1,))[0] == 2:
            arr = np.cumsum(arr.astype(arr.dtype.newbyteorder("=")), axis=1, dtype=arr.dtype.newbyteorder("="))
        return arr

    def read_page(self, f, idx, out):
        """Frame idx into out (h, w), native byte order."""
        h, w, dt = self._h, self._w, self._dtype
        if self._contiguous is not None:
            f.seek(self._contiguous + idx * h * w * dt.itemsize)
            page = np.frombuffer(f.read(h * w * dt.itemsize), dtype=dt, count=h * w).reshape(h, w)
            out[...] = page
            return
        tags = self.pages[idx]
        if idx > 0:
            self._check_page(tags)
        span = self._flat[idx]
        if span is not None:
            f.seek(span[0])
            if dt.isnative and out.flags.c_contiguous and out.dtype == dt:
                got = f.readinto(memoryview(out).cast("B"))
                if got != span[1]:
                    raise ValueError("TIFF page {} is truncated".format(idx))
            else:
                out[...] = np.frombuffer(f.read(span[1]), dtype=dt, count=h * w).reshape(h, w)
            return
        if 322 in tags:     # tiles, row-major over the page, edge tiles padded to the full tile size
            tw, tl = int(tags[322][0]), int(tags[323][0])
            across = (w + tw - 1) // tw
            for k, (off, cnt) in enumerate(zip(tags[324], tags[325])):
                i0, j0 = (k // across) * tl, (k % across) * tw
                if i0 >= h:
                    break
                f.seek(off)
                tile = self._decode(tags, f.read(cnt), tw * tl * dt.itemsize, tl, tw)
                out[i0:i0 + tl, j0:j0 + tw] = tile[:h - i0, :w - j0]
            return
        rps = int(tags.get(278, (h,))[0])
        rps = h if rps <= 0 or rps > h else rps
        cnts = tags.get(279)
        if cnts is None:   # a single uncompressed strip may omit its byte count
            cnts = (h * w * dt.itemsize,)
        for k, (off, cnt) in enumerate(zip(tags[273], cnts)):
            i0 = k * rps
            rows = min(rps, h - i0)
            if rows <= 0:
                break
            f.seek(off)
            out[i0:i0 + rows] = self._decode(tags, f.read(cnt), rows * w * dt.itemsize, rows, w)

    def read(self, keys):
        """Frames `keys` as a (len(keys), h, w) array in native byte order.  Opens its own file handle: safe to call from
        several threads at once."""
        n_frames = self.shape[0]
        out = np.empty((len(keys), self._h, self._w), dtype=self._dtype.newbyteorder("="))
        with open(self.filename, "rb") as f:
            for n, k in enumerate(keys):
                k = int(k)
                if k < 0:
                    k += n_frames
                if not 0 <= k < n_frames:
                    raise IndexError("frame {} out of range for a TIFF of {} pages".format(k, n_frames))
                self.read_page(f, k, out[n])
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
