"""
Writes the independent TIFF fixtures of tests/golden/ with Pillow (libtiff underneath) - files that the
repository's own writer (localmd_amd/_minitiff.write_tiff) had no part in - together with the pixel values
they hold (tiff_fixture_expected.npz).  TiffArray must read them pixel-exact
(tests/test_host_logic.py::test_tiff_fixtures_written_by_pillow).

    python tests/golden/make_tiff_fixtures.py

Layouts: multipage uint16 - uncompressed (several strips per page: libtiff cuts pages into 8 KiB strips), LZW,
Adobe deflate, PackBits; LZW + horizontal predictor; uint8 and float32 uncompressed.  The movie is a small
synthetic calcium-imaging-like stack (smooth blobs x spiky traces + noise), 24 frames of 40 x 52 pixels, so that
the compressed files stay a few tens of KB.
"""
import os
import sys

import numpy as np
from PIL import Image, TiffImagePlugin

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def movie():
    rng = np.random.default_rng(20261004)
    T, h, w = 24, 40, 52
    yy, xx = np.mgrid[0:h, 0:w]
    mov = np.full((T, h, w), 400.0)
    for _ in range(6):
        ci, cj = rng.uniform(4, h - 4), rng.uniform(4, w - 4)
        blob = np.exp(-((yy - ci) ** 2 + (xx - cj) ** 2) / 18.0)
        trace = np.convolve((rng.random(T) < 0.2) * rng.uniform(200, 900), np.exp(-np.arange(8) / 3.0))[:T]
        mov += trace[:, None, None] * blob[None]
    mov += rng.normal(0, 12.0, mov.shape)
    return mov


def save(name, frames, mode, **kw):
    ims = [Image.fromarray(f, mode=mode) if mode else Image.fromarray(f) for f in frames]
    path = os.path.join(HERE, name)
    ims[0].save(path, format="TIFF", save_all=True, append_images=ims[1:], **kw)
    return path


def main():
    mov = movie()
    u16 = np.clip(np.round(mov), 0, 65535).astype(np.uint16)
    u8 = np.clip(np.round(mov / 8.0), 0, 255).astype(np.uint8)
    f32 = mov.astype(np.float32)
    TiffImagePlugin.WRITE_LIBTIFF = True     # libtiff encoder for every file (strip layout as written in the wild)
    written = {}
    written["pillow_u16_raw.tif"] = save("pillow_u16_raw.tif", u16, None, compression="raw")
    written["pillow_u16_lzw.tif"] = save("pillow_u16_lzw.tif", u16, None, compression="tiff_lzw")
    written["pillow_u16_deflate.tif"] = save("pillow_u16_deflate.tif", u16, None, compression="tiff_adobe_deflate")
    written["pillow_u16_packbits.tif"] = save("pillow_u16_packbits.tif", u16, None, compression="packbits")
    written["pillow_u16_lzw_pred.tif"] = save("pillow_u16_lzw_pred.tif", u16, None, compression="tiff_lzw",
                                              tiffinfo={317: 2})
    written["pillow_u8_raw.tif"] = save("pillow_u8_raw.tif", u8, None, compression="raw")
    written["pillow_f32_raw.tif"] = save("pillow_f32_raw.tif", f32, None, compression="raw")
    # pages larger than libtiff's 8 KiB default strip: several strips per page (RowsPerStrip 20: 4 frames of 96 x 128 uint16 in 5 strips each, the last one short)
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:96, 0:128]
    wide = np.stack([np.clip(np.round(500 + 300 * np.sin(yy / 9.0 + t) * np.cos(xx / 13.0) + rng.normal(0, 9, yy.shape)),
                             0, 65535) for t in range(4)]).astype(np.uint16)
    written["pillow_u16_strips_raw.tif"] = save("pillow_u16_strips_raw.tif", wide, None, compression="raw", tiffinfo={278: 20})
    written["pillow_u16_strips_lzw.tif"] = save("pillow_u16_strips_lzw.tif", wide, None, compression="tiff_lzw", tiffinfo={278: 20})
    # a movie long enough to decompose (260 frames >= one Welch segment), deflate-compressed: the end-to-end GPU test
    # decomposes it through TiffArray and from the array stored here
    from localmd_amd.synthetic import make_movie
    long_u16 = np.clip(np.round(make_movie(260, 32, 24, seed=77) * 6.0 + 300.0), 0, 65535).astype(np.uint16)
    written["pillow_u16_movie_deflate.tif"] = save("pillow_u16_movie_deflate.tif", long_u16, None,
                                                   compression="tiff_adobe_deflate")
    np.savez_compressed(os.path.join(HERE, "tiff_fixture_expected.npz"), u16=u16, u8=u8, f32=f32, wide=wide, movie=long_u16)
    for k, p in written.items():
        with Image.open(p) as im:
            print(k, os.path.getsize(p), "bytes; pages", getattr(im, "n_frames", 1), "compression tag", im.tag_v2.get(259),
                  "predictor", im.tag_v2.get(317), "strips", len(im.tag_v2.get(273, ())))


if __name__ == "__main__":
    main()
