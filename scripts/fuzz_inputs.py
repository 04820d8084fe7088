"""The same movie handed over in every container / layout / dtype the host driver accepts must give bit-identical results
(after the same rounding for the integer types).    python scripts/fuzz_inputs.py SEED"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd.dataset import ArrayDataset, TiffArray
from localmd_amd._minitiff import write_tiff
from localmd_amd.synthetic import make_movie
from localmd_amd._lib import Context

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
ctx = Context(0)
Dm.QUIET = True
bad = 0
for trial in range(6):
    T = int(rng.integers(90, 400)); d1 = int(rng.integers(20, 50)); d2 = int(rng.integers(20, 50))
    b1, b2 = int(2 * rng.integers(5, 1 + min(10, d1 // 2))), int(2 * rng.integers(5, 1 + min(10, d2 // 2)))
    fbs = int(rng.choice([10000, T - 1, 64, 37, T // 2 + 1]))
    kw = dict(max_components=int(rng.integers(2, 8)), background_rank=int(rng.integers(0, 4)), thresholds=(1.0, 1.7), frame_batch_size=fbs,
              temporal_avg_factor=int(rng.choice([1, 2, 5])), order=str(rng.choice(["F", "C"])))
    base = np.round(make_movie(T, d1, d2, seed=50 + trial) * 8.0).astype(np.float32)      # integer-valued: every dtype holds it exactly

    def run(src):
        np.random.seed(3)
        return localmd_amd.localmd_decomposition(src, (b1, b2), T, seed=7, ctx=ctx, **kw)

    ref = run(base)
    big = np.zeros((T, d1 + 3, d2 + 5), np.float32); big[:, 1:1 + d1, 2:2 + d2] = base
    tdir = tempfile.mkdtemp()
    tif16 = os.path.join(tdir, "m16.tif"); write_tiff(tif16, base.astype(np.uint16))
    tif32 = os.path.join(tdir, "m32.tif"); write_tiff(tif32, base)
    variants = {
        "fortran order": np.asfortranarray(base), "non-contiguous view": big[:, 1:1 + d1, 2:2 + d2], "time-strided storage": np.ascontiguousarray(base.transpose(1, 2, 0)).transpose(2, 0, 1),
        "float64": base.astype(np.float64), "uint16": base.astype(np.uint16), "int32": base.astype(np.int32), "int16": base.astype(np.int16),
        "torch cpu": torch.from_numpy(base.copy()), "torch cuda": torch.from_numpy(base.copy()).to(ctx.device), "torch cuda uint16->int32": torch.from_numpy(base.astype(np.int32)).to(ctx.device),
        "ArrayDataset": ArrayDataset(base), "ArrayDataset uint16": ArrayDataset(base.astype(np.uint16)), "TiffArray uint16": TiffArray(tif16), "TiffArray float32": TiffArray(tif32),
    }
    for name, src in variants.items():
        try:
            out = run(src)
        except Exception as e:        # noqa: BLE001
            bad += 1
            print(f"trial {trial} ({T}x{d1}x{d2}, batch {fbs}) {name}: {type(e).__name__}: {str(e)[:200]}")
            continue
        same = (np.array_equal(out.s, ref.s) and np.array_equal(out.r, ref.r) and np.array_equal(out.v, ref.v) and np.array_equal(out.u.data, ref.u.data)
                and np.array_equal(out.u.indices, ref.u.indices) and np.array_equal(out.mean_img, ref.mean_img) and np.array_equal(out.var_img, ref.var_img))
        if not same:
            bad += 1
            print(f"trial {trial} ({T}x{d1}x{d2}, batch {fbs}) {name}: differs; s {np.abs(out.s - ref.s).max() if out.s.shape == ref.s.shape else 'shape'}, mean {np.abs(out.mean_img - ref.mean_img).max()}")
print(f"input fuzz seed {seed}: {bad} disagreements")
