"""Host array -> HBM ingestion rate of _Movie._stream_in for several staging-buffer sizes / thread counts (GPU box)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from localmd_amd._lib import Context
from localmd_amd.decomposition import _Movie

ctx = Context(0)
T, d1, d2 = 10000, 512, 512
mov = np.empty((T, d1, d2), dtype=np.float32)
mov[:] = 1.0   # touch every page
print("cpus", os.cpu_count(), flush=True)
for stage_mb, bufs, threads in ((64, 3, 32), (256, 3, 32), (256, 4, 64), (512, 4, 64), (512, 4, 96), (1024, 3, 64), (256, 4, 16)):
    _Movie.STAGE_BYTES = stage_mb << 20
    _Movie.STAGE_BUFFERS = bufs
    _Movie._stage_cache.clear()
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = _Movie(ctx, mov, 10000, num_workers=threads)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep > 0:
            best = min(best, dt)
        del m
    print(f"stage {stage_mb} MB x {bufs}, {threads} threads: {best:.3f} s = {mov.nbytes / best / 1e9:.1f} GB/s", flush=True)
