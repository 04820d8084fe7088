"""Host -> HBM staging rate of localmd_amd.decomposition._Movie._stream_in for a pageable NumPy movie:
staging buffer size x worker threads (second call of each setting = page-locked ring already cached)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from localmd_amd._lib import Context
from localmd_amd.decomposition import _Movie

ctx = Context(0)
T, d1, d2 = 5000, 512, 512
mov = np.random.default_rng(0).standard_normal((T, d1, d2), dtype=np.float32)
gb = mov.nbytes / 1e9
for stage_mb in (64, 128, 256):
    for threads in (8, 16, 32, 64):
        _Movie.STAGE_BYTES = stage_mb << 20
        _Movie._stage_cache.clear()
        times = []
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m = _Movie(ctx, mov, 10000, num_workers=threads)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            del m
        print(f"stage {stage_mb} MB, {threads} threads: first {gb / times[0]:.1f} GB/s, cached {gb / min(times[1:]):.1f} GB/s", flush=True)
