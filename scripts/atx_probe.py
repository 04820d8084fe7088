"""Micro-benchmark of tile_atx: real gather vs. an L2-resident gather (all tiles read the same rows)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from localmd_amd._lib import Context, ptr
from localmd_amd import grid
ctx = Context(0)
lib = ctx.lib
T, d1, d2, b = 10000, 512, 512, 20
D = d1 * d2
ld = lib.pmd_time_ld(T)
X = torch.randn((D, ld), dtype=torch.float32, device=ctx.device)
it1, it2 = grid.tile_origins((d1, d2), (b, b))
pix, _ = grid.tile_pixel_lists((d1, d2), (b, b), it1, it2)
n, d = pix.shape
dpad = lib.pmd_tile_dpad(d)
A = torch.randn((n, 64, dpad), dtype=torch.float32, device=ctx.device)
Out = torch.empty((n, 64, ld), dtype=torch.float32, device=ctx.device)
pix_real = torch.from_numpy(pix).to(ctx.device)
pix_same = torch.from_numpy(np.tile(pix[:1], (n, 1))).to(ctx.device)
def run(p, slices, label):
    for _ in range(2):
        ctx.call("pmdk_tile_atx", ptr(X), ld, ptr(p), d, 0, d, ptr(A), 64 * dpad, dpad, ptr(Out), 64 * ld, ld, n, T, slices)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ctx.call("pmdk_tile_atx", ptr(X), ld, ptr(p), d, 0, d, ptr(A), 64 * dpad, dpad, ptr(Out), 64 * ld, ld, n, T, slices)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{label}: {ms:.2f} ms  -> {2*50*d*T*n/ms/1e9:.1f} TF/s algorithmic (r=50), {2*64*d*T*n/ms/1e9:.1f} TF/s issued", flush=True)
for s in (1, 2, 4, 8):
    run(pix_real, s, f"real gather, slices={s}")
run(pix_same, 2, "same rows for every tile (L2 resident), slices=2")
