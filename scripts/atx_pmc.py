import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from localmd_amd._lib import Context, ptr
from localmd_amd import grid
ctx = Context(0); lib = ctx.lib
T, d1, d2, b = 10000, 512, 512, 20
D = d1 * d2; ld = lib.pmd_time_ld(T)
X = torch.randn((D, ld), dtype=torch.float32, device=ctx.device)
it1, it2 = grid.tile_origins((d1, d2), (b, b)); pix, _ = grid.tile_pixel_lists((d1, d2), (b, b), it1, it2)
n, d = pix.shape; dpad = lib.pmd_tile_dpad(d)
A = torch.randn((n, 64, dpad), dtype=torch.float32, device=ctx.device)
Out = torch.empty((n, 64, ld), dtype=torch.float32, device=ctx.device)
p = torch.from_numpy(pix).to(ctx.device)
for _ in range(3):
    ctx.call("pmdk_tile_atx", ptr(X), ld, ptr(p), d, 0, d, ptr(A), 64 * dpad, dpad, ptr(Out), 64 * ld, ld, n, T, 4)
torch.cuda.synchronize()
