"""tile_atx on the config-3 shape (2601 tiles, d = 400, T = 10000): the four-row-tile kernel against the row-hinted forms
(GPU box):  python scripts/atx_rows_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from localmd_amd._lib import Context, ptr as P
ctx = Context(0)
lib = ctx.lib
n_tiles, d, T = 2601, 400, 10000
ld, dpad = lib.pmd_time_ld(T), lib.pmd_tile_dpad(d)
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn((262144, ld), device="cuda", generator=g)
A = torch.zeros((n_tiles, 64, dpad), device="cuda")
A[:, :50, :d] = torch.randn((n_tiles, 50, d), device="cuda", generator=g)
pix = torch.stack([torch.randperm(262144, device="cuda", generator=g)[:d] for _ in range(64)]).repeat(41, 1)[:n_tiles].to(torch.int32).contiguous()
Out = torch.empty((n_tiles, 64, ld), device="cuda")
def run(rows):
    if rows:
        ctx.call("pmdk_tile_atx_rows", P(X), ld, P(pix), d, d, d, P(A), 64 * dpad, dpad, P(Out), 64 * ld, ld, n_tiles, T, 2, rows)
    else:
        ctx.call("pmdk_tile_atx", P(X), ld, P(pix), d, d, d, P(A), 64 * dpad, dpad, P(Out), 64 * ld, ld, n_tiles, T, 2)
for rows in (0, 50, 48, 0, 50):
    run(rows); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run(rows)
    e1.record(); torch.cuda.synchronize()
    print(f"rows hint {rows:2d}: {e0.elapsed_time(e1) / 5:.2f} ms per launch", flush=True)
