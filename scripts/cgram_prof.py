"""Times pmd_gram_mtgm (C = M^T GM, lower block triangle) at config-3 size for several block counts."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localmd_amd import _lib  # noqa: E402

Rc, m = 55694, 9999
ctx = _lib.Context(0)
P = _lib.ptr
g = torch.Generator(device="cuda").manual_seed(0)
M = torch.randn((Rc, 10000), device="cuda", generator=g)
GM = torch.randn((Rc, 10000), device="cuda", generator=g)
Cm = torch.empty((10000, 10000), device="cuda")
ws = ctx.workspace(ctx.lib.pmd_gram_mtgm_workspace_bytes(Rc, m))
for nb in (8, 2, 3, 4, 5, 6, 8, 12, 16):
    os.environ["PMD_C_BLOCKS"] = str(nb)
    ctx.call("pmd_gram_mtgm", P(M), Rc, m, 10000, P(GM), 10000, P(Cm), 10000, P(ws), ws.numel())
    ctx.sync()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(2):
        ctx.call("pmd_gram_mtgm", P(M), Rc, m, 10000, P(GM), 10000, P(Cm), 10000, P(ws), ws.numel())
    b.record()
    torch.cuda.synchronize()
    print("blocks", nb, "%.1f ms" % (a.elapsed_time(b) / 2), flush=True)
