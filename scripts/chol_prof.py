"""Cholesky route of the orthogonaliser alone (for rocprofv3 --kernel-trace --stats)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localmd_amd import _lib  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 9999
Rc = m + 512
ctx = _lib.Context(0)
P = _lib.ptr
g = torch.Generator(device="cuda").manual_seed(0)
M = torch.randn((Rc, m), device="cuda", generator=g)
Et = torch.empty((m, m), device="cuda")
ok = C.c_int(0)
ws = ctx.workspace(ctx.lib.pmd_orthogonalize_chol_workspace_bytes(Rc, m))
for rep in range(2):
    ctx.profile_enable(True)
    ctx.call("pmd_orthogonalize_chol", P(M), Rc, m, m, P(M), m, P(Et), m, C.byref(ok), P(ws), ws.numel())
    ctx.sync()
    print(ok.value, {k: round(v[0], 2) for k, v in ctx.profile_summary().items()})
    ctx.profile_enable(False)
