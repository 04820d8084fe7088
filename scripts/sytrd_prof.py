"""One tridiagonalisation with the library's own kernels (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localmd_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
impl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = _lib.Context(0)
P = _lib.ptr
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn((n, n + 2000), device="cuda", generator=g)
S = (X @ X.T).contiguous()
del X
d = torch.zeros(n, device="cuda"); e = torch.zeros(n, device="cuda"); tau = torch.zeros(n, device="cuda")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for rep in range(reps):
    A = S.clone()
    ctx.call("pmdk_sytrd", n, P(A), n, P(d), P(e), P(tau), impl)
    ctx.sync()
print("done", d[:2].tolist())
