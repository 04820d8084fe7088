"""Per-stage times of the two-stage eigensolver (library profiler) at one order.  usage: ts_profile.py n [mode]"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from localmd_amd._lib import Context, ptr
n = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else "twostage"
os.environ["PMD_SYEVD"] = mode
ctx = Context(0)
g = torch.Generator(device="cpu"); g.manual_seed(n)
X = torch.randn((n, n + 50), generator=g).to(ctx.device) * torch.linspace(1, 30, n + 50, device=ctx.device)[None, :]
A0 = X @ X.T
lda = (n + 3) // 4 * 4
for rep in range(3):
    A = torch.zeros((n, lda), dtype=torch.float32, device=ctx.device); A[:, :n] = A0
    w = torch.zeros(n, dtype=torch.float32, device=ctx.device); work = torch.zeros(n, dtype=torch.float32, device=ctx.device)
    info = torch.zeros(4, dtype=torch.int32, device=ctx.device)
    if rep == 2: ctx.profile_enable(True)
    ctx.sync(); t0 = time.perf_counter()
    ctx.call("pmdk_syevd", n, ptr(A), lda, ptr(w), ptr(work), ptr(info))
    ctx.sync(); dt = time.perf_counter() - t0
    print(f"n = {n} {mode} call {rep}: {dt*1e3:.1f} ms", flush=True)
for k, (ms, cnt) in sorted(ctx.profile_summary().items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:28s} {ms:9.2f} ms  {cnt:6d}")
