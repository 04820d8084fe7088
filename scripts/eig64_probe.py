"""Cost of a float64 eigendecomposition (torch -> hipSOLVER/rocSOLVER dsyevd) next to the library's fp32 pmdk_syevd, small orders."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from localmd_amd._lib import Context, ptr
ctx = Context(0)
for n in (40, 72, 128, 191, 300, 512, 768, 1024, 1536, 2048):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, n + 20)).astype(np.float32)
    A = torch.from_numpy(X @ X.T).to(ctx.device)
    ld = (n + 3) // 4 * 4
    def own():
        B = torch.zeros((n, ld), dtype=torch.float32, device=ctx.device); B[:, :n] = A
        w = torch.zeros(n, dtype=torch.float32, device=ctx.device); work = torch.zeros(n, dtype=torch.float32, device=ctx.device)
        info = torch.zeros(4, dtype=torch.int32, device=ctx.device)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.call("pmdk_syevd", n, ptr(B), ld, ptr(w), ptr(work), ptr(info)); ctx.sync()
        return time.perf_counter() - t0
    def t64():
        B = A.double()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        w, v = torch.linalg.eigh(B); torch.cuda.synchronize()
        return time.perf_counter() - t0
    def t32():
        B = A.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        w, v = torch.linalg.eigh(B); torch.cuda.synchronize()
        return time.perf_counter() - t0
    for f in (own, t64, t32): f()
    print(f"n = {n:5d}: pmdk_syevd {min(own() for _ in range(3))*1e3:7.2f} ms   torch eigh f64 {min(t64() for _ in range(3))*1e3:7.2f} ms   torch eigh f32 {min(t32() for _ in range(3))*1e3:7.2f} ms", flush=True)
