"""Lower bound for stage 1 of a two-stage tridiagonalisation (dense -> band, b = 64) at n = 10^4 on this GPU: only the
two GEMM-shaped steps of every panel through rocBLAS (X = A22 V: n' x n' x 64; A22 -= [V W][W V]^T: n' x n' x 128) - no
panel factorisation, no small products.  DESIGN section 7 quotes the result."""
import sys, time
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
b = 64
dev = torch.device("cuda")
A = torch.randn((n, n), device=dev)
A = (A + A.T) * 0.5
V = torch.randn((n, b), device=dev)
VW = torch.randn((n, 2 * b), device=dev)
WV = torch.randn((n, 2 * b), device=dev)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_x = t_u = 0.0
    e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for j0 in range(0, n - b, b):
        m = n - j0 - b
        A22 = A[j0 + b:, j0 + b:]
        e0.record()
        X = A22 @ V[:m]
        e1.record()
        A22.addmm_(VW[:m], WV[:m].T, beta=1.0, alpha=-1.0)
        e2.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"n = {n}, b = {b}: {len(range(0, n - b, b))} panels, the two GEMM steps alone {dt * 1e3:.1f} ms "
          f"(flops {sum(2.0 * (n - j - b) ** 2 * 3 * b for j in range(0, n - b, b)) / 1e12:.2f} T)", flush=True)
