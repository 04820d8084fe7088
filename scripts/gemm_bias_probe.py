"""Is a long fp32 contraction on the matrix cores biased?  diag(V V^T) with K = 10^4 .. 5.6e4 positive terms: rocBLAS sgemm
(through pmd_gemm and through torch.matmul), split-K in chunks, against float64."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from localmd_amd._lib import Context, ptr
ctx = Context(0)
torch.manual_seed(0)
for n, K in ((512, 10000), (512, 55804)):
    V = torch.randn((n, K), dtype=torch.float32, device="cuda")
    ref = (V.double() @ V.double().T)
    dref = ref.diagonal()
    def rep(name, G):
        G = G.double()
        d = (G.diagonal() / dref - 1)
        off = (G - ref).abs().max().item() / dref.mean().item()
        print(f"n={n} K={K} {name:28s} diag rel err: mean {d.mean().item():+.2e} max|.| {d.abs().max().item():.2e}; max |offdiag err| / mean diag {off:.2e}")
    rep("torch.matmul fp32", V @ V.T)
    G = torch.empty((n, n), dtype=torch.float32, device="cuda")
    ctx.call("pmd_gemm", 0, 1, n, n, K, 1.0, ptr(V), K, ptr(V), K, 0.0, ptr(G), n)
    ctx.sync()
    rep("pmd_gemm (rocBLAS sgemm)", G)
    for ch in (4096, 1024, 256):
        acc = torch.zeros((n, n), dtype=torch.float64, device="cuda")
        for k0 in range(0, K, ch):
            acc += (V[:, k0:k0 + ch] @ V[:, k0:k0 + ch].T).double()
        rep(f"chunks of {ch} summed in fp64", acc)
