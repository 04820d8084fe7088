"""Device -> host copy rates: pinned destination (fresh / reused), sliced rows, side stream (GPU box)."""
import time, torch
dev = torch.device("cuda", 0)
x = torch.randn((309842, 1000), device=dev)
for rep in range(3):
    t0 = time.perf_counter(); h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True); t1 = time.perf_counter()
    h.copy_(x, non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rep {rep}: pinned alloc {1e3*(t1-t0):.1f} ms (is_pinned {h.is_pinned()}), copy {1e3*(t2-t1):.1f} ms = {x.numel()*4/(t2-t1)/1e9:.1f} GB/s", flush=True)
    del h
side = torch.cuda.Stream(device=dev)
h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
torch.cuda.synchronize(); t1 = time.perf_counter()
with torch.cuda.stream(side):
    h[100:309000].copy_(x[100:309000], non_blocking=True)
side.synchronize(); t2 = time.perf_counter()
print(f"row slice on a side stream: {1e3*(t2-t1):.1f} ms", flush=True)
hp = torch.empty(x.shape, dtype=x.dtype)
torch.cuda.synchronize(); t1 = time.perf_counter(); hp.copy_(x); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"pageable destination: {1e3*(t2-t1):.1f} ms", flush=True)
