"""Two ranks on one GPU (gloo) vs one rank: the decomposition must not depend on the rank count."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import localmd_amd
from localmd_amd import decomposition as Dm
from localmd_amd.synthetic import make_movie
Dm.QUIET = True
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
rank = dist.get_rank()
mov = make_movie(400, 50, 46, seed=5)
np.random.seed(3)
a, da = localmd_amd.localmd_decomposition(mov, (20, 20), 400, max_components=6, background_rank=2, seed=4, sim_iters=8,
                                          distributed=True, return_diagnostics=True)
np.random.seed(3)
b, db = localmd_amd.localmd_decomposition(mov, (20, 20), 400, max_components=6, background_rank=2, seed=4, sim_iters=8,
                                          distributed=False, return_diagnostics=True)
ok = (np.array_equal(da["tile_ranks"], db["tile_ranks"]) and np.array_equal(a.u.indices, b.u.indices)
      and np.array_equal(a.u.data, b.u.data) and np.allclose(a.s, b.s, rtol=1e-6) and np.allclose(np.abs(a.v), np.abs(b.v), atol=1e-4))
print(f"rank {rank}: distributed == single: {ok}; ranks {da['tile_ranks'].tolist()}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
