import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.ones(1000, device="cuda")
dist.all_reduce(x); dist.barrier()
box = [{"a": 1}]
dist.broadcast_object_list(box, src=0)
t = torch.arange(12., device="cuda").reshape(4, 3)
dist.broadcast(t[1:3], src=0)
print("backend", dist.get_backend(), "ok", float(x.sum()), box)
import sys; sys.path.insert(0, "/root/repo")
from localmd_amd.parallel import Dist
d = Dist(True)
print("Dist enabled", d.enabled, d.rank, d.world)
dist.destroy_process_group()
