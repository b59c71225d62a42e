import os, sys, torch, torch.distributed as dist
sys.path.insert(0, "music-generation-emotion-adaptive_amd")
from mgea import dist as mdist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
a = torch.arange(1 << 20, dtype=torch.float32, device="cuda:0")
dist.broadcast(a, src=0)
t = torch.tensor([3.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
ids = torch.ones(4, 8, dtype=torch.int32, device="cuda:0")
out = [torch.empty_like(ids)]
dist.all_gather(out, ids)
dist.barrier()
torch.cuda.synchronize()
print("rccl ok", float(a.sum()), float(t), int(out[0].sum()), dist.get_backend())
dist.destroy_process_group()
