import os, torch, torch.distributed as dist, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from seed_amd.dist import gather_token_ids
ids = torch.arange(256 * 32, dtype=torch.int64, device="cuda").view(256, 32) % 8192
out = gather_token_ids(ids)
torch.cuda.synchronize()
print("gathered", tuple(out.shape), out.dtype, bool(torch.equal(out, ids)))
dist.barrier(); dist.destroy_process_group()
