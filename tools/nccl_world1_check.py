import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
from dist_amd import distributed as du
torch.cuda.set_device(0)
du.init_process_group(0, 1, 0)
import torch.distributed as dist
t = torch.ones(1 << 20, device="cuda")
s = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(s):
    dist.all_reduce(t)
torch.cuda.synchronize()
du.barrier()
print("nccl world-1 ok", float(t.sum()), dist.get_backend())
du.destroy()
