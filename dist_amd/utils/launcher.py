"""One process per GPU (reference utils/launcher.py:9-83): `launch_task(cfg, init_method, func)` spawns
NUM_GPUS workers, each initialises torch.distributed (backend nccl = RCCL on ROCm) with a tcp init method
on 127.0.0.1 and binds its GPU.  Under `torchrun` (RANK / WORLD_SIZE in the environment) the current
process is a worker already."""
import os

import torch

from . import distributed as du


def run(local_rank, num_proc, func, init_method, shard_id, num_shards, backend, cfg):
    world = num_proc * num_shards
    rank = shard_id * num_proc + local_rank
    if world > 1:
        du.init_process_group(rank, world, local_rank, backend=backend, init_method=init_method)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    try:
        func(cfg)
    finally:
        du.destroy()


def launch_task(cfg, init_method, func):
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:          # torchrun already spawned us
        lr = int(os.environ.get("LOCAL_RANK", 0))
        world = int(os.environ["WORLD_SIZE"])
        if world > 1:
            du.init_process_group(int(os.environ["RANK"]), world, lr, backend=cfg.DIST_BACKEND)
        torch.cuda.set_device(lr)
        try:
            func(cfg)
        finally:
            du.destroy()
        return
    n = int(cfg.NUM_GPUS)
    if n > 1:
        init_method = init_method.replace("localhost", "127.0.0.1")
        torch.multiprocessing.spawn(run, nprocs=n, args=(n, func, init_method, cfg.SHARD_ID, cfg.NUM_SHARDS, cfg.DIST_BACKEND, cfg), daemon=False)
    else:
        func(cfg=cfg)
