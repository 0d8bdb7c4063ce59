"""Two (or more) ranks through the data-parallel step: bucketed, overlapped gradient all-reduce + fused AdamW.
Launched by tests/test_ddp_gpu.py with torch.distributed.run; with DIST_AMD_BACKEND=gloo all ranks share GPU 0.
Checks on rank 0: reduced gradients == sum of the per-rank gradients computed locally by a second engine;
parameters after the optimizer step are identical on every rank."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from dist_amd import synth
from dist_amd.utils import distributed as du
from dist_amd.engine import Engine, config_from_geometry


def main():
    world, rank, local_rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    du.init_process_group(rank, world, local_rank)
    gname, b = sys.argv[1], int(sys.argv[2])
    dtype = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == "fp32" else torch.bfloat16
    g = synth.geometry(gname)
    sd = synth.state_dict(g)
    text = torch.from_numpy(synth.text_features(g)).cuda()

    def data(r):
        return (torch.from_numpy(synth.video(g, b, seed=1 + r)).cuda(), torch.from_numpy(synth.soft_target(g, b, seed=3 + r)[0]).cuda())

    eng = Engine(config_from_geometry(g, b, dtype)); eng.load_state_dict(sd)
    red = du.GradReducer(eng, world, bucket_bytes=1 << 16)      # small buckets: several collectives even for the tiny model
    for it in range(2):
        video, tgt = data(rank)
        eng.vit_forward(video); eng.branch_forward(text)
        _, dl = eng.loss(tgt)
        red.backward_and_reduce(dl)
        torch.cuda.synchronize()
        reduced = eng.grads.clone()
        ncoll = red.n_collectives
        eng.adamw_step(1e-3, 1e-4, lr_mult=10.0, grad_scale=1.0 / world)
        torch.cuda.synchronize()
        if it == 0:
            first_reduced, first_ncoll = reduced, ncoll
    # every rank holds the same parameters
    chk = torch.stack([eng.theta.double().sum(), eng.theta.double().abs().sum()]).cuda()
    parts = [torch.empty_like(chk) for _ in range(world)]
    dist.all_gather(parts, chk)
    same = all(torch.equal(parts[0], p) for p in parts)
    ok = True
    if rank == 0:
        ref = Engine(config_from_geometry(g, b, dtype)); ref.load_state_dict(sd)
        total = torch.zeros_like(ref.grads)
        for r in range(world):
            video, tgt = data(r)
            ref.vit_forward(video); ref.branch_forward(text)
            _, dl = ref.loss(tgt); ref.backward(dl)
            torch.cuda.synchronize()
            total += ref.grads
        err = (first_reduced - total).abs().max().item()
        scale = total.abs().max().item()
        ok = same and err <= 1e-5 * scale + 1e-7 and first_ncoll >= 2
        print(f"DDP_CHECK world={world} collectives={first_ncoll} grad_err={err:.3e} grad_max={scale:.3e} params_identical={same} -> {'OK' if ok else 'FAIL'}", flush=True)
    du.barrier()
    du.destroy()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
