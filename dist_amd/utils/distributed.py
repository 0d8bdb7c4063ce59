"""torch.distributed (RCCL over xGMI on ROCm, gloo on CPU) launcher helpers.

Replaces the reference's utils/distributed.py (same function names and argument
meaning: all_reduce(tensors, average=True), all_gather(tensors), is_master_proc,
get_world_size, get_rank, synchronize, init_distributed_training, get_local_size,
get_local_rank; reference utils/distributed.py:19-303) and utils/launcher.py.

Data parallel scheme of the DiST hot path: clips are independent
units sharded per rank; the frozen ViT / text weights are replicated and never
reduced; ONE exchange step per train step: a sum all-reduce over the flat dist_net
gradient buffer (19.0 M fp32 for ViT-B/16 = 76 MB, vs the reference's DDP which
reduces all 168.6 M parameters with find_unused_parameters=True).
"""
import os

import torch
import torch.distributed as dist

_LOCAL_WORLD = 1
_LOCAL_RANK = 0


def init_process_group(rank, world, local_rank=0, backend=None, init_method=None):
    """One process per GPU; backend 'nccl' IS RCCL on ROCm; gloo for the CPU tests."""
    global _LOCAL_WORLD, _LOCAL_RANK
    if backend is None:
        # DIST_AMD_BACKEND=gloo lets two ranks share ONE GPU (RCCL refuses duplicate devices): that is how the
        # multi-rank path (hooks, buckets, streams) is exercised on a 1-GPU box (tests/test_ddp_gpu.py)
        backend = os.environ.get("DIST_AMD_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if init_method is None:
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = os.environ.get("MASTER_PORT", "29500")
        init_method = f"tcp://{addr}:{port}"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", local_rank)
        # RCCL's internal stream stays at default priority (see GradReducer: a high-priority stream beside the engine's streams
        # slows the whole step down)
    try:
        dist.init_process_group(backend=backend, init_method=init_method, rank=rank, world_size=world, **kw)
    except TypeError:
        kw.pop("pg_options", None)
        dist.init_process_group(backend=backend, init_method=init_method, rank=rank, world_size=world, **kw)
    _LOCAL_WORLD = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    _LOCAL_RANK = local_rank


def init_distributed_training(cfg):
    """reference utils/distributed.py:262-277 creates one process group per machine; with a single
    node (the scope of this build) the default group is the local group."""
    return None


def destroy():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_local_size():
    return _LOCAL_WORLD


def get_local_rank():
    return _LOCAL_RANK


def is_master_proc(num_gpus=8):
    """reference utils/distributed.py:98-105: rank % num_gpus == 0 (always true without a group)."""
    return get_rank() % num_gpus == 0 if dist.is_available() and dist.is_initialized() else True


def synchronize():
    barrier()


def barrier():
    if get_world_size() > 1:
        dist.barrier()


def all_reduce(tensors, average=True):
    """reference utils/distributed.py:41-57, but ONE collective for the whole list: the tensors are
    packed into a single buffer (the reference issues one blocking all_reduce per 0-d metric)."""
    world = get_world_size()
    if world == 1:
        return tensors
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    dist.all_reduce(flat)
    if average:
        flat.mul_(1.0 / world)
    out, o = [], 0
    for t in tensors:
        n = t.numel()
        out.append(flat[o:o + n].view(t.shape).to(t.dtype))
        o += n
    return out


def all_reduce_max(t):
    if get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def all_gather(tensors):
    """reference utils/distributed.py:19-38: gather each tensor from all ranks, concatenate on dim 0."""
    world = get_world_size()
    if world == 1:
        return tensors
    out = []
    for t in tensors:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous())
        out.append(torch.cat(parts, dim=0))
    return out


class GradReducer:
    # (see exposed_ms below)
    """Sum all-reduce of the flat dist_net gradient buffer, overlapped with backward.

    The engine reports each gradient slice as soon as the kernels producing it are enqueued (ada-pooling + head,
    then layer L-1 ... 0, then the stem); slices are coalesced into buckets of >= `bucket_bytes` and every bucket's
    all-reduce is issued on a side HIP stream behind an event of the compute stream, so RCCL runs over xGMI while
    the remaining layers' backward kernels execute.  The 1/world average is folded into AdamW (`grad_scale`);
    the 1-element logit_scale gradient stays local (it is never optimised)."""

    def __init__(self, engine, world, bucket_bytes=16 << 20, overlap=True, grad_dtype=None):
        self.eng = engine
        # exchange precision ("fp32 (76.0 / 158.3 MB) or bf16"): fp32 by default; torch.bfloat16 halves the bytes on xGMI - the
        # bucket is rounded into a staging buffer on the communication stream, summed there and added back as fp32
        self.grad_dtype = grad_dtype or (torch.bfloat16 if os.environ.get("DIST_AMD_REDUCER_DTYPE", "") == "bf16" else torch.float32)
        self._timing = []             # (backward enqueued on the compute stream, last bucket reduced on the communication stream) per step
        self.world = world
        self.grad_scale = 1.0 / world
        self.bucket_elems = bucket_bytes // 4
        force = bool(os.environ.get("DIST_AMD_FORCE_REDUCER"))     # measurement knob: the reducer's streams / events at world size 1
        self.force = force
        self.overlap = overlap and (world > 1 or force) and torch.cuda.is_available()
        self.mode = os.environ.get("DIST_AMD_REDUCER_MODE", "")      # measurement knob: noop / hiprio / sync
        if self.mode == "sync":
            self.overlap = False
        # DEFAULT-priority stream.  A high-priority stream beside the engine's four (caller's, two side streams, ViT prefetch) costs
        # 8-9 ms per step on MI355X (22.6 -> 31.3 ms with the reducer's streams / events alone, no collective issued): measured with
        # DIST_AMD_FORCE_REDUCER=1 at world size 1, profiles/r01_streams_and_queues.md
        self.comm = (torch.cuda.Stream(priority=-1) if self.mode == "hiprio" else torch.cuda.Stream()) if self.overlap else None
        self._pending = None          # [begin, end) not yet sent (slices arrive in descending order)
        self._sent = []
        self.n_collectives = 0
        if self.overlap:
            engine.set_grad_ready_hook(self._on_slice)

    def _send(self, begin, end):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.comm.wait_event(ev)
        with torch.cuda.stream(self.comm):
            if self.mode != "noop":
                g = self.eng.grads[begin:end]
                if self.grad_dtype == torch.float32:
                    dist.all_reduce(g)
                else:
                    st = g.to(self.grad_dtype)
                    dist.all_reduce(st)
                    g.copy_(st)
        self.n_collectives += 1
        self._sent.append((begin, end))

    def _on_slice(self, begin, end):
        if self._pending is None:
            self._pending = [begin, end]
        elif end == self._pending[0]:
            self._pending[0] = begin
        else:                          # not adjacent: flush what we have
            self._send(*self._pending)
            self._pending = [begin, end]
        if self._pending[1] - self._pending[0] >= self.bucket_elems:
            self._send(*self._pending)
            self._pending = None

    def exposed_ms(self):
        """Mean time per step the optimizer waited for the exchange AFTER the backward pass had finished on the compute stream (0 when the
        last bucket's all-reduce ends first): the un-hidden part of the RCCL leg.  Synchronises; call it outside the timed region."""
        if not self._timing:
            return None
        torch.cuda.synchronize()
        return sum(max(0.0, a.elapsed_time(b)) for a, b in self._timing) / len(self._timing)

    def backward_and_reduce(self, dlogits):
        self.n_collectives = 0
        self._sent = []
        self.eng.backward(dlogits)
        if self.world <= 1 and not self.force:
            return
        if not self.overlap:
            dist.all_reduce(self.eng.grads)
            self.n_collectives = 1
            return
        if self._pending is not None:
            self._send(*self._pending)
            self._pending = None
        # what AdamW waits for: `exposed_ms` = the part of the exchange that is NOT hidden behind the backward pass (two timed events:
        # backward fully enqueued on the compute stream -> last bucket reduced on the communication stream)
        bwd_done = torch.cuda.Event(enable_timing=True)
        bwd_done.record(torch.cuda.current_stream())
        done = torch.cuda.Event(enable_timing=True)
        done.record(self.comm)
        torch.cuda.current_stream().wait_event(done)      # AdamW waits for the last bucket
        self._timing.append((bwd_done, done))
        if len(self._timing) > 64:
            self._timing.pop(0)
        covered = sum(e - b for b, e in self._sent)
        if covered != self.eng.grads.numel():              # must never happen: every element is reduced exactly once
            raise RuntimeError(f"gradient buckets cover {covered} of {self.eng.grads.numel()} elements")
