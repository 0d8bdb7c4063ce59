"""Host-resident input path (reference runs/train.py:81-101 moves every loader batch with `.cuda(non_blocking=True)` inside the step).

The C-ABI boundary takes device pointers, and a 32-clip fp32 batch is 308 MB: copied inside the step it costs ~5 ms of a 16.5 ms step.  The frozen ViT
pass of batch n+1 is the only consumer of its frames and already runs one step ahead (dist_vit_prefetch), so the copy of batch n+2 can run TWO steps
ahead, beside step n: by the time step n+1 hands batch n+2 to the prefetch, its frames have been resident for a whole step and nothing waits.
`HostStager` is that double buffer: pinned staging on the host side (a pageable batch is first copied into a pinned buffer - pageable memory would make
the "asynchronous" copy synchronous), a ring of device buffers, one event per buffer in each direction.

What sits on the copy stream matters on this system (profiles/r06_host_input.md, tools/host_input_probe.py).  Copies alone - SDMA work - cost the step 0.2-0.4 ms.
An EVENT recorded on the copy stream, or waited for by it, ties the SDMA copies to compute-queue signals (barrier packets, probably the copy itself as a blit kernel) on a
queue beside the engine's four, and that costs 1.0 ms (record only) to 2.6 ms (record + wait) of a 16.4 ms step; putting the copies on the stream that carries the frozen-ViT prefetch
serialises 11 ms of ViT with 5.4 ms of copy (21 ms).  So the default is HOST-ordered: the host waits for `freed` before it issues a copy and for the copy stream
to drain before it hands a buffer out - both waits find their work finished a step ago - and the copy stream never sees an event: 16.9 against 16.5 ms resident.
"""
import torch


class HostStager:
    def __init__(self, depth=3, device="cuda", stream=None, host_ordered=True):
        if not torch.cuda.is_available():
            raise RuntimeError("HostStager needs a GPU")
        self.depth = depth
        self.device = torch.device(device)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)     # the stream that carries the H2D copies
        self.dev = [None] * depth
        self.pin = [None] * depth
        self.ready = [torch.cuda.Event() for _ in range(depth)]       # copy stream -> consumers: buffer i holds its batch
        self.freed = [None] * depth                                  # consumer stream -> copy stream: buffer i may be overwritten
        self.k = 0
        # host_ordered: the copy stream never sees an event (no record on it, no wait on it): the HOST waits for `freed` before it issues a copy and for the copy
        # stream to drain before it hands a buffer out.  Both waits find their work finished a step ago, so they cost nothing - while an event recorded on / waited
        # for by the copy stream ties the copies to compute-queue signals and costs the step 1-2.6 ms on this system
        # (profiles/r06_host_input.md: 17.4 / 18.9 ms with events against 16.3 resident).
        self.host_ordered = host_ordered

    def submit(self, host, pinned=None):
        """Start the copy of `host` (CPU tensor, pinned or pageable) into the next device buffer; returns a ticket for `wait`.
        `pinned=True`: the caller knows the tensor is pinned (Tensor.is_pinned() is a driver query per call)."""
        assert not host.is_cuda
        i = self.k % self.depth
        self.k += 1
        if self.dev[i] is None or self.dev[i].shape != host.shape or self.dev[i].dtype != host.dtype:
            self.dev[i] = torch.empty(host.shape, dtype=host.dtype, device=self.device)
            self.freed[i] = None
        src = host
        if not (pinned if pinned is not None else host.is_pinned()):
            if self.pin[i] is None or self.pin[i].shape != host.shape or self.pin[i].dtype != host.dtype:
                self.pin[i] = torch.empty(host.shape, dtype=host.dtype, pin_memory=True)
            elif self.host_ordered:
                self.stream.synchronize()
            else:
                self.ready[i].synchronize()                          # the previous copy out of this pinned buffer has finished
            self.pin[i].copy_(host)
            src = self.pin[i]
        if self.host_ordered:
            if self.freed[i] is not None:
                self.freed[i].synchronize()                          # (recorded on the consumer's stream two or three steps ago)
            with torch.cuda.stream(self.stream):
                self.dev[i].copy_(src, non_blocking=True)
            return i
        with torch.cuda.stream(self.stream):
            if self.freed[i] is not None:
                self.stream.wait_event(self.freed[i])
            self.dev[i].copy_(src, non_blocking=True)
            self.ready[i].record(self.stream)
        return i

    def wait(self, ticket):
        """The CURRENT stream waits for the copy of `ticket`; returns the device tensor (valid until `release(ticket)` + depth - 1 further submits)."""
        if self.host_ordered:
            self.stream.synchronize()                                # call BEFORE submitting the next batch: every copy issued so far is then at least a step old
        else:
            torch.cuda.current_stream().wait_event(self.ready[ticket])
        return self.dev[ticket]

    def release(self, ticket):
        """Call on the consumer's stream once the last kernel reading buffer `ticket` has been enqueued there."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.freed[ticket] = ev
