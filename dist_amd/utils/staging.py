"""Host-resident input path (reference runs/train.py:81-101 moves every loader batch with `.cuda(non_blocking=True)` inside the step).

The C-ABI boundary takes device pointers, and a 32-clip fp32 batch is 308 MB: copied inside the step it costs ~5 ms of a 16.5 ms step.  The frozen ViT
pass of batch n+1 is the only consumer of its frames and already runs one step ahead (dist_vit_prefetch), so the copy of batch n+2 can run TWO steps
ahead, beside step n, on a copy stream of its own (SDMA engine): by the time step n+1 hands batch n+2 to the prefetch, its frames have been resident
for a whole step and nothing waits.  `HostStager` is that double buffer: pinned staging on the host side (a pageable batch is first copied into a
pinned buffer - pageable memory would make the "asynchronous" copy synchronous), a ring of device buffers, one event per buffer in each direction.
"""
import torch


class HostStager:
    def __init__(self, depth=3, device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("HostStager needs a GPU")
        self.depth = depth
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)          # copies only (H2D through SDMA)
        self.dev = [None] * depth
        self.pin = [None] * depth
        self.ready = [torch.cuda.Event() for _ in range(depth)]       # copy stream -> consumers: buffer i holds its batch
        self.freed = [None] * depth                                  # consumer stream -> copy stream: buffer i may be overwritten
        self.k = 0

    def submit(self, host):
        """Start the copy of `host` (CPU tensor, pinned or pageable) into the next device buffer; returns a ticket for `wait`."""
        assert not host.is_cuda
        i = self.k % self.depth
        self.k += 1
        if self.dev[i] is None or self.dev[i].shape != host.shape or self.dev[i].dtype != host.dtype:
            self.dev[i] = torch.empty(host.shape, dtype=host.dtype, device=self.device)
            self.freed[i] = None
        src = host
        if not host.is_pinned():
            if self.pin[i] is None or self.pin[i].shape != host.shape or self.pin[i].dtype != host.dtype:
                self.pin[i] = torch.empty(host.shape, dtype=host.dtype, pin_memory=True)
            else:
                self.ready[i].synchronize()                          # the previous copy out of this pinned buffer has finished
            self.pin[i].copy_(host)
            src = self.pin[i]
        with torch.cuda.stream(self.stream):
            if self.freed[i] is not None:
                self.stream.wait_event(self.freed[i])
            self.dev[i].copy_(src, non_blocking=True)
            self.ready[i].record(self.stream)
        return i

    def wait(self, ticket):
        """The CURRENT stream waits for the copy of `ticket`; returns the device tensor (valid until `release(ticket)` + depth - 1 further submits)."""
        torch.cuda.current_stream().wait_event(self.ready[ticket])
        return self.dev[ticket]

    def release(self, ticket):
        """Call on the consumer's stream once the last kernel reading buffer `ticket` has been enqueued there."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.freed[ticket] = ev
