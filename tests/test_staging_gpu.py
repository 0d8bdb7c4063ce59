"""dist_amd/utils/staging.py: the host -> device double buffer in front of the C ABI (reference runs/train.py:81-101 copies every loader batch inside the step)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("host_ordered", [True, False])
@pytest.mark.parametrize("pinned", [True, False])
def test_host_stager_delivers_every_batch_once_and_never_overwrites_a_live_buffer(gpu_lib, host_ordered, pinned):
    """20 batches through a ring of 3 device buffers with the consumer two batches behind the producer (the train loop's order: wait for batch n+1, submit
    batch n+2, release batch n after its last reader was enqueued): every consumer sees exactly its batch - a long kernel on the consumer's stream keeps
    each buffer alive while later copies are queued, so a copy that did not honour `freed` would corrupt it."""
    from dist_amd.utils.staging import HostStager
    st = HostStager(depth=3, host_ordered=host_ordered)
    n, shape = 20, (4, 3, 8, 32, 32)
    hosts = [torch.full(shape, float(i)) + torch.arange(shape[-1]).float() * 1e-3 for i in range(n)]
    if pinned:
        hosts = [h.pin_memory() for h in hosts]
    busy = torch.randn(2048, 2048, device="cuda")
    tickets, sums = {}, []
    tickets[0] = st.submit(hosts[0], pinned=pinned)
    tickets[1] = st.submit(hosts[1], pinned=pinned)
    for i in range(n):
        dev = st.wait(tickets[i])                          # (host-ordered: before the next submit)
        if i + 2 < n:
            tickets[i + 2] = st.submit(hosts[i + 2], pinned=pinned)
        for _ in range(3):
            busy = busy @ busy * 1e-3                      # the consumer's stream is busy: the read below runs well after later copies were queued
        sums.append((dev.double().sum(), dev[0, 0, 0, 0, :4].clone()))
        st.release(tickets.pop(i))
    torch.cuda.synchronize()
    for i, (s, head) in enumerate(sums):
        assert float(s) == pytest.approx(float(hosts[i].double().sum()), rel=1e-12), i
        assert torch.equal(head.cpu(), hosts[i][0, 0, 0, 0, :4]), i


def test_host_stager_refuses_device_tensors(gpu_lib):
    from dist_amd.utils.staging import HostStager
    with pytest.raises(AssertionError):
        HostStager().submit(torch.zeros(4, device="cuda"))
