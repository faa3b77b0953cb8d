"""Host -> device staging for the hot path (first piece).

The reference feeds the GPU from ``DataLoader`` worker processes that hand over pageable CPU
tensors (``tests/profilers/profile_loudness.py:41-43``, ``core/util.py:426-479`` prepare_batch).
On MI355X the kernels consume a 1.8 GB batch in under 3 ms while PCIe moves it in ~32 ms, so the
copy must at least never serialise with the compute: :class:`DeviceStager` copies batch ``k+1``
from PINNED host memory on a side stream into one of ``depth`` device buffers while batch ``k`` is
being processed on the caller's stream, with events in both directions (buffer filled / buffer
free again).  Measured with the north-star step: 56 GB/s over PCIe, 159 k audio-seconds/sec
(``tools/h2dbench.py``).  On CPU devices it is a pass-through.
"""
import torch


class DeviceStager:
    """Iterate device-resident copies of host batches, overlapping H2D with the consumer's work.

    ``batches``: iterable of CPU float tensors of one shape (pinned here if they are not already).
    The yielded tensor is only valid until the next iteration (its buffer is recycled)."""

    def __init__(self, batches, device, depth: int = 2):
        self.batches = batches
        self.device = torch.device(device)
        self.depth = max(2, int(depth))

    def __iter__(self):
        if self.device.type != "cuda":
            for b in self.batches:
                yield b.to(self.device)
            return
        dev = self.device
        copy_stream = torch.cuda.Stream(device=dev)
        bufs, filled, free = [], [], []
        it = iter(self.batches)
        pending = []  # (slot) in flight, in order

        def issue(slot, host):
            if not host.is_pinned():
                host = host.pin_memory()
            if len(bufs) <= slot:
                bufs.append(torch.empty(host.shape, dtype=host.dtype, device=dev))
                filled.append(torch.cuda.Event())
                free.append(torch.cuda.Event())
                free[slot].record(torch.cuda.current_stream(dev))
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(free[slot])            # the consumer is done with this buffer
                bufs[slot].copy_(host, non_blocking=True)
                filled[slot].record(copy_stream)
            # keep the pinned source alive until the copy has been consumed
            pending.append((slot, host))

        k = 0
        for host in it:
            issue(k % self.depth, host)
            k += 1
            if len(pending) == self.depth:
                break
        while pending:
            slot, _host = pending.pop(0)
            torch.cuda.current_stream(dev).wait_event(filled[slot])
            yield bufs[slot]
            free[slot].record(torch.cuda.current_stream(dev))   # consumer's work on this buffer is enqueued
            nxt = next(it, None)
            if nxt is not None:
                issue(slot, nxt)
