#!/usr/bin/env python
"""PCIe-inclusive rate of the north-star step: the batch starts in PINNED HOST memory, H2D copies run
on a side stream into two device buffers and overlap the compute of the previous batch.
(bench.py's `value` is the device-resident rate; this is the number DESIGN.md quotes next to it.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiotools_amd as A

B, C, T, SR = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 2, 441000, 44100
host = [(0.1 * torch.randn(B, C, T)).clamp_(-1, 1).pin_memory() for _ in range(2)]
dev = [torch.empty(B, C, T, device="cuda") for _ in range(2)]
copy_stream = torch.cuda.Stream()
ready = [torch.cuda.Event() for _ in range(2)]
done = [torch.cuda.Event() for _ in range(2)]


def step(x):
    s = A.AudioSignal(x, SR)
    return s.mel_spectrogram(80), s.loudness()


def run(n):
    for k in range(n + 1):
        i = k & 1
        if k < n:
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done[i])          # buffer i free again
                dev[i].copy_(host[i], non_blocking=True)
                ready[i].record(copy_stream)
        if k > 0:
            j = (k - 1) & 1
            torch.cuda.current_stream().wait_event(ready[j])
            step(dev[j])
            done[j].record()


for i in range(2):
    done[i].record()
run(2)
torch.cuda.synchronize()
n = 8
t0 = time.perf_counter()
run(n)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
gb = B * C * T * 4 / 1e9
print(f"pinned-host -> device -> STFT+mel+LUFS, B={B}: {dt * 1e3:.2f} ms/batch, {gb / dt:.1f} GB/s over PCIe, "
      f"{B * 10.0 / dt:,.0f} audio-seconds/sec (device-resident step alone: see bench.py)")
