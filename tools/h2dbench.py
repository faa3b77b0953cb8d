#!/usr/bin/env python
"""PCIe-inclusive rate of the north-star step: batches start in PINNED HOST memory and go through
audiotools_amd.data.DeviceStager (H2D on a side stream into two device buffers, overlapped with the
compute of the previous batch).  bench.py's `value` is the device-resident rate; this is the number
DESIGN.md quotes next to it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiotools_amd as A
from audiotools_amd.data import DeviceStager

B, C, T, SR = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 2, 441000, 44100
host = [(0.1 * torch.randn(B, C, T)).clamp_(-1, 1).pin_memory() for _ in range(2)]


def run(n):
    for x in DeviceStager((host[i & 1] for i in range(n)), "cuda"):
        s = A.AudioSignal(x, SR)
        s.mel_spectrogram(80), s.loudness()


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
