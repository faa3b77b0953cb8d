#!/usr/bin/env python
"""FIR-only run (cfg4 equalizer shape: B=1024 mono 5 s @ 48 kHz, 677 taps per item) for counter passes / timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiotools_amd import kernels
B, T, L = 1024, 240000, int(sys.argv[1]) if len(sys.argv) > 1 else 677
x = (0.1 * torch.randn(B, 1, T, device="cuda")).clamp_(-1, 1)
taps = torch.randn(B, L, device="cuda") / L ** 0.5
method = sys.argv[2] if len(sys.argv) > 2 else "fft"
for _ in range(2):
    y = kernels.fir_per_item(x, taps, method=method)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    y = kernels.fir_per_item(x, taps, method=method)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
print(f"fir {method} L={L}: {ms:.3f} ms  {2 * B * T * 4 / ms / 1e6:.0f} GB/s algorithmic")
