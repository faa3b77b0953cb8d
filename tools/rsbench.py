#!/usr/bin/env python
"""Resampler-only run (cfg5 per-GPU share) for rocprofv3 counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audiotools_amd import kernels
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = (0.1 * torch.randn(B, 2, 1323000, device="cuda")).clamp_(-1, 1)
for _ in range(3):
    y = kernels.resample(x, 44100, 16000)
torch.cuda.synchronize()
print(y.shape)
