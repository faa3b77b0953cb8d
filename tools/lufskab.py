#!/usr/bin/env python
"""Interleaved A/B of the two hop-energy kernels of csrc/loudness.hip in ONE process (GPU box).

AT_LUFS_TUNE=1 makes at_lufs_f32 re-read AT_LUFS_KERNEL on every call: 0 = kweight_hop_energy_dma (LDS-DMA
prefetch, scalar bookkeeping, piecewise hop energies), 1 = kweight_hop_energy (register-staged, round 1-2).
Round-robin timing (box drift hits both alike); the LUFS outputs of the two kernels are compared.
usage: python tools/lufskab.py [batch] [sr]
"""
import os

os.environ.setdefault("AT_DEV_KNOBS", "1")      # A/B tool: the development build of the library (lib/libaudiotools_amd_dev.so) and its AT_* switches
import sys

os.environ["AT_LUFS_TUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiotools_amd import kernels  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
SR = int(sys.argv[2]) if len(sys.argv) > 2 else 44100
C, T = 2, 10 * SR
g = torch.Generator(device="cuda").manual_seed(3)
x = (0.1 * torch.randn(B, C, T, device="cuda", generator=g)).clamp_(-1, 1)
x *= (10 ** (-30 * torch.rand(B, 1, 1, device="cuda", generator=g) / 20))
x[::7, :, T // 3: T // 3 + 2 * SR] = 0.0      # digital-silence gaps: the absolute gate
outs = {}
for k in (0, 1):
    os.environ["AT_LUFS_KERNEL"] = str(k)
    outs[k] = kernels.integrated_loudness(x, SR).clone()
torch.cuda.synchronize()
d = (outs[0] - outs[1]).abs().max().item()
print(f"batch {B} sr {SR}: max |LUFS(dma) - LUFS(staged)| = {d:.3e} LU", flush=True)
times = {0: [], 1: []}
for rep in range(5):
    for k in (0, 1):
        os.environ["AT_LUFS_KERNEL"] = str(k)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            kernels.integrated_loudness(x, SR)
        b.record()
        torch.cuda.synchronize()
        times[k].append(a.elapsed_time(b) / 20)
nbytes = B * C * T * 4
for k, name in ((0, "kweight_hop_energy_dma"), (1, "kweight_hop_energy (staged)")):
    t = sorted(times[k])
    print(f"{name:32s} median {t[2]:.4f} ms  min {t[0]:.4f}  {nbytes / t[2] / 1e9:.2f} TB/s = {100 * nbytes / t[2] / 1e9 / 8:.1f} % of 8 TB/s",
          flush=True)
