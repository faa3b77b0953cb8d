"""Seeded synthetic signals shared by the tests, the golden-vector script and bench.py
(0.1*randn clipped to [-1,1], per-item gain 10**(U(-30,0)/20), 5 % of the
items get a 2 s digital-silence gap so the absolute gate of the LUFS meter is exercised)."""
import numpy as np
import torch


def audio_batch(B, C, T, seed=1234, device="cpu", sample_rate=44100, levels=True, gaps=True):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (0.1 * torch.randn(B, C, T, generator=g)).clamp_(-1, 1)
    if levels:
        gain = 10 ** (-30 * torch.rand(B, generator=g) / 20)
        x = x * gain[:, None, None]
    if gaps and T >= 3 * sample_rate:
        n_gap = max(1, int(round(0.05 * B)))
        for i in range(n_gap):
            start = int(torch.randint(0, T - 2 * sample_rate, (1,), generator=g))
            x[(i * 7919) % B, :, start: start + 2 * sample_rate] = 0
    return x.to(device)


def sine(freq, sr, dur, amp=1.0, channels=1):
    t = np.arange(int(sr * dur)) / sr
    x = amp * np.sin(2 * np.pi * freq * t)
    return torch.from_numpy(np.tile(x[None, None], (1, channels, 1))).float()
