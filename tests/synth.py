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


def structured_batch(T, sample_rate=44100, seed=7):
    """(6, 1, T) float32 signals with STRUCTURE, one per row (parity tests; white noise is spectrally flat):
    0 tonal: three sines, the weakest 100 dB below the strongest;  1 pink (1/f power) noise;
    2 logarithmic sweep 20 Hz -> 0.45 sr;  3 100 dB of dynamic range in time: a -100 dBFS noise floor with
    full-scale 5 ms bursts;  4 an impulse train (every bin excited, phases aligned);  5 DC + Nyquist-rate
    alternation + a slow ramp (the self-paired bins of the real FFT)."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(T, dtype=torch.float64) / sample_rate
    rows = []
    rows.append(0.8 * torch.sin(2 * np.pi * 997.0 * t) + 8e-3 * torch.sin(2 * np.pi * 5003.0 * t + 0.3)
                + 8e-6 * torch.sin(2 * np.pi * 12007.0 * t + 1.1))
    w = torch.randn(T, generator=g, dtype=torch.float64)
    W = torch.fft.rfft(w)
    f = torch.arange(W.shape[0], dtype=torch.float64).clamp_min(1.0)
    pink = torch.fft.irfft(W / f.sqrt(), n=T)
    rows.append(0.5 * pink / pink.abs().max())
    f0, f1 = 20.0, 0.45 * sample_rate
    k = np.log(f1 / f0) / max(float(t[-1]), 1e-9)
    rows.append(0.7 * torch.sin(2 * np.pi * f0 * (torch.exp(k * t) - 1.0) / k))
    dyn = 1e-5 * torch.randn(T, generator=g, dtype=torch.float64)
    burst = max(int(0.005 * sample_rate), 1)
    for c in range(0, T, max(T // 7, burst + 1)):
        dyn[c: c + burst] += 0.9 * torch.randn(min(burst, T - c), generator=g, dtype=torch.float64).clamp(-1, 1)
    rows.append(dyn)
    imp = torch.zeros(T, dtype=torch.float64)
    imp[:: max(T // 13, 1)] = 0.9
    rows.append(imp)
    alt = 0.25 + 0.25 * (1.0 - 2.0 * (torch.arange(T) % 2).double()) + 0.3 * t / max(float(t[-1]), 1e-9)
    rows.append(alt)
    return torch.stack(rows)[:, None, :].float()
