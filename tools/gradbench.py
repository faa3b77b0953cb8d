#!/usr/bin/env python
"""Forward + backward of a magnitude loss through stft(): native kernels vs torch.stft autograd on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiotools_amd as A
from audiotools_amd import spectral
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = (0.1 * torch.randn(B, 2, 441000, device="cuda")).clamp_(-1, 1)

def run(native):
    saved = spectral._native_autograd_ok
    if not native:
        spectral._native_autograd_ok = lambda *a: False
    try:
        def fn():
            xa = x.clone().requires_grad_(True)
            X = A.AudioSignal(xa, 44100).stft()
            X.abs().sum().backward()
            return xa.grad
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): g = fn()
        torch.cuda.synchronize()
        print(f"stft fwd + |X|.sum() bwd, B={B} ({'native' if native else 'torch.stft'}): {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms", flush=True)
        return g
    finally:
        spectral._native_autograd_ok = saved

g1 = run(True); g2 = run(False)
print("grad rel diff", float((g1 - g2).abs().max() / g2.abs().max()))


def run_mel(native):
    saved = spectral._native_autograd_ok
    if not native:
        spectral._native_autograd_ok = lambda *a: False
    try:
        def fn():
            xa = x.clone().requires_grad_(True)
            mel = A.AudioSignal(xa, 44100).mel_spectrogram(80)
            mel.clamp(1e-5).log10().sum().backward()
            return xa.grad
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): g = fn()
        torch.cuda.synchronize()
        print(f"log-mel loss fwd + bwd, B={B} ({'native' if native else 'torch.stft'}): {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms", flush=True)
        return g
    finally:
        spectral._native_autograd_ok = saved

m1 = run_mel(True); m2 = run_mel(False)
print("mel grad rel diff", float((m1 - m2).abs().max() / m2.abs().max()))


def run_dac_mel(native):
    """The 7-scale MelSpectrogramLoss of the Descript Audio Codec recipe, estimate vs reference."""
    from audiotools_amd import metrics
    loss = metrics.spectral.MelSpectrogramLoss(n_mels=[5, 10, 20, 40, 80, 160, 320], window_lengths=[32, 64, 128, 256, 512, 1024, 2048],
                                               mel_fmin=[0] * 7, mel_fmax=[None] * 7, pow=1.0, mag_weight=0.0)
    xs, ys = x[:32], x[32:64].clone()
    saved = spectral._native_autograd_ok
    if not native:
        spectral._native_autograd_ok = lambda *a: False
    try:
        def fn():
            xa = xs.clone().requires_grad_(True)
            loss(A.AudioSignal(xa, 44100), A.AudioSignal(ys, 44100)).backward()
            return xa.grad
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): g = fn()
        torch.cuda.synchronize()
        print(f"7-scale MelSpectrogramLoss fwd + bwd, B=32x2chx10s ({'native' if native else 'torch.stft'}): {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms", flush=True)
        return g
    finally:
        spectral._native_autograd_ok = saved

d1 = run_dac_mel(True); d2 = run_dac_mel(False)
print("7-scale grad rel diff", float((d1 - d2).abs().max() / d2.abs().max()))
