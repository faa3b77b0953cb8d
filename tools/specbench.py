#!/usr/bin/env python
"""Timing of the in-place stft_data edits against the reference's polar formulation on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audiotools_amd as A
from audiotools_amd import kernels
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = (0.1 * torch.randn(B, 2, 441000, device="cuda")).clamp_(-1, 1)
s = A.AudioSignal(x, 44100)
s.stft()
X0 = s.stft_data

def timed(fn, label, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{label:44s} {(time.perf_counter() - t0) / n * 1e3:8.2f} ms", flush=True)

def run(name, native, **kw):
    def fn():
        s.stft_data = X0
        if native:
            getattr(s, name)(**kw)
        else:
            saved = kernels.spec_native
            kernels.spec_native = lambda X: False
            try: getattr(s, name)(**kw)
            finally: kernels.spec_native = saved
    timed(fn, f"{name} ({'native' if native else 'torch polar chain'})")

for name, kw in (("mask_frequencies", dict(fmin_hz=1000.0, fmax_hz=3000.0)), ("mask_timesteps", dict(tmin_s=2.0, tmax_s=2.5)),
                 ("shift_phase", dict(shift=1.0)), ("mask_low_magnitudes", dict(db_cutoff=-10.0))):
    run(name, True, **kw)
    run(name, False, **kw)

# ---- the SpectralTransform round trip: edit + istft, eager (edit kernel, then inverse) vs deferred (inverse applies it)
from audiotools_amd import transforms as tfm
win_len = s.stft_params.window_length
for name, kw in (("mask_frequencies", dict(fmin_hz=1000.0, fmax_hz=3000.0)), ("shift_phase", dict(shift=1.0)),
                 ("mask_low_magnitudes", dict(db_cutoff=-10.0))):
    def eager():
        s.stft_data = X0
        getattr(s, name)(**kw)
        s.istft()
    def deferred():
        s.stft_data = X0
        tfm._deferring(s, lambda: getattr(s, name)(**kw))
        s.istft()
    def plain():
        s.stft_data = X0
        s.istft()
    timed(eager, f"{name} + istft, eager")
    timed(deferred, f"{name} + istft, folded into the inverse")
timed(plain, "istft alone")
