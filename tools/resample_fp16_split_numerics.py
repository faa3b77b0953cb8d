#!/usr/bin/env python
"""Would fp16 matrix products do for the resampler?  (design study for csrc/fir.hip's MFMA kernels, CPU only)

The 44.1 -> 16 kHz polyphase product runs on v_mfma_f32_16x16x4_f32 (1/16 of the fp16 rate) and is bound by that pipe.
Split samples and taps into fp16 high + low halves: x = xh + xl, w = wh + wl; fp16 x fp16 products are exact in fp32, so
    x w ~= xh wh + (xh wl + xl wh)          (three 16x16x16 products; the dropped xl wl term is 2^-22 relative)
with fp32 accumulation.  This script measures that against float64 on the product's own tap bank, next to plain fp32:
the split matches fp32 (5e-7 of the row maximum) PROVIDED the samples are scaled into fp16's normal range first (a power
of two per tile; quiet material otherwise loses its low halves to fp16 subnormals: 1.4e-4 at an amplitude of 1e-4).

    python tools/resample_fp16_split_numerics.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiotools_amd import tables  # noqa: E402
bank, old, new, width = tables.resample_bank(44100, 16000)
W = bank.numpy().astype(np.float32)            # (new, taps)
taps = W.shape[1]
rng = np.random.default_rng(0)

def split16(a):
    hi = a.astype(np.float16)
    lo = (a - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)

def frames(x, nf):
    return np.stack([x[f * old: f * old + taps] for f in range(nf)])   # (nf, taps)

def run(x, label):
    nf = (len(x) - taps) // old
    A = frames(x, nf)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    f32 = (A @ W.T).astype(np.float64)                                   # fp32 products, (numpy: pairwise fp32 sums)
    Ah, Al = split16(A); Wh, Wl = split16(W)
    # fp16 x fp16 products are exact in fp32; accumulation in fp32
    s3 = (Ah @ Wh.T + (Ah @ Wl.T + Al @ Wh.T)).astype(np.float64)
    s4 = (Ah @ Wh.T + (Ah @ Wl.T + Al @ Wh.T) + Al @ Wl.T).astype(np.float64)
    # scaled variant: scale x by 2^k so that the low halves stay out of the fp16 subnormals
    sc = 2.0 ** np.floor(np.log2(60000.0 / max(np.abs(A).max(), 1e-30)))
    Ah2, Al2 = split16((A * sc).astype(np.float32)); Wh2, Wl2 = split16((W * 1024).astype(np.float32))
    s3s = ((Ah2 @ Wh2.T + (Ah2 @ Wl2.T + Al2 @ Wh2.T)) / (sc * 1024)).astype(np.float64)
    m = np.abs(ref).max()
    print(f"{label:28s} fp32 {np.abs(f32-ref).max()/m:.2e}  split3 {np.abs(s3-ref).max()/m:.2e}  split4 {np.abs(s4-ref).max()/m:.2e}  split3 scaled {np.abs(s3s-ref).max()/m:.2e}")

n = 44100 * 2
t = np.arange(n) / 44100
run((0.1 * rng.standard_normal(n)).clip(-1, 1).astype(np.float32), "white 0.1")
run((0.9 * np.sin(2 * np.pi * 440 * t)).astype(np.float32), "sine 440 Hz 0.9")
run((1e-4 * rng.standard_normal(n)).astype(np.float32), "white 1e-4 (quiet)")
x = (0.5 * np.sin(2 * np.pi * 1000 * t) + 1e-5 * rng.standard_normal(n)).astype(np.float32)
run(x, "tone + -100 dB noise")
run((rng.standard_normal(n) * np.exp(-t * 8)).astype(np.float32).clip(-1, 1), "decaying burst")
