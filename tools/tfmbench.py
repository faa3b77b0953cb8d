#!/usr/bin/env python
"""Every loader-free transform of data.transforms on one HIP batch, one after the other: milliseconds per call (GPU span under
HIP events, parameters drawn before the timed region) next to the bytes a single read + write of the batch would move.
A development aid for finding torch fallbacks that are pathologically slow on this stack (the reference's fold formulation of
overlap_and_add ran 500 x slower than a gather kernel).
usage: python tools/tfmbench.py [--batch 256] [--seconds 5] [--sr 44100] [--iters 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audiotools_amd as A
from audiotools_amd import transforms as T

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--seconds", type=float, default=5.0)
ap.add_argument("--sr", type=int, default=44100)
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()

dev = torch.device("cuda")
B, SR = args.batch, args.sr
n = int(args.seconds * SR)
x = (0.1 * torch.randn(B, 1, n, device=dev)).clamp_(-1, 1)
rw_gb = 2 * x.numel() * 4 / 1e9

cases = [("ClippingDistortion", T.ClippingDistortion()), ("Quantization", T.Quantization()), ("MuLawQuantization", T.MuLawQuantization()),
         ("VolumeChange", T.VolumeChange()), ("VolumeNorm", T.VolumeNorm()), ("GlobalVolumeNorm", T.GlobalVolumeNorm()),
         ("LowPass", T.LowPass()), ("HighPass", T.HighPass()), ("Equalizer", T.Equalizer()), ("RescaleAudio", T.RescaleAudio()),
         ("Silence", T.Silence()), ("NoiseFloor", T.NoiseFloor()), ("Smoothing", T.Smoothing()), ("ShiftPhase", T.ShiftPhase()),
         ("InvertPhase", T.InvertPhase()), ("MaskLowMagnitudes", T.MaskLowMagnitudes()), ("CorruptPhase", T.CorruptPhase()),
         ("FrequencyMask", T.FrequencyMask()), ("TimeMask", T.TimeMask()), ("TimeNoise", T.TimeNoise()), ("FrequencyNoise", T.FrequencyNoise()),
         ("SpectralDenoising", T.SpectralDenoising())]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / args.iters


print(f"batch {B} x 1 ch x {args.seconds} s @ {SR} Hz; one read + write of the batch = {rw_gb:.3f} GB "
      f"({rw_gb / 5e3 * 1e3:.3f} ms at 5 TB/s)", flush=True)
for name, tfm in cases:
    try:
        sig = A.AudioSignal(x.clone(), SR)
        if name in ("VolumeNorm", "GlobalVolumeNorm"):
            sig.metadata["loudness"] = -20.0
        kw = tfm.batch_instantiate(list(range(B)), sig)
        kw = A.util.prepare_batch(kw, dev)

        def run():
            s = A.AudioSignal(x, SR)
            s.metadata.update(sig.metadata)
            return tfm(s.clone(), **kw)

        ms = timed(run)
        clone_ms = timed(lambda: A.AudioSignal(x, SR).clone())
        print(f"{name:20s} {ms:9.3f} ms   (clone alone {clone_ms:.3f} ms; {ms / (rw_gb / 5e3 * 1e3):6.1f} x one read+write at 5 TB/s)", flush=True)
    except Exception as e:  # a development aid: report and go on
        print(f"{name:20s} FAILED {type(e).__name__}: {str(e)[:150]}", flush=True)
