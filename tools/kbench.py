#!/usr/bin/env python
"""Per-kernel micro-benchmark on the north-star shapes (development aid).
usage: python tools/kbench.py [--batch 512] [--iters 10] [--what stft,stftmel,melonly,lufs,istft,copy]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audiotools_amd as A
from audiotools_amd import kernels, tables

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--what", default="stft,stftmel,lufs")
ap.add_argument("--sr", type=int, default=44100)
ap.add_argument("--nfft", type=int, default=2048)
ap.add_argument("--placed", type=int, default=1, help="1: opt into the placement-aware output pool (best of 12 buffers: comparable across processes); 0: plain allocations")
args = ap.parse_args()
if args.placed:
    kernels.output_placement(enabled=True, calibrate_after=1)

dev = torch.device("cuda")
B, C, SR = args.batch, 2, args.sr
T = 10 * SR
x = (0.1 * torch.randn(B, C, T, device=dev)).clamp_(-1, 1)
n_fft, hop = args.nfft, args.nfft // 4
win = tables.window("hann", n_fft, dev)
info, w = tables.mel_units(SR, n_fft, 80, 0.0, None, dev) if kernels.stft_fused_supported(n_fft) else (None, None)
rows = B * C
N = 1 + T // hop
F = n_fft // 2 + 1


def timeit(fn, nbytes, label):
    fn(); fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / args.iters
    print(f"{label:10s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s  ({100 * nbytes / ms / 1e6 / 8000:.1f}% of 8 TB/s)", flush=True)


for what in args.what.split(","):
    if what == "stft":
        timeit(lambda: kernels.stft_mel(x, win, n_fft, hop), rows * T * 4 + rows * N * F * 8, "stft")
    elif what == "stftmel":
        timeit(lambda: kernels.stft_mel(x, win, n_fft, hop, mel=(info, w, 80)),
               rows * T * 4 + rows * N * F * 8 + rows * N * 80 * 4, "stft+mel")
    elif what == "genmel":          # generic sizes (--nfft 4096 --sr 96000): banded mel fused into the tiled kernel
        band, bw = tables.mel_bands(SR, n_fft, 80, 0.0, None, dev)
        timeit(lambda: kernels.stft_mel(x, win, n_fft, hop, mel=(band, bw, 80)),
               rows * T * 4 + rows * N * F * 8 + rows * N * 80 * 4, "stft+mel(g)")
    elif what == "melonly":
        timeit(lambda: kernels.stft_mel(x, win, n_fft, hop, want_stft=False, mel=(info, w, 80)),
               rows * T * 4 + rows * N * 80 * 4, "mel-only")
    elif what == "lufs":
        timeit(lambda: kernels.integrated_loudness(x, SR), rows * T * 4, "lufs")
    elif what == "lufs3":       # the three-stage weighting class (three waves per SIMD: scan matrices of three stages in LDS)
        timeit(lambda: kernels.integrated_loudness(x, SR, "Fenton/Lee 1"), rows * T * 4, "lufs FL1")
    elif what == "istft":
        X, _ = kernels.stft_mel(x, win, n_fft, hop)
        timeit(lambda: kernels.istft(X, win, n_fft, hop, T), rows * T * 4 + rows * N * F * 8, "istft")
        del X
    elif what == "copy":
        y = torch.empty(rows * N * F * 2, device=dev)
        timeit(lambda: y.fill_(1.0), y.numel() * 4, "fill")
        z = torch.empty_like(x)
        timeit(lambda: z.copy_(x), 2 * x.numel() * 4, "copy")
for r in kernels.output_placement():       # the placement pool's calibrations (kernel ms per candidate buffer)
    print("pool", r["op"], r["shape"], " ".join(f"{t:.3f}" for t in r["calibration_ms"]))
