#!/usr/bin/env python
"""can the north-star step read x from HBM once?  A/B of the two kernels of the step
(stft_mel_kernel_v2 and kweight_hop_energy_dma walk the same rows) launched back to back on ONE stream against launched on
TWO streams so that the second reader may find x in L2 / the 256 MB Infinity Cache.  Prints ms per step of both orders."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audiotools_amd as A
from audiotools_amd import kernels, tables

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda")
x = (0.1 * torch.randn(B, 2, 441000, device=dev)).clamp_(-1, 1)
win = tables.window("hann", 2048, dev)
info, w = tables.mel_units(44100, 2048, 80, 0.0, None, dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def serial():
    kernels.stft_mel(x, win, 2048, 512, mel=(info, w, 80))
    kernels.integrated_loudness(x, 44100)


def serial_lufs_first():
    kernels.integrated_loudness(x, 44100)
    kernels.stft_mel(x, win, 2048, 512, mel=(info, w, 80))


def two_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        kernels.stft_mel(x, win, 2048, 512, mel=(info, w, 80))
    with torch.cuda.stream(s2):
        kernels.integrated_loudness(x, 44100)
    cur.wait_stream(s1); cur.wait_stream(s2)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    for _ in range(n):
        fn()
    e[1].record()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / n


for r in range(3):
    print(f"round {r}: one stream mel->lufs {timeit(serial):.3f} ms, lufs->mel {timeit(serial_lufs_first):.3f} ms, two streams {timeit(two_streams):.3f} ms", flush=True)
