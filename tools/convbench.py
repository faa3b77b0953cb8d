#!/usr/bin/env python
"""Circular FFT convolution at the cfg4 shape: four-step engine (csrc/longconv.hip) vs rocFFT.
usage: python tools/convbench.py [--batch 1024] [--T 240000] [--iters 10]"""
import argparse
import os

os.environ.setdefault("AT_DEV_KNOBS", "1")      # A/B tool: the development build of the library (lib/libaudiotools_amd_dev.so) and its AT_* switches
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from audiotools_amd import kernels

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--T", type=int, default=240000)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--engines", default="fourstep,rocfft")
args = ap.parse_args()
dev = torch.device("cuda")
B, T = args.batch, args.T
x = torch.randn(B, 1, T, device=dev)
ir = torch.randn(B, 1, T, device=dev) * torch.exp(-torch.arange(T, device=dev) / (0.1 * T))
scale = torch.rand(B, 1, 1, device=dev) + 0.5
outs = {}
for eng in args.engines.split(","):
    fn = lambda: kernels.fftconv(x, ir, scale, engine=eng)
    outs[eng] = fn(); fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / args.iters
    alg = 3 * B * T * 4           # read x, read ir, write y
    print(f"{eng:9s} {ms:8.3f} ms   {alg / ms / 1e6:7.1f} GB/s algorithmic (x + ir + y) = {100 * alg / ms / 8e9:.1f}% of 8 TB/s", flush=True)
if len(outs) == 2:
    a, b = outs.values()
    print("max |fourstep - rocfft| / max|rocfft| = %.3g" % float((a - b).abs().max() / b.abs().max()))
