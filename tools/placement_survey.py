#!/usr/bin/env python
"""Which kernels besides the fused STFT follow the physical placement of their buffers?  

For every op: K input tensors (each its own allocation, all alive) and K output placements.  The op allocates its own output
from torch's caching allocator; freeing it right after the call hands the SAME block to the next call, so `iters` launches
measure one placement; keeping one result alive as a spacer moves the following calls to another block.
  phase 1: output block fixed (the first one), the K inputs in turn, two interleaved rounds;
  phase 2: input fixed (the first one), K output blocks in turn.
Prints min / max / spread of the per-placement times.  The STFT pool is disabled (plain allocations throughout).
usage: python tools/placement_survey.py [--k 5] [--iters 8] [--ops resample,stft4096,stft512,istft,lufs,lowpass,convolve]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audiotools_amd as A
from audiotools_amd import kernels, tables

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=5)
ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--ops", default="resample,stft4096,stft512,stft1024,istft,lufs,lowpass,convolve")
args = ap.parse_args()
dev = torch.device("cuda")
kernels.output_placement(enabled=False)
K = args.k


def rand(shape, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (0.1 * torch.randn(*shape, device=dev, generator=g)).clamp_(-1, 1)


def timed(fn, x):
    y = fn(x); del y
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.iters):
        y = fn(x); del y
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / args.iters


def survey(name, make_inputs, fn, gb):
    xs = make_inputs()
    y = fn(xs[0]); del y
    torch.cuda.synchronize()
    t_in = [[], []]
    for rnd in range(2):
        for x in xs:
            t_in[rnd].append(timed(fn, x))
    tin = [min(a, b) for a, b in zip(*t_in)]
    spacers, tout = [], []
    for _ in range(K):
        tout.append(min(timed(fn, xs[0]), timed(fn, xs[0])))
        spacers.append(fn(xs[0]))          # stays alive: the next calls get another block
    torch.cuda.synchronize()
    f = lambda v: " ".join(f"{t:.3f}" for t in v)
    best = min(tin + tout)
    print(f"{name:10s} {gb:6.2f} GB  best {best:.3f} ms = {100 * gb / best / 8:.1f} %  | inputs: {f(tin)}  spread {100 * (max(tin) / min(tin) - 1):.1f} %"
          f"  | outputs: {f(tout)}  spread {100 * (max(tout) / min(tout) - 1):.1f} %", flush=True)
    del xs, spacers
    torch.cuda.empty_cache()


for op in args.ops.split(","):
    if op == "resample":
        T = 1323000
        survey(op, lambda: [rand((256, 2, T), 10 + i) for i in range(K)], lambda x: kernels.resample(x, 44100, 16000),
               512 * (T + 480000) * 4 / 1e9)
    elif op in ("stft4096", "stft512", "stft1024", "stft2048"):
        n_fft = int(op[4:])
        sr, B = {4096: (96000, 256), 512: (16000, 512), 1024: (22050, 512), 2048: (44100, 512)}[n_fft]
        T, hop = 10 * sr, n_fft // 4
        win = tables.window("hann", n_fft, dev)
        N, F = 1 + T // hop, n_fft // 2 + 1
        survey(op, lambda: [rand((B, 2, T), 20 + i) for i in range(K)], lambda x: kernels.stft_mel(x, win, n_fft, hop)[0],
               2 * B * (T * 4 + N * F * 8) / 1e9)
    elif op == "istft":
        n_fft, hop, T, B = 2048, 512, 441000, 512
        win = tables.window("hann", n_fft, dev)
        N, F = 1 + T // hop, n_fft // 2 + 1

        def specs():
            out = []
            for i in range(K):
                out.append(kernels.stft_mel(rand((B, 2, T), 30 + i), win, n_fft, hop)[0])
            return out
        survey(op, specs, lambda X: kernels.istft(X, win, n_fft, hop, T), 2 * B * (T * 4 + N * F * 8) / 1e9)
    elif op == "lufs":
        T, B = 441000, 512
        survey(op, lambda: [rand((B, 2, T), 40 + i) for i in range(K)], lambda x: kernels.integrated_loudness(x, 44100), 2 * B * T * 4 / 1e9)
    elif op == "lowpass":
        T, B = 240000, 1024
        cut = torch.full((B,), 8000.0 / 48000.0, device=dev)
        hc = cut.cpu()
        survey(op, lambda: [rand((B, 1, T), 50 + i) for i in range(K)], lambda x: kernels.sinc_filter(x, cut, 51, False, host_cutoffs=hc),
               2 * B * T * 4 / 1e9)
    elif op == "convolve":
        T, B = 240000, 1024
        ir = rand((B, 1, 96000), 99) * torch.exp(-torch.arange(96000, device=dev) / (0.3 * 48000))
        irp = torch.nn.functional.pad(ir, (0, T - 96000))
        survey(op, lambda: [rand((B, 1, T), 60 + i) for i in range(K)], lambda x: kernels.fftconv(x, irp), (2 * B * T * 4 + B * 96000 * 4) / 1e9)
