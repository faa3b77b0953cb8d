#!/usr/bin/env python
"""Interleaved A/B of the v2 STFT kernel's schedule knobs in ONE process (GPU box).

The knobs are environment variables that csrc/stft.hip re-reads on every call when AT_STFT_TUNE=1:
  AT_STFT_RUNMAX     consecutive frames a wave handles before it jumps ahead
  AT_STFT_NX         contiguous spans the frame range is cut into (one per XCD)
  AT_STFT_STAGGERV2  start-up delay per wave slot of a CU, units of 64 cycles
  AT_STFT_FLAGS      1: nt stores, 4: nt loads, 8: raised priority while a frame's loads + stores
                     are issued, 16: the two waves of a SIMD at different static priority
Configurations are "run:stagger:flags[:nx]" and are timed round-robin (--reps rounds of --iters
launches between one event pair each), so box drift hits all of them alike; every configuration is
checked bit-for-bit against the default one.

usage: python tools/stftsweep.py [--batch 512] [--mel 1] --cfg 16:0:0,431:16:0,...
"""
import argparse
import os

os.environ.setdefault("AT_DEV_KNOBS", "1")      # A/B tool: the development build of the library (lib/libaudiotools_amd_dev.so) and its AT_* switches
import sys

os.environ["AT_STFT_TUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiotools_amd import kernels, tables  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--mel", type=int, default=1)
ap.add_argument("--cfg", default="16:0:0")
ap.add_argument("--tag", default="")
args = ap.parse_args()

dev = torch.device("cuda")
B, C, T, SR = args.batch, 2, 441000, 44100
g = torch.Generator(device=dev).manual_seed(5)
x = (0.1 * torch.randn(B, C, T, device=dev, generator=g)).clamp_(-1, 1)
n_fft, hop = 2048, 512
win = tables.window("hann", n_fft, dev)
info, w = tables.mel_units(SR, n_fft, 80, 0.0, None, dev)
rows, N, F = B * C, 1 + T // hop, n_fft // 2 + 1
nbytes = rows * T * 4 + rows * N * F * 8 + (rows * N * 80 * 4 if args.mel else 0)
mel = (info, w, 80) if args.mel else None


def run():
    return kernels.stft_mel(x, win, n_fft, hop, mel=mel)


def setenv(c):
    os.environ["AT_STFT_RUNMAX"], os.environ["AT_STFT_STAGGERV2"], os.environ["AT_STFT_FLAGS"] = str(c[0]), str(c[1]), str(c[2])
    os.environ["AT_STFT_NX"] = str(c[3])


cfgs = []
for item in args.cfg.split(","):
    v = [int(q) for q in item.split(":")]
    cfgs.append(tuple(v + [8] * (4 - len(v))))

setenv((16, 0, 0, 8))
ref = run()
torch.cuda.synchronize()
ref_X = ref[0].clone()
ref_m = ref[1].clone() if args.mel else None
del ref

times = {c: [] for c in cfgs}
ok = {}
for rep in range(args.reps):
    for c in cfgs:
        setenv(c)
        out = run()
        if rep == 0:
            torch.cuda.synchronize()
            ok[c] = torch.equal(out[0], ref_X) and (ref_m is None or torch.equal(out[1], ref_m))
        del out
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            run()
        b.record()
        torch.cuda.synchronize()
        times[c].append(a.elapsed_time(b) / args.iters)

print(f"# lib={os.environ.get('AT_LIB_PATH', 'default')} batch={B} mel={args.mel} iters={args.iters} reps={args.reps} {args.tag}")
print(f"{'run':>5s} {'stag':>4s} {'fl':>3s} {'nx':>3s} {'ms(med)':>9s} {'ms(min)':>9s} {'ms(max)':>9s} {'TB/s':>6s} {'%':>6s}  ok")
for c in cfgs:
    ts = sorted(times[c])
    med = ts[len(ts) // 2]
    print(f"{c[0]:5d} {c[1]:4d} {c[2]:3d} {c[3]:3d} {med:9.3f} {ts[0]:9.3f} {ts[-1]:9.3f} {nbytes / med / 1e9:6.2f} "
          f"{100 * nbytes / med / 1e6 / 8000:6.1f}  {ok[c]}", flush=True)
