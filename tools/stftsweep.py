#!/usr/bin/env python
"""Sweep of the v2 STFT kernel's schedule / cache-policy knobs in ONE process (GPU box).

The knobs are environment variables that csrc/stft.hip re-reads on every call when AT_STFT_TUNE=1:
AT_STFT_RUNMAX (consecutive frames a wave handles before it jumps ahead), AT_STFT_NX (number of
contiguous spans the frame range is cut into, one per XCD), AT_STFT_FLAGS (1: nt stores, 2: sc1
stores, 4: nt loads).  Every configuration is checked bit-for-bit against the default one.

usage: python tools/stftsweep.py [--batch 512] [--iters 8] [--mel 1] [--runs 8,16,...] [--nx 8] [--flags 0]
"""
import argparse
import os
import sys

os.environ["AT_STFT_TUNE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiotools_amd import kernels, tables  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--mel", type=int, default=1)
ap.add_argument("--runs", default="16")
ap.add_argument("--nx", default="8")
ap.add_argument("--flags", default="0")
ap.add_argument("--nw", default="4", help="waves per workgroup of the v2 kernel: 4 (shipped) or 12")
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


def setenv(run_max, nx, flags, nw=4):
    os.environ["AT_STFT_V2NW"] = str(nw)
    os.environ["AT_STFT_RUNMAX"] = str(run_max)
    os.environ["AT_STFT_NX"] = str(nx)
    os.environ["AT_STFT_FLAGS"] = str(flags)


setenv(16, 8, 0)
ref = run()
torch.cuda.synchronize()
ref_X = ref[0].clone()
ref_m = ref[1].clone() if args.mel else None
del ref

print(f"# lib={os.environ.get('AT_LIB_PATH', 'default')} batch={B} mel={args.mel} {args.tag}")
print(f"{'nw':>3s} {'run':>5s} {'nx':>3s} {'fl':>3s} {'ms(med)':>9s} {'ms(min)':>9s} {'TB/s':>6s} {'%':>6s}  ok")
for nw, flags in [(int(a), int(b)) for a in args.nw.split(",") for b in args.flags.split(",")]:
    for nx in [int(v) for v in args.nx.split(",")]:
        for rm in [int(v) for v in args.runs.split(",")]:
            setenv(rm, nx, flags, nw)
            out = run()
            torch.cuda.synchronize()
            ok = torch.equal(out[0], ref_X) and (ref_m is None or torch.equal(out[1], ref_m))
            del out
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
            for a, b in evs:
                a.record()
                run()
                b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            med = ts[len(ts) // 2]
            print(f"{nw:3d} {rm:5d} {nx:3d} {flags:3d} {med:9.3f} {ts[0]:9.3f} {nbytes / med / 1e9:6.2f} {100 * nbytes / med / 1e6 / 8000:6.1f}  {ok}",
                  flush=True)
