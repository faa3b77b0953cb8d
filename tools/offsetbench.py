#!/usr/bin/env python
"""Does the time of the north-star STFT + mel kernel (and of its zero-compute twin) depend on WHERE its three buffers lie?
(development aid, round 4: the twin's time moved between 1.65 and 2.09 ms from box to box and between a traced and an
untraced run on the same box, the real kernel's did not.)  One pool, the signal / spectrum / mel buffers carved out of it at
a list of relative offsets, both kernels through the raw C ABI, K launches back to back under HIP events.
usage: python tools/offsetbench.py [--iters 20] [--batch 512]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from audiotools_amd import _native, tables

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=512)
args = ap.parse_args()

dev = torch.device("cuda")
B, C, SR, n_fft, hop, n_mels = args.batch, 2, 44100, 2048, 512, 80
T = 10 * SR
rows, N, F = B * C, 1 + T // hop, n_fft // 2 + 1
nx, ns, nm = rows * T * 4, rows * N * F * 8, rows * N * n_mels * 4
SL = 256 << 20
pool = torch.empty(nx + ns + nm + 4 * SL, dtype=torch.uint8, device=dev)
base = (pool.data_ptr() + (1 << 21) - 1) & ~((1 << 21) - 1)        # 2 MiB aligned
xsrc = (0.1 * torch.randn(rows, T, device=dev)).clamp_(-1, 1)
win = tables.window("hann", n_fft, dev)
tw = tables.stft_twiddles(n_fft, dev)
info, w = tables.mel_units(SR, n_fft, n_mels, 0.0, None, dev)
lib = _native.lib()
_native.dev_lib()          # binds the measurement entry points (the tools run the development build: same handle)
st = _native.current_stream(dev)
nbytes = nx + ns + nm


def launch(fn, px, ps, pm):
    rc = fn(ctypes.c_void_p(px), rows, T, _native.ptr(win), _native.ptr(tw), n_fft, hop, 0, 0, 1, 0, N, ctypes.c_void_p(ps),
            _native.ptr(info), _native.ptr(w), int(info.shape[0]), n_mels, ctypes.c_void_p(pm), st)
    assert rc == 0, rc


def timeit(fn, px, ps, pm):
    for _ in range(3):
        launch(fn, px, ps, pm)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.iters):
        launch(fn, px, ps, pm)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / args.iters


def up(a, al):
    return (a + al - 1) // al * al


print(f"pool base {base:#x}; x {nx / 2**20:.1f} MiB, spectrum {ns / 2**20:.1f} MiB, mel {nm / 2**20:.1f} MiB", flush=True)
# (label, offset of x, gap before the spectrum, gap before mel), all relative to a 2 MiB-aligned packing
cases = [("packed, 2 MiB aligned", 0, 0, 0), ("+256 B / +512 B", 0, 256, 512), ("+4 KiB / +8 KiB", 0, 4096, 8192),
         ("+64 KiB / +128 KiB", 0, 65536, 131072), ("+1 MiB / +1 MiB", 0, 1 << 20, 1 << 20),
         ("+1 MiB+4 KiB / +3 MiB+12 KiB", 0, (1 << 20) + 4096, (3 << 20) + 12288), ("x +4 KiB", 4096, 0, 0),
         ("x +1 MiB", 1 << 20, 0, 0), ("+37 MiB / +91 MiB", 0, 37 << 20, 91 << 20), ("+128 MiB / +128 MiB", 0, 128 << 20, 128 << 20),
         ("mel first", -1, 0, 0)]
for label, ox, gs, gm in cases:
    if ox >= 0:
        px = base + ox
        ps = up(px + nx, 1 << 21) + gs
        pm = up(ps + ns, 1 << 21) + gm
    else:
        pm = base
        ps = up(pm + nm, 1 << 21)
        px = up(ps + ns, 1 << 21)
    assert max(px + nx, ps + ns, pm + nm) <= pool.data_ptr() + pool.numel()
    # fill the signal at its place
    off = px - pool.data_ptr()
    pool[off: off + nx].view(torch.float32).view(rows, T).copy_(xsrc)
    real = timeit(lib.at_stft_mel_f32, px, ps, pm)
    floor = timeit(lib.at_stft_mel_floor_f32, px, ps, pm)
    real2 = timeit(lib.at_stft_mel_f32, px, ps, pm)
    print(f"{label:34s} kernel {real:6.3f} / {real2:6.3f} ms ({nbytes / real / 1e6:6.0f} GB/s)   twin {floor:6.3f} ms ({nbytes / floor / 1e6:6.0f} GB/s)",
          flush=True)
