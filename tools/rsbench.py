#!/usr/bin/env python
"""Resampler A/B on the cfg5 shapes (development aid): the fp16-split matrix-core kernel (at_resample_f16s_f32) against
the float32 matrix-core kernel (at_resample_mfma_f32), interleaved in one process, through the raw C ABI.
f16 = the register-prefetch form (default), f16dma = the LDS-DMA form (AT_RESAMPLE_F16_RP=0).
usage: python tools/rsbench.py [--batch 256] [--seconds 30] [--iters 20] [--rounds 3] [--only f16,f16dma,mfma]"""
import argparse
import math
import os

os.environ.setdefault("AT_DEV_KNOBS", "1")      # A/B tool: the development build of the library (lib/libaudiotools_amd_dev.so) and its AT_* switches
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AT_RESAMPLE_F16_TUNE", "1")     # the library re-reads AT_RESAMPLE_F16_RP per call
import numpy as np
import torch

from audiotools_amd import _native, tables

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--seconds", type=float, default=30.0)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--old", type=int, default=44100)
ap.add_argument("--new", type=int, default=16000)
ap.add_argument("--only", default="")
args = ap.parse_args()

dev = torch.device("cuda")
g = math.gcd(args.old, args.new)
old, new = args.old // g, args.new // g
rows, T = args.batch * 2, int(args.old * args.seconds)
x = (0.1 * torch.randn(rows, T, device=dev)).clamp_(-1, 1)
out_len = new * T // old
lib = _native.lib()
st = _native.current_stream(dev)
nbytes = rows * T * 4 + rows * out_len * 4

W, lo, _, _, width, NPB, NC, wk = tables.resample_f16_bank(old, new)
Wd, lod = torch.from_numpy(W.view(np.int32)).to(dev), torch.from_numpy(lo).to(dev)
W2, lo2, _, _, _, NPB2, NC2 = tables.resample_mfma_bank(old, new)
W2d, lo2d = torch.from_numpy(W2).to(dev), torch.from_numpy(lo2).to(dev)
y1 = torch.empty(rows, out_len, device=dev)
y2 = torch.empty(rows, out_len, device=dev)


def f16(rp="1", depth="4"):
    os.environ["AT_RESAMPLE_F16_RP"] = rp
    os.environ["AT_RESAMPLE_F16_D"] = depth
    rc = lib.at_resample_f16s_f32(_native.ptr(x), rows, T, _native.ptr(Wd), _native.ptr(lod), old, new, width, NPB, NC,
                                  int(lo.max()), wk, _native.ptr(y1), out_len, st)
    assert rc == 0, rc


def mfma():
    rc = lib.at_resample_mfma_f32(_native.ptr(x), rows, T, _native.ptr(W2d), _native.ptr(lo2d), old, new, width, NPB2, NC2,
                                  int(lo2.max()), _native.ptr(y2), out_len, st)
    assert rc == 0, rc


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(args.iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / args.iters


y3 = torch.empty(rows, out_len, device=dev)


def f16dma():
    os.environ["AT_RESAMPLE_F16_RP"] = "0"
    rc = lib.at_resample_f16s_f32(_native.ptr(x), rows, T, _native.ptr(Wd), _native.ptr(lod), old, new, width, NPB, NC,
                                  int(lo.max()), wk, _native.ptr(y3), out_len, st)
    assert rc == 0, rc


kinds = [k for k in (("f16", f16), ("f16d5", lambda: f16("1", "5")), ("f16dma", f16dma), ("mfma", mfma)) if not args.only or k[0] in args.only.split(",")]
print(f"resample {old}->{new}, rows {rows} x T {T}: {nbytes / 1e9:.3f} GB algorithmic", flush=True)
for r in range(args.rounds):
    for name, fn in kinds:
        ms = timeit(fn)
        print(f"round {r} {name:5s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s  ({100 * nbytes / ms / 1e6 / 8000:.1f}% of 8 TB/s)", flush=True)
if len(kinds) == 4:
    d = (y1 - y2).abs().amax(-1) / y2.abs().amax(-1)
    print(f"max per-row |f16 - mfma| / max|mfma| = {float(d.max()):.2e};  register-prefetch form == LDS-DMA form: {torch.equal(y1, y3)}")
