#!/usr/bin/env python
"""A/B of two builds of the LUFS entry point in ONE process (round-robin), both libraries loaded side by
side through ctypes.  usage: python tools/lufsab.py libA.so libB.so [batch]"""
import ctypes
import os

os.environ.setdefault("AT_DEV_KNOBS", "1")      # A/B tool: the development build of the library (lib/libaudiotools_amd_dev.so) and its AT_* switches
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from audiotools_amd import _native, kernels, tables  # noqa: E402

B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
C, T, SR = 2, 441000, 44100
x = (0.1 * torch.randn(B, C, T, device="cuda")).clamp_(-1, 1)
libs = []
for p in sys.argv[1:3]:
    h = ctypes.CDLL(os.path.abspath(p))
    for name in ("at_lufs_f32", "at_lufs_workspace_bytes"):
        res, args = _native.SIGNATURES[name]
        getattr(h, name).restype, getattr(h, name).argtypes = res, args
    libs.append(h)
sos, gains = tables.weighting_sos(SR)
sos = np.ascontiguousarray(sos, dtype=np.float64)
gains = np.ascontiguousarray(gains, dtype=np.float64)
K, S = kernels.lufs_block_params(SR, 0.4)
need = libs[0].at_lufs_workspace_bytes(B, C, T, K, S)
ws = torch.empty(int(need), dtype=torch.uint8, device="cuda")
warm = min(tables.lufs_warmup(sos), 1 << 30)
st = _native.current_stream(x.device)
outs = [torch.empty(B, device="cuda") for _ in libs]


def run(i):
    rc = libs[i].at_lufs_f32(_native.ptr(x), B, C, T, sos.ctypes.data, gains.ctypes.data, len(sos), K, S, 1.0 / (0.4 * SR),
                             float("nan"), warm, _native.ptr(outs[i]), _native.ptr(ws), ws.numel(), st)
    assert rc == 0, rc


run(0); run(1)
torch.cuda.synchronize()
print("bit-equal:", torch.equal(outs[0], outs[1]), flush=True)
times = [[], []]
for rep in range(5):
    for i in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run(i)
        b.record()
        torch.cuda.synchronize()
        times[i].append(a.elapsed_time(b) / 20)
for p, t in zip(sys.argv[1:3], times):
    t.sort()
    print(f"{os.path.basename(p):36s} median {t[2]:.4f} ms  min {t[0]:.4f}", flush=True)
