#!/usr/bin/env python
"""Per-kernel timings of one library build (AT_LIB_PATH) -- used to A/B cache-policy builds of the
streaming kernels: LUFS, inverse STFT, overlap-save FIR, resampler.  usage: python tools/ntbench.py [tag]"""
import os

os.environ.setdefault("AT_DEV_KNOBS", "1")      # A/B tool: the development build of the library (lib/libaudiotools_amd_dev.so) and its AT_* switches
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from audiotools_amd import kernels, tables  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("AT_LIB_PATH", "default")
dev = torch.device("cuda")


def timeit(fn, nbytes, label, iters=10, reps=3):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters)
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"{tag:12s} {label:16s} {med:8.3f} ms (min {ts[0]:.3f})  {nbytes / med / 1e6:8.1f} GB/s  {100 * nbytes / med / 1e6 / 8000:.1f}%", flush=True)


B, C, T, SR = 512, 2, 441000, 44100
x = (0.1 * torch.randn(B, C, T, device=dev)).clamp_(-1, 1)
rows = B * C
timeit(lambda: kernels.integrated_loudness(x, SR), rows * T * 4, "lufs B512")
x64 = x[:64].contiguous()
timeit(lambda: kernels.integrated_loudness(x64, SR), 128 * T * 4, "lufs B64", iters=30)
win = tables.window("hann", 2048, dev)
X, _ = kernels.stft_mel(x, win, 2048, 512)
N, F = 1 + T // 512, 1025
timeit(lambda: kernels.istft(X, win, 2048, 512, T), rows * T * 4 + rows * N * F * 8, "istft B512")
del X, x
xf = (0.1 * torch.randn(1024, 1, 240000, device=dev)).clamp_(-1, 1)
for L in (153, 677):
    taps = torch.randn(1024, L, device=dev) / L ** 0.5
    timeit(lambda: kernels.fir_per_item(xf, taps, method="fft"), 2 * 1024 * 240000 * 4, f"fir_fft L={L}")
del xf
xr = (0.1 * torch.randn(256, 2, 1323000, device=dev)).clamp_(-1, 1)
out_len = 1323000 * 160 // 441
timeit(lambda: kernels.resample(xr, 44100, 16000), 512 * (1323000 + out_len) * 4, "resample cfg5")
