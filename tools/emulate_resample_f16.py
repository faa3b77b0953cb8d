#!/usr/bin/env python
"""CPU model of an fp16-split MFMA resampler (not a kernel yet).

Everything the planned kernel would do to the numbers, in numpy, on whole signals:
  * the reference's framing (julius.resample_frac: replicate pad (width, width + old), one frame of `new` outputs per `old`
    inputs, floor(new T / old) samples kept), tiles of 16 frames, phase blocks of 16 with the union support window of
    their taps, K-steps of 16 taps -- one v_mfma_f32_16x16x16_f16 each: lane (row i, group g) holds taps 16 s + 4 g .. + 3;
  * per tile: a power-of-two scale that brings max |x| just under 2^15, samples split into fp16 high + low planes;
  * operand fetch as the kernel would do it: three ALIGNED dwords of a plane and two v_alignbit_b32 with a per-lane shift
    of 0 or 16 bits (a row starts at an odd sample every second frame: frames are 441 samples apart);
  * products hh + (hl + lh) of exactly representable fp16 x fp16 terms, accumulated in fp32; taps split the same way
    after a fixed scale of 2^10.
`python tools/emulate_resample_f16.py` prints the error against float64 next to plain fp32's for a few inputs.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiotools_amd import tables  # noqa: E402

W_SCALE = 1024.0


def split16(a):
    hi = a.astype(np.float16)
    lo = (a - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def plan(old_sr, new_sr):
    bank, old, new, width = tables.resample_bank(old_sr, new_sr)
    b = bank.numpy().astype(np.float32)
    thr = 1e-12 * np.abs(b).max()
    NPB = (new + 15) // 16
    lo = np.zeros(NPB, dtype=np.int64)
    span = 0
    for P in range(NPB):
        rows = b[16 * P: min(16 * P + 16, new)]
        nz = np.nonzero((np.abs(rows) > thr).any(0))[0]
        lo[P] = nz[0]
        span = max(span, int(nz[-1]) + 1 - int(nz[0]))
    NS = (span + 15) // 16                                  # K-steps of 16 taps
    # B operands: Bh / Bl [P, s, k (16 taps), j (16 phases)] as fp32 values of fp16 numbers
    Bw = np.zeros((NPB, NS, 16, 16), dtype=np.float32)
    for P in range(NPB):
        for j in range(16):
            ph = 16 * P + j
            if ph >= new:
                continue
            taps = lo[P] + np.arange(16 * NS)
            ok = taps < b.shape[1]
            v = np.where(ok, b[ph, np.minimum(taps, b.shape[1] - 1)], 0.0)
            v = np.where(np.abs(v) > thr, v, 0.0)
            Bw[P, :, :, j] = (v * W_SCALE).reshape(NS, 16)
    Bh, Bl = split16(Bw)
    return dict(b=b, old=old, new=new, width=width, NPB=NPB, NS=NS, lo=lo, Bh=Bh.astype(np.float32), Bl=Bl.astype(np.float32))


def fetch4(plane_u16, a):
    """Halfs a .. a + 3 of a plane through three aligned dwords and two v_alignbit_b32 (a: int array, one per lane)."""
    dw = plane_u16.view(np.uint32).astype(np.uint64)
    d0 = a >> 1
    sh = (16 * (a & 1)).astype(np.uint64)
    x0, x1, x2 = dw[d0], dw[d0 + 1], dw[d0 + 2]
    r0 = (((x1 << np.uint64(32)) | x0) >> sh) & np.uint64(0xffffffff)
    r1 = (((x2 << np.uint64(32)) | x1) >> sh) & np.uint64(0xffffffff)
    h = np.stack([r0 & np.uint64(0xffff), r0 >> np.uint64(16), r1 & np.uint64(0xffff), r1 >> np.uint64(16)], -1).astype(np.uint16)
    return h.view(np.float16).astype(np.float32)


def resample(x, old_sr, new_sr, pl=None):
    """x (T,) float32 -> (floor(new T / old),) float32, the way the planned kernel would compute it."""
    pl = pl or plan(old_sr, new_sr)
    old, new, width, NPB, NS, lo = pl["old"], pl["new"], pl["width"], pl["NPB"], pl["NS"], pl["lo"]
    T = len(x)
    out_len = new * T // old
    xp = np.pad(x.astype(np.float32), (width, width + old), mode="edge")
    taps = pl["b"].shape[1]
    n_frames = (len(xp) - taps) // old + 1
    y = np.zeros(n_frames * new + 16 * new, dtype=np.float32)
    i16, g4 = np.arange(64) % 16, np.arange(64) // 16          # lane -> (row, k group)
    for f0 in range(0, n_frames, 16):
        need = 15 * old + int(lo.max()) + 16 * NS + 8          # samples a tile touches (+ the dwords of the last fetch)
        xs = np.zeros(need + (need & 1), dtype=np.float32)
        seg = xp[f0 * old: f0 * old + need]
        xs[:len(seg)] = seg                                    # past the padded row: zeros
        m = float(np.abs(xs).max())
        scale = 2.0 ** np.floor(np.log2(32768.0 / m)) if m > 0 else 1.0
        hi, lw = split16(xs * np.float32(scale))
        hi_u, lo_u = hi.view(np.uint16), lw.view(np.uint16)
        for P in range(NPB):
            acc = np.zeros((16, 16), dtype=np.float32)
            for s in range(NS):
                a = i16 * old + lo[P] + 16 * s + 4 * g4        # first half of the lane's four
                Ah = np.zeros((16, 16), dtype=np.float32)
                Al = np.zeros((16, 16), dtype=np.float32)
                fh, fl = fetch4(hi_u, a), fetch4(lo_u, a)
                for lane in range(64):
                    Ah[i16[lane], 4 * g4[lane]: 4 * g4[lane] + 4] = fh[lane]
                    Al[i16[lane], 4 * g4[lane]: 4 * g4[lane] + 4] = fl[lane]
                Bh, Bl = pl["Bh"][P, s], pl["Bl"][P, s]
                acc = acc + (Ah @ Bh + (Ah @ Bl + Al @ Bh)).astype(np.float32)
            res = acc * np.float32(1.0 / (scale * W_SCALE))
            for i in range(16):
                f = f0 + i
                if f < n_frames:
                    ncol = min(16, new - 16 * P)
                    y[f * new + 16 * P: f * new + 16 * P + ncol] = res[i, :ncol]
    return y[:out_len]


def reference(x, old_sr, new_sr, dtype=np.float64):
    pl = plan(old_sr, new_sr)
    old, new, width = pl["old"], pl["new"], pl["width"]
    b = pl["b"].astype(dtype)
    xp = np.pad(x.astype(dtype), (width, width + old), mode="edge")
    taps = b.shape[1]
    n_frames = (len(xp) - taps) // old + 1
    fr = np.stack([xp[f * old: f * old + taps] for f in range(n_frames)])
    return (fr @ b.T).reshape(-1)[: new * len(x) // old]


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    n = 30000
    t = np.arange(n) / 44100
    cases = {"white 0.1": 0.1 * rng.standard_normal(n), "sine 440 Hz 0.9": 0.9 * np.sin(2 * np.pi * 440 * t),
             "white 1e-4": 1e-4 * rng.standard_normal(n), "tone + noise at -100 dB": 0.5 * np.sin(2 * np.pi * 1000 * t) + 1e-5 * rng.standard_normal(n),
             "unclipped (|x| up to 5)": 1.5 * rng.standard_normal(n)}
    pl = plan(44100, 16000)
    print("blocks", pl["NPB"], "K-steps of 16 taps", pl["NS"])
    for name, x in cases.items():
        x = x.astype(np.float32)
        ref = reference(x, 44100, 16000)
        f32 = reference(x, 44100, 16000, np.float32).astype(np.float64)
        got = resample(x, 44100, 16000, pl).astype(np.float64)
        m = np.abs(ref).max()
        print(f"{name:26s} fp32 {np.abs(f32 - ref).max() / m:.2e}   fp16 split {np.abs(got - ref).max() / m:.2e}")
