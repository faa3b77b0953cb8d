#!/usr/bin/env python
"""CPU replay of the index plan of the paired inverse kernel (csrc/istft.hip, PAIRED, M = 16 L = 1024).

A lane t < L owns the radix-4 groups j = t, t + L, 3L - t, 4L - t (lane 0: 0, L, 3L, 2L), point j + 4L r in register
b + 4 r.  The replay folds the one-sided spectrum with in-lane partners only (register 15 - q; lane 0's two self-paired
groups and the Nyquist bin by selection), runs the Stockham passes 4 . 16 . (M / 64) with the twiddle rule of
fft_wave.h (w_{NS R}^{r (j mod NS)}, outputs at (j / NS) NS R + j mod NS + r NS) and compares with numpy's irfft.

    python tools/emulate_istft_paired.py          # prints the two errors (1e-16, 1e-14)
"""
import numpy as np


def replay(M=1024, seed=0):
    L, N = M // 16, 2 * M
    R3 = M // 64                      # last radix: 16 (M = 1024), 8 (512), 4 (256)
    rng = np.random.default_rng(seed)
    X = rng.standard_normal(M + 1) + 1j * rng.standard_normal(M + 1)
    X[0] = X[0].real
    X[M] = X[M].real
    want = np.fft.irfft(X, N)
    tw = np.exp(-2j * np.pi * np.arange(N) / N)          # the kernel's table: (cos, -sin)(2 pi m / N)

    def fold_one(xk, xm, k):                             # conj(Z[k]) of z[n] = x[2n] + i x[2n + 1]
        c, s = tw[k].real, -tw[k].imag
        sr, si = xk.real + xm.real, xk.imag - xm.imag
        dr, di = xk.real - xm.real, xk.imag + xm.imag
        return complex(sr - s * dr - c * di, -(si + c * dr - s * di))

    def dft(v):
        n = len(v)
        return np.array([sum(v[m] * np.exp(-2j * np.pi * m * kk / n) for m in range(n)) for kk in range(n)])

    a_ref = np.array([fold_one(X[k], X[M - k], k) for k in range(M)])
    buf = np.zeros(M, complex)
    for t in range(L):
        jb = [t, t + L, 3 * L - t, (4 * L - t) if t else 2 * L]
        kq = lambda q: jb[q & 3] + 4 * L * (q >> 2)      # noqa: E731
        xa = [X[kq(q)] for q in range(16)]
        a = [None] * 16
        for r in range(4):                               # middle groups: one evaluation per pair
            q, qp = 1 + 4 * r, 14 - 4 * r
            k = kq(q)
            assert kq(qp) == M - k
            xk, xm = xa[q], xa[qp]
            c, s = tw[k].real, -tw[k].imag
            sr, si = xk.real + xm.real, xk.imag - xm.imag
            dr, di = xk.real - xm.real, xk.imag + xm.imag
            P, Q = s * dr + c * di, c * dr - s * di
            a[q], a[qp] = complex(sr - P, -(si + Q)), complex(sr + P, si - Q)
        for q in (0, 4, 8, 12, 3, 7, 11, 15):            # outer groups: one evaluation per register
            if t == 0:
                xm = (X[M] if q == 0 else xa[16 - q]) if (q & 3) == 0 else xa[18 - q]
            else:
                xm = xa[15 - q]
            xk = xa[q]
            if kq(q) == 0:
                xk, xm = complex(xk.real, 0.0), complex(xm.real, 0.0)
            a[q] = fold_one(xk, xm, kq(q))
        assert max(abs(a[q] - a_ref[kq(q)]) for q in range(16)) < 1e-12
        for b in range(4):                               # pass 1: R = 4, NS = 1
            v = dft(np.array([a[b + 4 * r] for r in range(4)]))
            buf[4 * jb[b]: 4 * jb[b] + 4] = v
    nxt = np.zeros(M, complex)
    for t in range(L):                                   # pass 2: R = 16, NS = 4
        v = np.array([buf[t + L * q] for q in range(16)])
        for r in range(1, 16):
            v[r] *= tw[r * (t % 4) * (N // 64)]
        v = dft(v)
        o0 = (t // 4) * 64 + t % 4
        nxt[o0 + 4 * np.arange(16)] = v
    out = np.zeros(M, complex)
    NB = 16 // R3
    for t in range(L):                                   # last pass: R = M / 64, NS = 64, in place on "point t + L q"
        pts = np.array([nxt[t + L * q] for q in range(16)])
        for b in range(NB):
            j = t + b * L
            v = np.array([pts[b + r * NB] for r in range(R3)])
            for r in range(1, R3):
                v[r] *= tw[r * (j % 64) * (N // (64 * R3))]
            v = dft(v)
            for r in range(R3):
                out[j + 64 * r] = v[r]
    z = np.fft.fft(a_ref)
    x = np.empty(N)
    x[0::2], x[1::2] = out.real / N, -out.imag / N
    return float(np.abs(np.fft.fft(a_ref) - out).max() / np.abs(z).max()), float(np.abs(x - want).max())


if __name__ == "__main__":
    for M in (1024, 512, 256):
        print(M, replay(M))
