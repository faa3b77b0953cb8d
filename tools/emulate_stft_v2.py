"""Lane-level numpy emulation of the n_fft = 2048 wave FFT of csrc/stft.hip, "paired pass 3" form
(stft_mel_kernel_v2): the last radix-4 pass gives every thread BOTH members of each (k, M - k)
pair, so the real-FFT split step runs in registers (no slab write / re-read), and the descending
partner stream is handed to its storing lane with one ds_bpermute per dword.

Checks, for one frame, that the 16 store streams + the Nyquist store reproduce numpy's rfft.
Development aid (index / twiddle / thread-0 special cases); run: python tools/emulate_stft_v2.py
"""
import numpy as np

M, L, N = 1024, 64, 2048
rng = np.random.default_rng(0)
x = rng.standard_normal(N)
w = np.hanning(N + 1)[:N]
ref = np.fft.rfft(x * w)

t = np.arange(L)


def phys(i):
    return (i ^ ((i >> 4) & 15)) + (i >> 8)   # XOR bank swizzle + one pad slot per 256 points


def dft(v, R):  # v: (R, lanes) -> (R, lanes)
    k = np.arange(R)
    return np.exp(-2j * np.pi * np.outer(k, k) / R) @ v


buf = np.zeros(M + 4, complex)
# window halved (the 1/2 of the split step rides on the window)
xs = x * (0.5 * w)
a = np.stack([xs[2 * (t + L * q)] + 1j * xs[2 * (t + L * q) + 1] for q in range(16)])
# pass 1: radix 16, NS = 1
v = dft(a, 16)
for r in range(16):
    buf[phys(16 * t + r)] = v[r]
# pass 2: radix 16, NS = 16
a = np.stack([buf[phys(t + L * q)] for q in range(16)])
tw2 = np.stack([np.exp(-2j * np.pi * (t % 16) * r / 256) for r in range(16)])
v = dft(a * tw2, 16)
o0 = (t // 16) * 256 + t % 16
for r in range(16):
    buf[phys(o0 + 16 * r)] = v[r]
# pass 3, paired: butterflies jA = t, jA' = 256 - t (thread 0: 128), jB = 64 + t, jB' = 192 - t
jA, jAp, jB, jBp = t.copy(), np.where(t == 0, 128, 256 - t), 64 + t, 192 - t


def bfly(j):
    inp = np.stack([buf[phys(j + 256 * r)] * np.exp(-2j * np.pi * j * r / 1024) for r in range(4)])
    return dft(inp, 4)  # row r' = Z[j + 256 r']


ZA, ZAp, ZB, ZBp = bfly(jA), bfly(jAp), bfly(jB), bfly(jBp)
Zfull = np.fft.fft(xs[0::2] + 1j * xs[1::2])
for r in range(4):
    assert np.allclose(ZA[r], Zfull[jA + 256 * r]) and np.allclose(ZBp[r], Zfull[jBp + 256 * r])


def split(zk, zm, k):
    c, s = np.cos(2 * np.pi * k / N), np.sin(2 * np.pi * k / N)
    sr, si = zk.real + zm.real, zk.imag - zm.imag
    dr, di = zk.real - zm.real, zk.imag + zm.imag
    pp, qq = s * dr - c * di, s * di + c * dr
    return (sr - pp) + 1j * (si - qq), (sr + pp) + 1j * (-si - qq)   # X[k], X[M - k]


t0 = t == 0
sA = []   # pair-A slots r = 0..3
for r in range(4):
    zk, zm, k = ZA[r], ZAp[3 - r], t + 256 * r
    if r == 0:
        zk = np.where(t0, ZAp[0], zk); k = np.where(t0, 128, k)          # zm = ZA'[3] for both
    if r == 1:
        zm = np.where(t0, ZA[3], zm)                                     # k = 256 for both
    if r == 2:
        zk = np.where(t0, ZAp[1], zk); zm = np.where(t0, ZAp[2], zm); k = np.where(t0, 384, k)
    sA.append(split(zk, zm, k))
sB = [split(ZB[r], ZBp[3 - r], 64 + t + 256 * r) for r in range(4)]
dc = 2 * (ZA[0].real + ZA[0].imag) + 0j
nyq = 2 * (ZA[0].real - ZA[0].imag) + 0j
x512 = 2 * np.conj(ZA[2])
ascA = [np.where(t0, dc, sA[0][0]), sA[1][0], np.where(t0, x512, sA[2][0]), np.where(t0, sA[1][1], sA[3][0])]
ascB = [sB[r][0] for r in range(4)]
sendAp = [np.where(t0, sB[3 - m][1], sA[3 - m][1]) for m in range(4)]
t0B = [sA[0][0], sA[2][0], sA[2][1], sA[0][1]]
sendBp = [np.where(t0, t0B[m], sB[3 - m][1]) for m in range(4)]
src = (64 - t) & 63          # bpermute: lane l reads lane (64 - l) mod 64
recvAp = [s[src] for s in sendAp]
recvBp = [s[src] for s in sendBp]

X = np.full(M + 1, np.nan + 0j)
for m in range(4):
    X[256 * m + t] = ascA[m]
    X[256 * m + 64 + t] = ascB[m]
    X[256 * m + 128 + t] = recvBp[m]
    X[256 * m + 192 + t] = recvAp[m]
X[M] = nyq[0]
err = np.abs(X - ref).max() / np.abs(ref).max()
print("max rel err vs numpy rfft:", err)
assert err < 1e-12
print("OK")
