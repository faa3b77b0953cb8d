"""numpy emulation of csrc/longconv.hip, statement for statement (tile layout, in-place Stockham
passes, row-pair mirror logic, real-FFT split / merge), driven by the tables the C library
builds on the host.  Used to check the index arithmetic without a GPU:

    python tools/emulate_longconv.py [T ...]
"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from audiotools_amd import _native  # noqa: E402


def factor(n):
    radix, ns, s = [], [], 1
    for r in (25, 5, 9, 3, 7, 16, 8, 4, 2):      # csrc/longconv.hip factor(): composite radices before their primes
        while n % r == 0:
            radix.append(r); ns.append(s); s *= r; n //= r
    assert n == 1
    return radix, ns


def passes(buf, tw, N, batches, addr, step):
    """buf: flat complex array; in-place passes exactly as pass_inplace does them."""
    radix, nss = factor(N)
    for R, NS in zip(radix, nss):
        nb = N // R
        tstep = nb // NS
        total = nb * batches
        ids = np.arange(total)
        batch, j = addr["split"](ids, nb)
        jd = ((j.astype(np.float32) + np.float32(0.5)) * (np.float32(1.0) / np.float32(NS))).astype(np.int64)
        assert np.array_equal(jd, j // NS)
        k = j - jd * NS
        v = np.stack([buf[addr["addr"](batch, j + nb * q)] for q in range(R)])
        if NS > 1:
            for q in range(1, R):
                v[q] = v[q] * tw[k * q * tstep]
        # DFT of size R along axis 0
        qq = np.arange(R)
        F = np.exp(-2j * np.pi * np.outer(qq, qq) / R)
        v = F @ v
        o = addr["addr"](batch, jd * NS * R + k)
        for q in range(R):
            buf[o + step(NS) * q] = v[q]
    return buf


def emulate(T, x, h, scale=1.0):
    lib = _native.lib()
    n1, n2 = ctypes.c_int(), ctypes.c_int()
    assert lib.at_longconv_plan(T, ctypes.byref(n1), ctypes.byref(n2)) == 0
    N1, N2 = n1.value, n2.value
    M = T // 2
    nf = lib.at_longconv_table_floats(T)
    tb = np.empty(nf, dtype=np.float32)
    assert lib.at_longconv_tables_host(T, tb.ctypes.data, nf) == 0
    tb = tb[0::2].astype(np.float64) + 1j * tb[1::2].astype(np.float64)
    nhi = (N2 + 63) // 64
    rt = 64 + nhi
    o = 0
    tw1 = tb[o:o + N1]; o += N1
    tw2 = tb[o:o + N2]; o += N2
    rowtw = tb[o:o + N1 * rt].reshape(N1, rt); o += N1 * rt
    sp_lo = tb[o:o + N1]; o += N1
    sp_hi = tb[o:o + N2]; o += N2
    assert o == len(tb)
    cw = 64
    while cw > 1 and N1 * cw > 4096:
        cw >>= 1
    lcw = cw.bit_length() - 1

    def colfft(src, conj_out):
        dst = np.zeros_like(src)
        tiles = (N2 + cw - 1) // cw
        col = {"split": lambda ids, nb: (ids & (cw - 1), ids >> lcw), "addr": lambda b, p: (p << lcw) + b}
        for t in range(tiles):
            tile = np.zeros(N1 * cw, complex)
            e = np.arange(N1 * cw)
            n1_, n2_ = e >> lcw, t * cw + (e & (cw - 1))
            ok = n2_ < N2
            tile[ok] = src[n1_[ok], n2_[ok]]
            passes(tile, tw1, N1, cw, col, lambda ns: ns << lcw)
            v = np.conj(tile) if conj_out else tile
            dst[n1_[ok], n2_[ok]] = v[ok]
        return dst

    zx = (x[0::2] + 1j * x[1::2]).reshape(N1, N2)
    zh = (h[0::2] + 1j * h[1::2]).reshape(N1, N2)
    ax, ah = colfft(zx, False), colfft(zh, False)
    inv_m = 1.0 / M
    for p in range(N1 // 2 + 1):
        k1a, k1b = p, (N1 - p) % N1
        self_ = k1a == k1b
        nrow = 1 if self_ else 2
        rows = [k1a, k1b][:nrow]
        slot_b = 0 if self_ else N2
        row = {"split": lambda ids, nb: ((ids >= nb).astype(np.int64), ids - np.where(ids >= nb, nb, 0)),
               "addr": lambda b, pt: b * N2 + pt}
        n2_ = np.arange(N2)
        X = None
        for ph in range(3):
            if ph < 2:
                g = ax if ph == 0 else ah
                buf = np.concatenate([g[k] * (rowtw[k][n2_ & 63] * rowtw[k][64 + (n2_ >> 6)]) for k in rows])
            passes(buf, tw2, N2, nrow, row, lambda ns: ns)
            if ph == 2:
                break
            k2 = np.arange(N2)
            k2m = np.where(k2 == 0, 0, N2 - k2) if k1a == 0 else N2 - 1 - k2
            act = (k2 <= k2m) if self_ else np.ones(N2, bool)
            k2, k2m = k2[act], k2m[act]
            zk, zm = buf[k2].copy(), buf[slot_b + k2m].copy()
            w = sp_lo[k1a] * sp_hi[k2]
            dc = (k2 == 0) & (k1a == 0)
            sk = 0.5 * (zk + np.conj(zm)) - 0.5j * w * (zk - np.conj(zm))
            sm = np.conj(0.5 * (zk + np.conj(zm)) + 0.5j * w * (zk - np.conj(zm)))
            sk[dc] = zk[dc].real + zk[dc].imag
            sm[dc] = zk[dc].real - zk[dc].imag
            if ph == 0:
                X = (sk, sm)
            else:
                yk, ym = X[0] * sk * scale * inv_m, X[1] * sm * scale * inv_m
                ok = 0.5 * (yk + np.conj(ym)) + 0.5j * np.conj(w) * (yk - np.conj(ym))
                om = np.conj(0.5 * (yk + np.conj(ym)) - 0.5j * np.conj(w) * (yk - np.conj(ym)))
                ok[dc] = 0.5 * (yk[dc].real + ym[dc].real) + 0.5j * (yk[dc].real - ym[dc].real)
                om[dc] = ok[dc]
                buf[k2] = np.conj(ok)
                wr = ~(self_ & (k2 == k2m))
                buf[slot_b + k2m[wr]] = np.conj(om[wr])
        for s, k in enumerate(rows):
            ax[k] = buf[s * N2:(s + 1) * N2] * (rowtw[k][n2_ & 63] * rowtw[k][64 + (n2_ >> 6)])
    yq = colfft(ax, True).reshape(-1)
    y = np.empty(T)
    y[0::2], y[1::2] = yq.real, yq.imag
    return y, (N1, N2)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for T in [int(a) for a in sys.argv[1:]] or [16, 64, 120, 2 * 7 * 9, 4096, 2 * 2048 * 3, 2 * 2048 * 2, 9600, 44100]:
        x, h = rng.standard_normal(T), rng.standard_normal(T)
        ref = np.fft.irfft(np.fft.rfft(x) * np.fft.rfft(h), T) * 0.37
        y, plan = emulate(T, x, h, 0.37)
        print(T, plan, "rel err %.3g" % (np.abs(y - ref).max() / np.abs(ref).max()))
