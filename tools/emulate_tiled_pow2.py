"""CPU emulation of the LDS addressing of stft_tiled_pow2_kernel (csrc/stft_generic.hip): every thread's slots for the
sample stores, the three passes and the paired split step, for both plans, checked against numpy's rfft.  Also checks
that the slots a lane group touches in one LDS instruction are bank-conflict free (MI355X_MICROARCH.md, LDS table:
ds_read_b64 = 2 groups of 32 lanes over 64 dword banks; ds_write_b64 = 4 groups of 16 lanes over 32 dword banks)."""
import numpy as np


def swz(e):
    return e ^ ((e >> 4) & 15)


def check_read(slots, what, allow=0):
    s = np.asarray(slots).reshape(-1, 32)            # groups of 32 lanes
    for g in s:
        banks = (2 * g) % 64
        assert len(set(g.tolist())) - len(set(banks.tolist())) <= allow, (what, g)


def check_write(slots, what):
    s = np.asarray(slots).reshape(-1, 16)            # groups of 16 lanes, banks mod 32 dwords
    for g in s:
        banks = (2 * g) % 32
        assert len(set(banks.tolist())) == len(set(g.tolist())), (what, g)


def run(plan, seed=0):
    M = 2048 if plan == 1 else 4096
    FB = 2 if plan == 1 else 1
    R3 = 8 if plan == 1 else 16
    NB3 = 2 if plan == 1 else 1
    NPI = M // 512
    FS = M + M // 256
    N = 2 * M
    rng = np.random.default_rng(seed)
    frames = rng.standard_normal((FB, N))
    win = np.hanning(N + 1)[:N]
    z = (frames * win).reshape(FB, M, 2)
    z = z[..., 0] + 1j * z[..., 1]                   # z[n] = x[2n] + i x[2n+1]
    buf = np.zeros(4096 + 16, dtype=complex)
    T = np.arange(256)
    st = swz(T)
    rq = (lambda q: 128 * q + (q >> 1)) if plan == 1 else (lambda q: 257 * q)
    # sample stores: e = t + 256 i
    for i in range(16):
        e = T + 256 * i
        fi = (i >> 3) if plan == 1 else 0
        slots = st + 257 * i
        check_write(slots, "samples")
        assert np.all(slots == swz(e) + (e >> 8))
        buf[slots] = z[fi, e - fi * M]
    fj = (T >> 7) if plan == 1 else 0 * T
    j = (T & 127) if plan == 1 else T
    jl = j & 15
    rb0 = fj * FS + (j ^ ((j >> 4) & 15))
    rb1 = fj * FS + (j ^ ((j >> 4) | 8)) if plan == 1 else rb0
    wb1 = fj * FS + 16 * j + (j >> 4)
    wb2 = fj * FS + 257 * (j >> 4)
    tw2 = np.zeros(240, dtype=complex)
    for idx in range(240):
        q1, k = idx >> 4, idx & 15
        tw2[idx] = np.exp(-2j * np.pi * (k * (q1 + 1) * (M // 256)) / M)
    tw3 = np.zeros(257 * (R3 - 1), dtype=complex)
    for idx in range(256 * (R3 - 1)):
        q1, k = idx >> 8, idx & 255
        tw3[257 * q1 + k] = np.exp(-2j * np.pi * (k * (q1 + 1)) / M)
    dft = lambda v: np.fft.fft(v, axis=0)
    # pass 1
    v = np.zeros((16, 256), dtype=complex)
    for q in range(16):
        slots = (rb1 if (q & 1) else rb0) + rq(q)
        check_read(slots, "p1 read")
        e = fj * M + j + (M // 16) * q
        assert np.all(slots == swz(e) + (e >> 8)), "p1 read addr"
        v[q] = buf[slots]
    v = dft(v)
    for q in range(16):
        slots = wb1 + (q ^ jl)
        check_write(slots, "p1 write")
        e = fj * M + 16 * j + q
        assert np.all(slots == swz(e) + (e >> 8)), "p1 write addr"
        buf[slots] = v[q]
    # pass 2
    for q in range(16):
        v[q] = buf[(rb1 if (q & 1) else rb0) + rq(q)]
    for q in range(1, 16):
        v[q] = v[q] * tw2[(q - 1) * 16 + jl]
    v = dft(v)
    for q in range(16):
        slots = wb2 + 16 * q + (jl ^ q)
        check_write(slots, "p2 write")
        e = fj * M + 256 * (j >> 4) + 16 * q + jl
        assert np.all(slots == swz(e) + (e >> 8)), "p2 write addr"
        buf[slots] = v[q]
    # pass 3
    for b in range(NB3):
        w = np.zeros((R3, 256), dtype=complex)
        for q in range(R3):
            slots = st + FS * b + 257 * q
            check_read(slots, "p3 read")
            e = M * b + T + 256 * q
            assert np.all(slots == swz(e) + (e >> 8)), "p3 addr"
            w[q] = buf[slots]
        for q in range(1, R3):
            w[q] = w[q] * tw3[(q - 1) * 257 + T]
        w = dft(w)
        for q in range(R3):
            buf[st + FS * b + 257 * q] = w[q]
    # the transform so far
    for fi in range(FB):
        e = fi * M + np.arange(M)
        Z = buf[swz(e) + (e >> 8)]
        assert np.allclose(Z, np.fft.fft(z[fi]), atol=1e-9), "Z"
    # split
    tm = (256 - T) & 255
    sm = swz(tm) + np.where(T == 0, 257, 0)
    stw = np.exp(-2j * np.pi * np.arange(M // 2 + 1) / N)
    X = np.zeros((FB, M + 1), dtype=complex)
    cnt = np.zeros((FB, M + 1), dtype=int)

    def split(zk, zm, w):
        c, sn = w.real, -w.imag
        sr, si = zk.real + zm.real, zk.imag - zm.imag
        dr, di = zk.real - zm.real, zk.imag + zm.imag
        pp = sn * dr - c * di
        qq = sn * di + c * dr
        return 0.5 * (sr - pp) + 0.5j * (si - qq), 0.5 * (sr + pp) + 0.5j * (-si - qq)

    for fi in range(FB):
        for i in range(NPI):
            zk_slots = fi * FS + st + 257 * i
            check_read(zk_slots, "split zk")
            am = fi * FS + sm + 257 * (2 * NPI - 1 - i)
            if i == 0:
                am = np.where(T == 0, fi * FS, am)
            check_read(am, "split zm", allow=1)        # the descending sweep crosses one pad: one 2-way conflict per group
            k = T + 256 * i
            mk = (M - k) % M
            e = fi * M + mk
            assert np.all(am == swz(e) + (e >> 8)), ("zm addr", i)
            xa, xb = split(buf[zk_slots], buf[am], stw[k])
            X[fi, k] = xa; cnt[fi, k] += 1
            hi = (M - 256 * i - 255) + (255 - T)
            assert np.all(hi == M - k)
            X[fi, hi] = xb; cnt[fi, hi] += 1
        zs = buf[fi * FS + M // 2 + M // 512]
        xa, _ = split(zs, zs, stw[M // 2])
        X[fi, M // 2] = xa; cnt[fi, M // 2] += 1
        assert np.all(cnt[fi] == 1)
        assert np.allclose(X[fi], np.fft.rfft(frames[fi] * win), atol=1e-9), "X"
    print("plan", plan, "ok")


if __name__ == "__main__":
    run(1)
    run(2)


def mel_reduction(sr, n_fft, n_mels, FB, NPC, seed=0):
    """The lane-level segmented sum of the mel epilogue: tasks T = t + 256 s (task = frame * n_chunks + chunk), flags as
    the kernel builds them, six doubling steps inside 64-lane waves, head lanes store one of two pieces per band."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from audiotools_amd import tables
    basis = tables.mel_filters_np(sr, n_fft, n_mels)
    info, w = tables.mel_bands_np(basis)
    nch = w.shape[0]
    bands = info[nch:].reshape(n_mels, 2)
    cb = np.zeros(nch, dtype=int)
    for m in range(n_mels):
        cb[bands[m, 0]: bands[m, 0] + bands[m, 1]] = m
    ntask = FB * nch
    assert ntask <= 768
    rng = np.random.default_rng(seed)
    vals = rng.standard_normal(ntask)
    part = np.zeros((FB, n_mels, NPC))
    writes = np.zeros((FB, n_mels, NPC), dtype=int)

    def same(T, T2):
        if T2 < 0 or T2 >= ntask:
            return False
        f, f2 = T // nch, T2 // nch
        return f == f2 and cb[T2 - f2 * nch] == cb[T - f * nch]

    for s in range(3):
        for wave in range(4):
            T0 = 256 * s + 64 * wave
            acc = np.array([vals[T0 + l] if T0 + l < ntask else 0.0 for l in range(64)])
            fl = [[(T0 + l < ntask) and l + (1 << i) < 64 and same(T0 + l, T0 + l + (1 << i)) for i in range(6)] for l in range(64)]
            for i in range(6):
                sh = np.concatenate([acc[1 << i:], acc[-(1 << i):]])      # out-of-range lanes: any value, masked by the flag
                acc = acc + np.where([fl[l][i] for l in range(64)], sh, 0.0)
            for l in range(64):
                T = T0 + l
                if T >= ntask:
                    continue
                cont = same(T, T - 1)
                if l == 0 or not cont:
                    f, c = T // nch, T % nch
                    piece = (T >> 6) - ((f * nch + bands[cb[c], 0]) >> 6)
                    assert piece < NPC
                    part[f, cb[c], piece] = acc[l]
                    writes[f, cb[c], piece] += 1
    assert writes.max() <= 1
    for f in range(FB):
        for m in range(n_mels):
            c0, cn = bands[m]
            ref = vals[f * nch + c0: f * nch + c0 + cn].sum()
            assert abs(part[f, m].sum() - ref) < 1e-9, (f, m)
    print("mel reduction", sr, n_fft, n_mels, "chunks", nch, "ok")


if __name__ == "__main__":
    mel_reduction(96000, 4096, 80, 2, 4)
    mel_reduction(96000, 4096, 128, 2, 4)
    mel_reduction(96000, 4096, 7, 2, 4)
    mel_reduction(96000, 4096, 2, 2, 4)
    mel_reduction(192000, 8192, 80, 1, 8)
    mel_reduction(192000, 8192, 128, 1, 8)
    mel_reduction(192000, 8192, 3, 1, 8)
