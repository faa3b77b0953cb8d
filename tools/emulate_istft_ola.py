#!/usr/bin/env python
"""Tile-level numpy model of ``istft_generic_ola_kernel`` (audiotools_amd/csrc/stft_generic.hip): the one-pass inverse of the
run-time transform sizes.  Everything the kernel does with INDICES -- the plan (frames per tile FB, history depth R), runs
of consecutive tiles per workgroup with a warm-up tile when a run starts inside a row, the R + FB frame slots (history in
front of the tile), zero spectra outside [0, n_frames), the pair-wise gather of a tile's FB hops in ascending frame order,
the 1 / envelope table and its bounds, the centre trim p = 2 j - n_fft / 2 and the store predicates -- with numpy's c2r
transform standing in for the fold + in-place passes (those are the forward tile's, tools/emulate_tiled_pow2.py and the
butterfly tests cover them).  ``python tools/emulate_istft_ola.py`` prints the error against torch.istft;
tests/test_host_logic.py runs it on the CPU.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TILE_POINTS = 4096


def smooth(m):
    for r in (2, 3, 5, 7):
        while m % r == 0:
            m //= r
    return m == 1


def plan(n_fft, hop):
    """generic_ola_plan: (FB, R) or None when the shape keeps the frame buffer + gather path."""
    if n_fft < 8 or n_fft % 4 or hop % 2 or hop <= 0 or hop > n_fft:
        return None
    M = n_fft // 2
    if M > TILE_POINTS or not smooth(M):
        return None
    R = (n_fft + hop - 1) // hop - 1
    fb = min(TILE_POINTS // M, 64)
    while fb > R and fb > 2 and (R + fb + 3) * M * 8 > 80 * 1024:
        fb -= 1
    if fb < R or fb < 1 or (R + fb + 3) * M * 8 > 160 * 1024 or R > 16:
        return None
    return fb, R


def envelope(window, n_frames, hop):
    """istft_env_generic_kernel: 1 / sum_f w^2 over the frames that cover a position, 0 where it vanishes."""
    N = len(window)
    total = (n_frames - 1) * hop + N
    env = np.zeros(total, dtype=np.float32)
    w2 = (window.astype(np.float32) ** 2).astype(np.float32)
    for f in range(n_frames):
        env[f * hop: f * hop + N] += w2
    with np.errstate(divide="ignore"):
        return np.where(env > 1e-11, np.float32(1) / env, np.float32(0)).astype(np.float32)


def istft(X, window, hop, length, slots_wg=3, force_runs=None):
    """X (rows, n_frames, M + 1) complex64 -> (rows, length) float32, or None when the kernel does not take the shape."""
    rows, n_frames, F = X.shape
    M = F - 1
    n_fft = 2 * M
    p = plan(n_fft, hop)
    if p is None:
        return None
    FB, R = p
    h2 = hop // 2
    inv_env = envelope(window, n_frames, hop)
    env_n = len(inv_env)
    win2 = np.stack([window[0::2], -window[1::2]], -1).astype(np.float32) / np.float32(n_fft)     # (M, 2): sign of the conj trick
    out = np.full((rows, length), np.nan, dtype=np.float32)
    need_pairs = (length + M + 1) // 2
    per_tile = FB * h2
    tiles_per_row = -(-need_pairs // per_tile)
    # runs: >= 4 per workgroup slot, >= 8 tiles each
    k = -(-4 * slots_wg // rows) if force_runs is None else force_runs
    k = max(1, min(k, max(tiles_per_row // 8, 1))) if force_runs is None else max(1, min(k, tiles_per_row))
    tiles_per_run = -(-tiles_per_row // k)
    runs_per_row = -(-tiles_per_row // tiles_per_run)

    def transform(row, f0):
        """the FB frames from f0 as pairs (re = sample 2 n, im = -(sample 2 n + 1)) * n_fft: what the passes leave in LDS"""
        buf = np.zeros((FB, M, 2), dtype=np.float32)
        for fi in range(FB):
            if f0 + fi < n_frames:
                spec = X[row, f0 + fi].astype(np.complex64).copy()
                spec[0] = spec[0].real
                spec[M] = spec[M].real                      # c2r ignores the imaginary part of DC and Nyquist
                y = np.fft.irfft(spec.astype(np.complex128), n_fft) * n_fft
                buf[fi, :, 0] = y[0::2]
                buf[fi, :, 1] = -y[1::2]
        return buf

    for run in range(rows * runs_per_row):
        row, r = divmod(run, runs_per_row)
        t_first = r * tiles_per_run
        t_last = min(t_first + tiles_per_run, tiles_per_row)
        slots = np.zeros((R + FB, M, 2), dtype=np.float32)
        if t_first > 0:
            slots[R:] = transform(row, (t_first - 1) * FB)
            slots[:R] = slots[FB: FB + R].copy()            # keep_history (FB >= R)
        for t in range(t_first, t_last):
            f0 = t * FB
            slots[R:] = transform(row, f0)
            for jj in range(per_tile):
                q, rem = divmod(jj, h2)
                acc = np.zeros(2, dtype=np.float32)
                for kk in range(R, -1, -1):                 # ascending frame order
                    n2 = rem + kk * h2
                    if n2 < M:
                        acc += slots[R + q - kk, n2] * win2[n2]
                j = f0 * h2 + jj
                p0 = 2 * j - M
                for e in range(2):
                    pe = p0 + e
                    if 0 <= pe < length:
                        assert np.isnan(out[row, pe]), "an output sample was written twice"
                        out[row, pe] = acc[e] * (inv_env[2 * j + e] if 2 * j + e < env_n else np.float32(0))
            slots[:R] = slots[FB: FB + R].copy()
    assert not np.isnan(out).any(), "an output sample was never written"
    return out


if __name__ == "__main__":
    import torch

    rng = np.random.default_rng(0)
    for n_fft, hop, T, rows in [(400, 160, 3203, 2), (400, 100, 2000, 1), (1200, 300, 6000, 2), (1920, 480, 24001, 1), (400, 400, 2000, 1),
                                (512, 100, 3000, 2), (320, 40, 1500, 1)]:
        n_frames = 1 + T // hop
        X = (rng.standard_normal((rows, n_frames, n_fft // 2 + 1)) + 1j * rng.standard_normal((rows, n_frames, n_fft // 2 + 1))).astype(np.complex64)
        win = np.hanning(n_fft + 1)[:-1].astype(np.float32) if n_fft % hop == 0 and n_fft // hop >= 2 else np.ones(n_fft, np.float32)
        got = istft(X, win, hop, T, force_runs=3)
        ref = torch.istft(torch.from_numpy(X).transpose(1, 2), n_fft, hop, window=torch.from_numpy(win), center=True, length=T).numpy()
        print(f"n_fft {n_fft:5d} hop {hop:4d} T {T:6d}: plan (FB, R) = {plan(n_fft, hop)}, max |model - torch.istft| / max|ref| = "
              f"{np.abs(got - ref).max() / np.abs(ref).max():.2e}")
