#!/usr/bin/env python
"""Lane-level numpy model of ``resample_f16s_kernel`` (audiotools_amd/csrc/resample_f16.hip).

Everything the kernel does to the numbers and to the indices, on whole (rows, T) signals:
  * tiles of 16 frames, contiguous runs of tiles per workgroup, the geometry of a tile (first sample gx = 16 tile old -
    width, the shift 0..3 that makes the DMA source 16-byte aligned for the row's base address, n4 float4 pieces);
  * the prefetch: NLD pieces per thread with the float4 index clamped into the row; tiles that reach over an end of the
    row are re-read element by element with replicate padding when they are staged;
  * per tile: maximum -> power-of-two scale from the exponent field (clamped so that s and 1/s are normal), every
    sample split IN PLACE into (hi | lo << 16) with hi = RN16(x s), lo = RN16(x s - hi);
  * operands: lane (frame i = lane % 16, k-group g = lane // 16) reads dwords shift + i old + lo[P] + koff[g] + 32 c + 0..7,
    v_perm_b32 separates the halves; B operands are decoded from tables.resample_f16_bank exactly as the kernel loads them;
  * three products per chunk, fp16 x fp16 exact, accumulated in float32 (hh in one accumulator, hl + lh in another);
  * the store mask (phase < new, output index < out_len).
``python tools/emulate_resample_f16s.py`` prints the error against float64 next to plain float32's.
tests/test_host_logic.py runs it against the oracle (framing, edges, lengths) on the CPU.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiotools_amd import tables  # noqa: E402

KOFF = np.asarray(tables.MFMA_KOFF)


def halves(dwords):
    """uint32 (..., 4) -> float32 (..., 8): the eight fp16 of a B / A operand register quad."""
    u = np.stack([dwords & 0xffff, dwords >> 16], -1).reshape(dwords.shape[:-1] + (8,)).astype(np.uint16)
    return u.view(np.float16).astype(np.float32)


def split_pack(e):
    h = e.astype(np.float16)
    r = (e - h.astype(np.float32)).astype(np.float32)
    lo = r.astype(np.float16)
    return h.view(np.uint16).astype(np.uint32) | (lo.view(np.uint16).astype(np.uint32) << 16)


def resample(x, old_sr, new_sr, base_word=0, n_wg=3, form="rp"):
    """x (rows, T) float32 -> (rows, floor(new T / old)); ``base_word``: address of x[0, 0] in floats (alignment of the
    rows); ``n_wg``: workgroups the tiles are dealt to in contiguous runs (exercises the run boundaries); ``form``: how the
    tiles that reach over an end of their row get their replicate padding -- "dma": element-wise re-read (resample_f16s_kernel),
    "rp": fix-up in registers from six scalar loads (resample_f16s_rp_kernel).  Everything else is common to both kernels."""
    W, lo, old, new, width, NPB, NC, wk = tables.resample_f16_bank(old_sr, new_sr)
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, T = x.shape
    assert T >= 16
    out_len = new * T // old
    out = np.full((rows, out_len), np.nan, dtype=np.float32)
    frames = (out_len + new - 1) // new
    tiles_per_row = (frames + 15) // 16
    n_tiles = rows * tiles_per_row
    max_lo = int(lo.max())
    need = 15 * old + max_lo + 32 * NC
    threads = NPB * 64
    NLD = ((need + 3 + 3) // 4 + threads - 1) // threads
    assert NLD <= 4
    inv_wscale = np.float32(2.0 ** -wk)
    Bh = halves(W[:, :, 0])          # (NPB, NC, 64, 8)
    Bl = halves(W[:, :, 1])
    lane = np.arange(64)
    j, g = lane % 16, lane // 16
    t_idx = np.arange(threads)
    flat = x.reshape(-1)

    def geom(r, tl):
        gx = tl * 16 * old - width
        word = base_word + r * T + gx
        shift = word & 3
        a0 = gx - shift
        n4 = (need + shift + 3) >> 2
        edge = not (a0 >= 0 and a0 + 4 * n4 <= T)
        return a0, n4, shift, edge

    def stage(r, tl):
        """the LDS image of the tile (dwords) and 1 / s"""
        a0, n4, shift, edge = geom(r, tl)
        buf = np.zeros(4 * NLD * threads, dtype=np.float32)
        # DMA: float4 index clamped into the row
        q_lo = 0 if a0 >= 0 else (-a0 + 3) >> 2
        q_hi = min(((T - a0) >> 2) - 1, n4 - 1)
        q_lo = min(q_lo, q_hi)
        for l in range(NLD):
            q = np.clip(t_idx + l * threads, q_lo, q_hi)
            src = r * T + a0 + 4 * q
            assert src.min() >= r * T and src.max() + 3 < (r + 1) * T, "DMA outside the row"
            assert ((base_word + src) % 4 == 0).all(), "DMA source not 16-byte aligned"
            for e in range(4):
                buf[4 * (t_idx + l * threads) + e] = flat[src + e]
        if edge and form == "dma":
            d_lo, d_hi = (-a0 if a0 < 0 else 0), T - 1 - a0
            for l in range(NLD):
                q = np.minimum(t_idx + l * threads, n4 - 1)
                for e in range(4):
                    d = np.clip(4 * q + e, d_lo, d_hi)
                    buf[4 * (t_idx + l * threads) + e] = flat[r * T + a0 + d]
        elif edge:
            # register-prefetch form: lanes whose float4 index was clamped take one of x[0..2] / x[T-3..T-1] (scalar loads)
            xs, xe = x[r, :3], x[r, T - 3:]
            d_lo, d_hi = (-a0 if a0 < 0 else 0), T - 1 - a0
            for l in range(NLD):
                q = np.minimum(t_idx + l * threads, n4 - 1)
                for e in range(4):
                    d = 4 * q + e
                    i_lo, i_hi = d - d_lo, d - d_hi
                    v_lo = np.where(i_lo <= 0, xs[0], np.where(i_lo == 1, xs[1], xs[2]))
                    v_hi = np.where(i_hi >= 0, xe[2], np.where(i_hi == -1, xe[1], xe[0]))
                    v = buf[4 * (t_idx + l * threads) + e]
                    buf[4 * (t_idx + l * threads) + e] = np.where(q < q_lo, v_lo, np.where(q > q_hi, v_hi, v))
        # what the image must hold wherever an operand read can land
        d_all = np.arange(need + shift)
        want = x[r, np.clip(a0 + d_all, 0, T - 1)]
        assert np.array_equal(buf[:need + shift], want), "LDS image differs from the replicate-padded row"
        ab = np.abs(buf)
        tm = np.float32(np.where(np.isfinite(ab), ab, 0).max())           # maximum of the FINITE samples
        field = 268 - int(np.float32(tm).view(np.uint32) >> 23)
        field = min(max(field, 1), 253)
        s = np.array([field << 23], dtype=np.uint32).view(np.float32)[0]
        inv = np.array([(254 - field) << 23], dtype=np.uint32).view(np.float32)[0]
        with np.errstate(over="ignore", invalid="ignore"):
            img = split_pack((buf * s).astype(np.float32))
        return img, inv, shift

    blocks = min(n_wg, n_tiles)
    tiles_per_wg = (n_tiles + blocks - 1) // blocks
    for wg in range((n_tiles + tiles_per_wg - 1) // tiles_per_wg):
        for tile_id in range(wg * tiles_per_wg, min((wg + 1) * tiles_per_wg, n_tiles)):
            r, tl = divmod(tile_id, tiles_per_row)
            img, inv, shift = stage(r, tl)
            for P in range(NPB):
                acc_m = np.zeros((16, 16), dtype=np.float32)
                acc_c = np.zeros((16, 16), dtype=np.float32)
                a_base = shift + j * old + int(lo[P]) + KOFF[g]
                for c in range(NC):
                    d = img[(a_base + 32 * c)[:, None] + np.arange(8)[None, :]]        # (64 lanes, 8 dwords)
                    xh = (d & 0xffff).astype(np.uint16).view(np.float16).astype(np.float32)
                    xl = (d >> 16).astype(np.uint16).view(np.float16).astype(np.float32)
                    # MFMA: D[i, jj] += sum over (g, s) A[lane (i, g)][s] B[lane (jj, g)][s]
                    Ah = np.zeros((16, 32), dtype=np.float32)
                    Al = np.zeros((16, 32), dtype=np.float32)
                    Bhm = np.zeros((32, 16), dtype=np.float32)
                    Blm = np.zeros((32, 16), dtype=np.float32)
                    for ln in range(64):
                        Ah[j[ln], 8 * g[ln]: 8 * g[ln] + 8] = xh[ln]
                        Al[j[ln], 8 * g[ln]: 8 * g[ln] + 8] = xl[ln]
                        Bhm[8 * g[ln]: 8 * g[ln] + 8, j[ln]] = Bh[P, c, ln]
                        Blm[8 * g[ln]: 8 * g[ln] + 8, j[ln]] = Bl[P, c, ln]
                    with np.errstate(over="ignore", invalid="ignore"):
                        acc_m = (acc_m + Ah @ Bhm).astype(np.float32)
                        acc_c = (acc_c + (Ah @ Blm + Al @ Bhm)).astype(np.float32)
                with np.errstate(over="ignore", invalid="ignore"):
                    y = ((acc_m + acc_c) * np.float32(inv * inv_wscale)).astype(np.float32)
                for i in range(16):
                    for jj in range(16):
                        ph = 16 * P + jj
                        o = (tl * 16 + i) * new + ph
                        if ph < new and o < out_len:
                            out[r, o] = y[i, jj]
    assert not np.isnan(out).any() or np.isnan(x).any() or np.isinf(x).any(), "an output was never written"
    return out


def reference(x, old_sr, new_sr, dtype=np.float64):
    bank, old, new, width = tables.resample_bank(old_sr, new_sr)
    b = bank.numpy().astype(dtype)
    taps = b.shape[1]
    outs = []
    for row in np.atleast_2d(x):
        xp = np.pad(row.astype(dtype), (width, width + old), mode="edge")
        n_frames = (len(xp) - taps) // old + 1
        fr = np.stack([xp[f * old: f * old + taps] for f in range(n_frames)])
        outs.append((fr @ b.T).reshape(-1)[: new * len(row) // old])
    return np.stack(outs)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    n = 30000
    t = np.arange(n) / 44100
    cases = {"white 0.1": 0.1 * rng.standard_normal(n), "sine 440 Hz 0.9": 0.9 * np.sin(2 * np.pi * 440 * t),
             "white 1e-4": 1e-4 * rng.standard_normal(n),
             "tone + noise at -100 dB": 0.5 * np.sin(2 * np.pi * 1000 * t) + 1e-5 * rng.standard_normal(n),
             "unclipped (|x| up to 5)": 1.5 * rng.standard_normal(n),
             "white 1e-8": 1e-8 * rng.standard_normal(n), "white 1e+6": 1e6 * rng.standard_normal(n),
             "burst: 1e-5 then 1.0": np.concatenate([1e-5 * rng.standard_normal(n // 2), rng.standard_normal(n - n // 2)])}
    for name, x in cases.items():
        x = x.astype(np.float32)[None]
        ref = reference(x, 44100, 16000)
        f32 = reference(x, 44100, 16000, np.float32).astype(np.float64)
        got = resample(x, 44100, 16000).astype(np.float64)
        m = np.abs(ref).max()
        print(f"{name:26s} fp32 {np.abs(f32 - ref).max() / m:.2e}   fp16 split {np.abs(got - ref).max() / m:.2e}")
