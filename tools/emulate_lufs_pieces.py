#!/usr/bin/env python
"""Lane-level model of the hop-energy bookkeeping of kweight_hop_energy_dma (csrc/loudness.hip).

Checks, on the build machine (no GPU), that the piece logic -- cuts at hop boundaries and at the end of the
data, boundary lanes split at a wave-uniform offset, lanes strictly between two cuts taken whole -- assigns every
squared sample of a segment to exactly the right hop, for random (T, S, segment, warm-up) combinations including
several cuts per 2048-sample block, cuts on lane and block boundaries, and a truncated last hop.
usage: python tools/emulate_lufs_pieces.py [n_trials]
"""
import sys
import numpy as np

CHUNK, SB = 32, 2048


def kernel_model(y2, T, S, H_data, seg_hops, seg, warm):
    """y2: squares of the (already filtered) row, length >= T.  Returns {hop: energy} written by this segment."""
    h0 = seg * seg_hops
    h1 = min(h0 + seg_hops, H_data)
    n0 = h0 * S
    n1 = min(h1 * S, T)
    start = max(n0 - warm, 0) & ~3
    out = {}
    h = start // S
    hb = (h + 1) * S
    acc, acc_valid = 0.0, False
    lane = np.arange(64)
    sb = start
    while sb < n1:
        v2 = np.zeros(SB)
        hi = min(sb + SB, T)
        v2[: hi - sb] = y2[sb:hi]
        v2 = v2.reshape(64, CHUNK)
        e_full = v2.sum(1)
        sb_end = sb + SB
        c_prev, o_prev, carry = -1, 0, np.zeros(64)
        while True:
            cut = hb if hb < n1 else n1
            if cut >= sb_end:
                if h >= h0:
                    acc += (np.where(lane > c_prev, e_full, 0.0) + carry).sum()
                    acc_valid = True
                break
            p = cut - sb
            c, o = p >> 5, p & 31
            elo = v2[:, :o].sum(1)
            ehi = v2[:, o:].sum(1)
            if h >= h0:
                if c != c_prev:
                    contrib = carry + np.where((lane > c_prev) & (lane < c), e_full, 0.0) + np.where(lane == c, elo, 0.0)
                else:   # two cuts inside one lane's chunk (hop boundary, then the end of the data)
                    contrib = np.where(lane == c, v2[:, o_prev:o].sum(1), 0.0)
                acc += contrib.sum()
                acc_valid = True
            if cut == n1:
                break
            if acc_valid:
                out[h] = acc
                acc, acc_valid = 0.0, False
            h += 1
            hb += S
            carry = np.where(lane == c, ehi, 0.0)
            c_prev = c
            o_prev = o
        sb += SB
    if acc_valid:
        out[h] = acc
    return out


def main(n=None):
    if n is None:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial in range(n):
        S = int(rng.choice([800, 1600, 2048, 2205, 4410, 4800, 6400, 19200, 777, 4096]))
        T = int(rng.integers(4, 40 * S)) & ~3
        T = max(T, 4)
        H_data = (T + S - 1) // S
        sp = int(rng.integers(1, min(H_data, 7) + 1))
        seg_hops = (H_data + sp - 1) // sp
        segs = (H_data + seg_hops - 1) // seg_hops
        warm = int(rng.choice([0, 100, 3333, 5000]))
        y2 = rng.random(T + 4 * SB) ** 2
        y2[T:] = 1e6          # anything past the end must never be counted (the kernel zeroes v there; model it)
        y2m = y2.copy()
        y2m[T:] = 0.0
        got = {}
        for seg in range(segs):
            part = kernel_model(y2m, T, S, H_data, seg_hops, seg, warm)
            for k, v in part.items():
                assert k not in got, (trial, "hop written twice", k)
                got[k] = v
        for hcur in range(H_data):
            ref = y2[hcur * S: min((hcur + 1) * S, T)].sum()
            assert hcur in got, (trial, "hop missing", hcur, S, T, seg_hops)
            err = abs(got[hcur] - ref) / max(ref, 1e-30)
            worst = max(worst, err)
            assert err < 1e-9, (trial, hcur, got[hcur], ref)
        assert set(got) == set(range(H_data)), (trial, sorted(got)[-3:], H_data)
    print(f"{n} trials ok, worst relative deviation {worst:.2e}")


if __name__ == "__main__":
    main()
