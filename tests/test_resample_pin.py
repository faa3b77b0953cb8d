"""An INDEPENDENT float64 evaluation of the resampling kernel behind ``AudioSignal.resample``
(reference: audiotools/core/audio_signal.py:716-736 -> julius.resample_frac, upstream julius 0.2.7, absent from
/root/reference), written from the published description by a separate derivation -- VERDICT r05 "what's missing" #3.

The product bank (``audiotools_amd.tables.resample_bank``) and the oracle leaf (``oracle/leaves/julius_leaf.py``) follow the
upstream code line by line in float32.  This file does not: it states the filter as mathematics and evaluates it with exact
rational time offsets.

    Resampling old -> new (reduced fraction) evaluates, for output sample m, the input at time u = m * old / new (in input
    samples) with the interpolation kernel

        h(tau) = sinc(tau) * (1 + cos(pi * tau / Z)) / 2        for |tau| <= Z,   0 beyond          (Z = 24 zero crossings)

    of the band limit  fc = rolloff * min(old, new) / 2  (rolloff = 0.945), i.e. tau = (n - u) * 2 fc / old ... with rates in
    units where the input rate is `old`:  tau = (n - u) * rolloff * min(old, new) / old.  (cos^2(x / 2) = (1 + cos x) / 2: the
    upstream "cos(t / zeros / 2) ** 2" window is the Hann taper over the kernel's 2 Z lobes.)  Output phase i = m mod new reads
    the taps j = n - (m // new) * old in [-W, W + old), W = ceil(Z * old / (rolloff * min)), and every phase is normalised to
    unit DC gain.

It is still ONE source (the published julius description); what the test adds is that a transcription slip in window, roll-off,
width, phase indexing or normalisation on either side would now show: the two evaluations share no code and no arithmetic
(float32 torch vs exact fractions + float64 numpy).
"""
from fractions import Fraction
import math

import numpy as np
import pytest

Z = 24
ROLLOFF = Fraction(945, 1000)


def independent_bank(old_sr: int, new_sr: int) -> np.ndarray:
    g = math.gcd(old_sr, new_sr)
    old, new = old_sr // g, new_sr // g
    band = ROLLOFF * min(old, new)                       # 2 fc, in units where the input rate is `old`
    W = math.ceil(Z * old / band)                        # taps that can lie within Z zero crossings on either side
    js = range(-W, W + old)
    bank = np.zeros((new, len(js)), dtype=np.float64)
    for i in range(new):
        for c, j in enumerate(js):
            tau = Fraction(j * new - i * old, new * old) * band       # (n - u) * band / old, exact
            if abs(tau) >= Z:
                continue                                               # outside the kernel's support (h(+-Z) = 0)
            x = float(tau)
            bank[i, c] = np.sinc(x) * 0.5 * (1.0 + math.cos(math.pi * x / Z))
        bank[i] /= bank[i].sum()                                       # unit DC gain per output phase
    return bank, old, new, W


@pytest.mark.parametrize("old_sr, new_sr", [(44100, 16000), (48000, 44100), (16000, 44100), (44100, 22050), (8000, 48000)])
def test_product_and_oracle_banks_equal_the_independent_evaluation(old_sr, new_sr):
    from audiotools_amd import tables
    from oracle.leaves import julius_leaf

    want, old, new, W = independent_bank(old_sr, new_sr)
    bank, o, n, width = tables.resample_bank(old_sr, new_sr)
    assert (o, n, width) == (old, new, W)
    got = bank.numpy().astype(np.float64)
    assert got.shape == want.shape
    # Tolerance: upstream evaluates t = (j / old - i / new) * sr * pi in FLOAT32 -- a difference of two O(1) numbers scaled by
    # ~150 pi, i.e. ~5e-5 of absolute phase error -- so its taps (peak ~0.35) sit up to ~1e-5 off the exact ones; product
    # and oracle reproduce that float32 evaluation (they equal each other to the last bit, tests/test_oracle_vs_reference.py).
    # A transcription slip (window form, roll-off, zero count, width, phase sign, normalisation) moves taps by 1e-3 or more.
    tol = 4e-5          # (upsampling kernels have taps of ~1: the same phase error weighs more; measured up to 1.1e-5)
    assert np.abs(got - want).max() < tol, float(np.abs(got - want).max())
    leaf = julius_leaf.ResampleFrac(old_sr, new_sr).kernel.reshape(new, -1).numpy().astype(np.float64)
    assert leaf.shape == want.shape and np.abs(leaf - want).max() < tol
    # ... and the float64 evaluation of the SAME upstream expression closes the gap to rounding: what is left above is float32
    j = np.arange(-W, W + old, dtype=np.float64)
    t = np.clip((j[None, :] / old - np.arange(new, dtype=np.float64)[:, None] / new) * float(ROLLOFF * min(old, new)), -Z, Z) * np.pi
    k64 = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t)) * np.cos(t / Z / 2) ** 2
    k64 /= k64.sum(1, keepdims=True)
    assert np.abs(k64 - want).max() < 1e-12, float(np.abs(k64 - want).max())


def test_independent_bank_has_the_published_frequency_response():
    """What the numbers must do regardless of who wrote them down: the 44.1 -> 16 kHz kernel passes 0.8 of the new Nyquist
    within 0.01 dB, is 6 dB down where the band limit sits (rolloff x Nyquist), and is more than 60 dB down 15 % above the new
    Nyquist.  Evaluated on the interleaved prototype (all phases together = the kernel sampled at old * new)."""
    bank, old, new, W = independent_bank(44100, 16000)
    # prototype sampled at rate old * new (per unit input rate `old`): h_proto[j * new - i * old] = bank[i, j]; every phase
    # sums to 1, so the prototype's DC gain is `new`
    taps = {}
    for i in range(new):
        for c, j in enumerate(range(-W, W + old)):
            taps[j * new - i * old] = bank[i, c]
    ks = np.array(sorted(taps))
    h = np.array([taps[k] for k in ks])
    fs = old * new                                          # prototype rate in units of 1 / (input sample)
    def gain_db(f_rel_new_nyquist):
        f = f_rel_new_nyquist * (new / 2) / old * new       # cycles per prototype-rate second ... in cycles per unit
        w = 2 * np.pi * (f_rel_new_nyquist * (new / 2.0) / old) / new
        H = np.sum(h * np.exp(-1j * w * ks)) / new
        return 20 * np.log10(abs(H))
    assert abs(gain_db(0.0)) < 1e-9
    assert abs(gain_db(0.8)) < 0.01
    assert abs(gain_db(0.945) + 6.02) < 0.1
    assert gain_db(1.15) < -60.0
