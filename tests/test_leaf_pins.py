"""Independent pins for the third-party leaves the reference delegates to (the
reference's own tests check shapes only, and the packages -- librosa, torchaudio, pyloudnorm,
julius -- are not installed here).  oracle/leaves/* and audiotools_amd/tables.py are two
restatements by the same author, so each is compared here with something that author did not write:

* the Slaney mel filterbank with HuggingFace ``transformers.audio_utils.mel_filter_bank`` (an
  independent implementation of librosa.filters.mel(norm="slaney", htk=False));
* the MFCC basis (torchaudio.functional.create_dct) with ``scipy.fft.dct(type=2, norm="ortho")``;
* the K-weighting design (pyloudnorm's RBJ-style shelf + high-pass) with the coefficient table
  printed in ITU-R BS.1770 for 48 kHz and with its frequency response;
* the periodic Hann window with its closed form;
* the julius windowed-sinc resampler with properties no restatement can fake: a band-limited sine
  keeps amplitude and phase through 44.1 -> 16 kHz and back up, and an integer-ratio resample agrees
  with ``scipy.signal.resample_poly`` run with the SAME kernel (checks the polyphase indexing, not the
  kernel formula);
* the julius low-pass design (low_pass / high_pass / equalizer) with ``scipy.signal.firwin``, and the
  HTK band edges of the equaliser with transformers' mel conversions;
* the BS.1770 meter as a whole with the expected readings EBU Tech 3341 publishes for its test signals.
Runs on CPU; the transformers check is skipped when that package is missing."""
import math

import numpy as np
import pytest
import torch

from audiotools_amd import tables
from oracle import restate
from oracle.leaves import misc_leaves, pyloudnorm_leaf

def _hf_audio_utils():
    """transformers.audio_utils, imported with the oracle's leaf shims (spec-less stand-ins for
    torchaudio, librosa, ... that oracle/ref_import.py may have put into sys.modules in this test
    session) out of the way: transformers probes optional packages with importlib.util.find_spec,
    which rejects a module without a __spec__."""
    import sys
    hidden = {k: sys.modules.pop(k) for k in list(sys.modules)
              if k != "__main__" and getattr(sys.modules[k], "__spec__", None) is None
              and getattr(sys.modules[k], "__file__", None) is None}
    try:
        au = pytest.importorskip("transformers.audio_utils")
    finally:
        for k, v in hidden.items():
            sys.modules.setdefault(k, v)
    return au


MEL_CASES = [(44100, 2048, 80, 0.0, None), (16000, 512, 40, 0.0, None), (48000, 2048, 128, 20.0, 16000.0),
             (22050, 1024, 64, 0.0, 8000.0)]


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", MEL_CASES)
def test_mel_basis_vs_transformers(sr, n_fft, n_mels, fmin, fmax):
    au = _hf_audio_utils()
    hf = au.mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin,
                            max_frequency=fmax if fmax is not None else sr / 2, sampling_rate=sr, norm="slaney",
                            mel_scale="slaney").T
    ours = tables.mel_filters_np(sr, n_fft, n_mels, fmin, fmax)
    leaf = misc_leaves.librosa_mel(sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax)
    scale = float(np.abs(hf).max())
    assert ours.shape == hf.shape == leaf.shape == (n_mels, n_fft // 2 + 1)
    assert float(np.abs(ours - hf).max()) < 1e-6 * scale       # float32 rounding of the product table
    assert float(np.abs(np.asarray(leaf, dtype=np.float64) - hf).max()) < 1e-6 * scale
    # the same support: a weight is zero in one exactly where it is (numerically) zero in the other
    assert np.array_equal(np.abs(hf) > 1e-9 * scale, np.abs(np.asarray(ours)) > 1e-9 * scale)


@pytest.mark.parametrize("n_mfcc,n_mels", [(20, 80), (13, 40), (40, 128), (80, 80)])
def test_dct_basis_vs_scipy(n_mfcc, n_mels):
    import scipy.fft
    want = scipy.fft.dct(np.eye(n_mels), type=2, norm="ortho", axis=0)[:n_mfcc].T      # (n_mels, n_mfcc)
    leaf = misc_leaves.create_dct(n_mfcc, n_mels, "ortho").numpy()
    ours = tables.dct_np(n_mfcc, n_mels, "ortho")
    assert leaf.shape == want.shape == ours.shape
    # torchaudio evaluates cos(pi / n_mels * (n + 0.5) * k) in float32 (arguments up to ~250 rad): 3e-6
    assert float(np.abs(leaf - want).max()) < 5e-6
    assert float(np.abs(ours - want).max()) < 5e-6


# ITU-R BS.1770-4, Annex 1, tables 1 and 2 (48 kHz)
ITU_SHELF_B = np.array([1.53512485958697, -2.69169618940638, 1.19839281085285])
ITU_SHELF_A = np.array([1.0, -1.69065929318241, 0.73248077421585])
ITU_HP_B = np.array([1.0, -2.0, 1.0])
ITU_HP_A = np.array([1.0, -1.99004745483398, 0.99007225036621])


def _response_db(stages, freqs, rate):
    import scipy.signal as ss
    h = np.ones(len(freqs), dtype=complex)
    for b, a in stages:
        h = h * ss.freqz(b, a, worN=freqs, fs=rate)[1]
    return 20 * np.log10(np.abs(h))


def test_k_weighting_vs_itu_table():
    meter = pyloudnorm_leaf.Meter(48000)
    leaf = [(np.asarray(f.b, dtype=np.float64), np.asarray(f.a, dtype=np.float64)) for f in meter._filters.values()]
    sos = np.asarray(tables.weighting_sos(48000)[0], dtype=np.float64)
    ours = [(sos[i, :3] / sos[i, 3], sos[i, 3:] / sos[i, 3]) for i in range(sos.shape[0])]
    for stages in (leaf, ours):
        assert len(stages) == 2
        (sb, sa), (hb, ha) = stages
        # the shelf reproduces the standard's table; the high-pass its poles (pyloudnorm normalises the
        # high-pass numerator by the passband gain instead of printing 1, -2, 1: a 0.04 dB level offset)
        assert np.abs(sb - ITU_SHELF_B).max() < 2e-4 and np.abs(sa - ITU_SHELF_A).max() < 1e-4
        assert np.abs(ha - ITU_HP_A).max() < 1e-4
        assert np.abs(hb / hb[0] - ITU_HP_B).max() < 1e-6
        f = np.geomspace(20.0, 20000.0, 300)
        d = _response_db(stages, f, 48000) - _response_db([(ITU_SHELF_B, ITU_SHELF_A), (ITU_HP_B, ITU_HP_A)], f, 48000)
        assert np.abs(d).max() < 0.05          # well inside the +-0.1 LU budget of the path
    # product design == oracle leaf (two write-ups of one formula) -- now both pinned to the table
    for (b1, a1), (b2, a2) in zip(leaf, ours):
        assert np.allclose(b1, b2, rtol=0, atol=1e-12) and np.allclose(a1, a2, rtol=0, atol=1e-12)


def test_hann_window_closed_form():
    for n in (512, 2048, 400):
        w = tables.window_np("hann", n)
        k = np.arange(n)
        assert np.abs(w - (0.5 - 0.5 * np.cos(2 * np.pi * k / n))).max() < 1e-7          # periodic (fftbins=True)
        assert np.abs(tables.window_np("sqrt_hann", n) - np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * k / n))).max() < 1e-6


def test_resample_keeps_a_bandlimited_sine():
    """44.1 k -> 16 k -> 44.1 k of a 1 kHz + 3 kHz mixture: amplitude and phase survive both ways
    (away from the edges), which pins the kernel's gain, centring and time alignment."""
    sr, new = 44100, 16000
    T = sr * 2
    t = torch.arange(T, dtype=torch.float64) / sr
    x = (0.5 * torch.sin(2 * math.pi * 1000 * t) + 0.25 * torch.sin(2 * math.pi * 3000 * t + 0.7)).float()[None, None]
    y = restate.resample(x, sr, new)
    tn = torch.arange(y.shape[-1], dtype=torch.float64) / new
    want = (0.5 * torch.sin(2 * math.pi * 1000 * tn) + 0.25 * torch.sin(2 * math.pi * 3000 * tn + 0.7)).float()
    mid = slice(2000, y.shape[-1] - 2000)
    assert float((y[0, 0, mid] - want[mid]).abs().max()) < 2e-3
    back = restate.resample(y, new, sr)
    n = min(back.shape[-1], T)
    mid = slice(6000, n - 6000)
    assert float((back[0, 0, mid] - x[0, 0, mid]).abs().max()) < 4e-3


@pytest.mark.parametrize("old,new", [(2, 1), (3, 1), (1, 2)])
def test_resample_integer_ratio_vs_scipy_polyphase(old, new):
    """The polyphase APPLICATION (tap alignment, phase order, edge replication) against
    scipy.signal.upfirdn-style filtering with the very same kernel bank."""
    import scipy.signal as ss
    torch.manual_seed(3)
    T = 4000
    x = torch.randn(1, 1, T)
    y = restate.resample(x, old, new)[0, 0].numpy().astype(np.float64)
    bank, o, n, width = tables.resample_bank(old, new)
    assert (o, n) == (old, new)
    bank = np.asarray(bank, dtype=np.float64)                        # (new phases, 2 width + old taps), julius layout
    xp = np.pad(x[0, 0].numpy().astype(np.float64), (width, width + old), mode="edge")
    out = np.zeros((new, (len(xp) - bank.shape[1]) // old + 1))
    for p in range(new):
        full = ss.correlate(xp, bank[p], mode="valid")
        out[p] = full[::old][: out.shape[1]]
    ref = out.T.reshape(-1)[: len(y)]
    assert np.abs(ref - y).max() < 1e-5 * max(1.0, np.abs(y).max())


@pytest.mark.parametrize("cutoff", [0.25, 0.1, 4000 / 44100, 8000 / 48000, 0.45, 0.02])
def test_lowpass_taps_vs_scipy_firwin(cutoff):
    """julius.lowpass.LowPassFilters (the leaf behind low_pass / high_pass / equalizer, dsp.py:153-215):
    a symmetric-Hann-windowed sinc of half length int(zeros / cutoff / 2) with unit DC gain -- the
    definition scipy.signal.firwin(window="hann") implements independently."""
    import scipy.signal as ss
    for zeros in (8, 51):            # 8: the band-split bank of equalizer(); 51: low_pass / high_pass (dsp.py:153)
        h = tables.lowpass_half_size(cutoff, zeros)
        ours = np.asarray(tables.lowpass_taps(torch.tensor(cutoff), float(zeros), h), dtype=np.float64).reshape(-1)
        want = ss.firwin(2 * h + 1, cutoff, window="hann", fs=1.0)
        assert ours.shape == want.shape and np.abs(ours - want).max() < 2e-7
    x = torch.randn(1, 1, 6000, generator=torch.Generator().manual_seed(1))
    y = restate.low_pass(x, torch.tensor([cutoff * 16000.0]), 16000)[0, 0].numpy().astype(np.float64)
    xp = np.pad(x[0, 0].numpy().astype(np.float64), (h, h), mode="edge")          # julius pads by replication
    ref = ss.correlate(xp, want, mode="valid")
    assert ref.shape == y.shape and np.abs(ref - y).max() < 2e-5


def test_htk_band_edges_vs_transformers():
    au = _hf_audio_utils()
    for sr, n_bands in [(44100, 6), (48000, 6), (16000, 4), (44100, 12)]:
        mels = np.linspace(au.hertz_to_mel(0.0, "htk"), au.hertz_to_mel(sr / 2, "htk"), n_bands + 1)
        want = au.mel_to_hertz(mels, "htk")[1:-1]
        assert np.abs(tables.htk_band_edges(sr, n_bands) - want).max() < 1e-6 * sr


def _tone(db, seconds, sr=48000, f=1000.0):
    t = torch.arange(int(seconds * sr), dtype=torch.float64) / sr
    return (10 ** (db / 20) * torch.sin(2 * math.pi * f * t)).float()


def test_loudness_ebu_tech_3341_cases():
    """EBU Tech 3341 minimum-requirement signals (published expected values, tolerance +-0.1 LU):
    case 1/2 -- stereo 1 kHz sine at -23 / -33 dBFS for 20 s reads -23.0 / -33.0 LUFS;
    case 3 -- -36 dBFS 10 s, -23 dBFS 60 s, -36 dBFS 10 s reads -23.0 LUFS (relative gate at work);
    case 5 -- -26 dBFS 20 s, -20 dBFS 20.1 s, -26 dBFS 20 s reads -23.0 LUFS.
    Run on the oracle (the reference's CPU / IIR branch) and on the product's CPU path."""
    import audiotools_amd as A
    sr = 48000
    cases = [
        (torch.cat([_tone(-23.0, 20)]), -23.0),
        (torch.cat([_tone(-33.0, 20)]), -33.0),
        (torch.cat([_tone(-36.0, 10), _tone(-23.0, 60), _tone(-36.0, 10)]), -23.0),
        (torch.cat([_tone(-26.0, 20), _tone(-20.0, 20.1), _tone(-26.0, 20)]), -23.0),
    ]
    for mono, want in cases:
        x = mono[None, None].repeat(1, 2, 1)                 # stereo, in phase
        got = float(restate.loudness(x, sr)[0])
        assert abs(got - want) < 0.1, (got, want)
        got2 = float(A.AudioSignal(x.clone(), sr).loudness()[0])
        assert abs(got2 - want) < 0.1, (got2, want)
