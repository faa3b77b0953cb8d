"""The travelling oracle (oracle/restate.py) against the committed fixtures that
oracle/make_golden.py produced from the unmodified reference.  Runs anywhere (no GPU,
no /root/reference)."""
import os

import numpy as np
import torch

from oracle import restate

G = os.path.join(os.path.dirname(__file__), "golden")


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_stft_golden():
    d = np.load(os.path.join(G, "stft_cfg1.npz"))
    x = _t(d["x"])
    assert torch.equal(restate.stft(x, 512, 128, "hann"), _t(d["stft"]))
    assert torch.equal(restate.stft(x, 512, 128, "sqrt_hann", match_stride=True), _t(d["stft_match_stride_sqrt_hann"]))
    assert torch.allclose(restate.istft(_t(d["stft"]), 512, 128, "hann", False, 16000), _t(d["istft"]), atol=1e-7)


def test_mel_golden():
    d = np.load(os.path.join(G, "mel_cfg2.npz"))
    x = _t(d["x"])
    X = restate.stft(x, 2048, 512, "hann")
    assert torch.equal(X, _t(d["stft"]))
    assert np.array_equal(restate.mel_basis(44100, 2048, 80), d["mel_basis"])
    mel = restate.mel_spectrogram(X, 44100, 80)
    assert torch.equal(mel, _t(d["mel"]))
    assert torch.allclose(restate.mfcc(mel), _t(d["mfcc"]), atol=1e-6)


def test_loudness_golden():
    d = np.load(os.path.join(G, "loudness.npz"))
    np.random.seed(0)
    arr = np.random.randn(16, 2, 16000).astype(np.float32)
    assert np.allclose(restate.loudness(_t(arr), 16000).numpy(), d["seeded_randn_16k"], atol=1e-4)
    xg = _t(d["gaps_x"].astype(np.float32))
    assert np.allclose(restate.loudness(xg, 16000).numpy(), d["gaps_lufs"], atol=1e-4)
    assert np.allclose(restate.loudness(xg[:2, :, :32000], 16000, "Fenton/Lee 1").numpy(), d["fenton_lee_1"], atol=1e-4)
    assert np.allclose(restate.loudness(xg[:2, :, :32000], 16000, "Dash et al.").numpy(), d["dash"], atol=1e-4)
    # the reference's literal golden for sine_1000.wav (tests/core/test_loudness.py:61)
    assert abs(float(d["sine_1000_lufs"][0]) - (-3.0523438444331137)) < 1e-3


def test_effects_golden():
    d = np.load(os.path.join(G, "effects_cfg4.npz"))
    x = _t(d["x"])
    assert torch.equal(restate.low_pass(x, _t(d["lp_cut"]), 48000), _t(d["low_pass"]))
    assert torch.equal(restate.high_pass(x, torch.tensor([500.0, 1000.0, 2000.0]), 48000), _t(d["high_pass"]))
    assert torch.equal(restate.equalizer(x, 48000, _t(d["eq_db"])), _t(d["equalizer"]))
    assert torch.equal(restate.convolve(x, _t(d["ir"])), _t(d["convolve"]))
    assert torch.equal(restate.resample(x, 48000, 16000), _t(d["resample_48k_16k"]))
    assert torch.equal(restate.resample(_t(d["xr"]), 44100, 16000), _t(d["resample_441_16k"]))
