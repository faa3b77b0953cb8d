"""time_stretch / pitch_shift (reference effects.py:247-309 = CPU libsox).
The reference's output cannot be pinned (no sox here, no fixtures; its own tests compare batched
with single only), so parity is by PROPERTIES -- output length, pitch ratio, tempo, batched ==
single -- plus the phase-vocoder arithmetic against the oracle's float64 restatement."""
import numpy as np
import pytest
import torch

import audiotools_amd as A
from audiotools_amd import fx
from oracle import restate
from tests import synth

SR = 44100


def _tone(freq, dur=1.5, B=1, C=1):
    t = torch.arange(int(SR * dur)) / SR
    return (0.5 * torch.sin(2 * np.pi * freq * t))[None, None].repeat(B, C, 1)


def dominant_hz(sig, item=0, ch=0):
    x = sig.audio_data[item, ch].detach().cpu().double()
    X = torch.fft.rfft(x * torch.hann_window(x.numel(), dtype=torch.float64))
    return float(X.abs().argmax()) * sig.sample_rate / x.numel()


@pytest.mark.parametrize("p,q", [(4, 5), (5, 4), (303, 227), (1, 2), (3, 1), (1, 1)])
def test_phase_vocoder_torch_vs_oracle(p, q):
    x = synth.audio_batch(2, 2, 30000, seed=p + q, gaps=False)
    X = A.AudioSignal(x, SR).stft()
    Y = fx.phase_vocoder_torch(X, p, q, 512)
    ref = restate.phase_vocoder_f64(X.numpy(), p, q, 512)
    assert Y.shape == ref.shape and Y.shape[-1] == -(-X.shape[-1] * q // p)
    assert np.abs(Y.numpy() - ref).max() / np.abs(ref).max() < 1e-4
    if p == q:   # rate 1: magnitudes unchanged, phases are the input's up to rounding
        assert np.abs(Y.numpy() - X.numpy()).max() / np.abs(ref).max() < 1e-4


@pytest.mark.parametrize("factor", [0.8, 1.25, 2.0])
def test_time_stretch_properties(factor):
    s = A.AudioSignal(_tone(440.0), SR)
    T = s.signal_length
    y = s.clone().time_stretch(factor)
    assert y.signal_length == int(round(T / factor)) and y.sample_rate == SR       # tempo changes the length ...
    assert abs(dominant_hz(y) - 440.0) < 2.0                                        # ... not the pitch
    assert float(y.audio_data.abs().max()) == pytest.approx(0.5, rel=0.3)
    assert y.stft_data is None


@pytest.mark.parametrize("n", [5, -5, 12, -12, 1])
def test_pitch_shift_properties(n):
    s = A.AudioSignal(_tone(440.0), SR)
    y = s.clone().pitch_shift(n)
    assert y.signal_length == s.signal_length and y.sample_rate == SR               # duration and rate kept
    want = 440.0 * 2 ** (n / 12)
    assert abs(dominant_hz(y) / want - 1) < 0.003                                   # within 5 cents
    assert s.clone().pitch_shift(0).audio_data.equal(s.audio_data)


def test_batched_equals_single():
    """The reference's own test (tests/core/test_effects.py:156-181): item 0 of a batch == the item alone."""
    x = synth.audio_batch(3, 2, 30000, seed=2, gaps=False)
    s = A.AudioSignal(x, SR)
    for fn in (lambda a: a.pitch_shift(5), lambda a: a.time_stretch(0.8)):
        single = fn(A.AudioSignal(x[:1].clone(), SR))
        batched = fn(s.clone())
        assert torch.allclose(batched.audio_data[:1], single.audio_data, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("p,q", [(4, 5), (5, 4), (303, 227), (1, 3), (3, 1)])
def test_phase_vocoder_kernel_vs_oracle(p, q):
    from audiotools_amd import kernels
    x = synth.audio_batch(3, 2, 60000, seed=p * q, gaps=False)
    s = A.AudioSignal(x, SR).to("cuda")
    X = s.stft()
    Y = kernels.phase_vocoder(X, p, q, 512)
    ref = restate.phase_vocoder_f64(X.cpu().numpy(), p, q, 512)
    assert tuple(Y.shape) == ref.shape and Y.stride()[-2] == 1
    assert np.abs(Y.cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4


@pytest.mark.gpu
def test_stretch_and_pitch_gpu_vs_cpu_and_properties():
    x = synth.audio_batch(4, 2, 2 * SR, seed=9, gaps=False)
    for fn in (lambda a: a.time_stretch(0.8), lambda a: a.time_stretch(1.3), lambda a: a.pitch_shift(4), lambda a: a.pitch_shift(-7)):
        ref = fn(A.AudioSignal(x.clone(), SR)).audio_data
        got = fn(A.AudioSignal(x.clone(), SR).to("cuda")).audio_data
        assert got.shape == ref.shape
        assert float((got.cpu() - ref).abs().max() / ref.abs().max()) < 1e-3
    tone = A.AudioSignal(_tone(440.0, B=2, C=2), SR).to("cuda")
    up = tone.clone().pitch_shift(12)
    assert up.signal_length == tone.signal_length and abs(dominant_hz(up) / 880.0 - 1) < 0.003
    slow = tone.clone().time_stretch(0.5)
    assert slow.signal_length == 2 * tone.signal_length and abs(dominant_hz(slow) - 440.0) < 2.0
