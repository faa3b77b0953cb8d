"""Training losses (audiotools/metrics/{spectral,distance}.py) on CPU tensors: value and gradient
equality with the UNMODIFIED reference, plus the reference's own property checks
(tests/metrics/test_spectral.py, test_distance.py: identical signals give 0 / -inf)."""
import numpy as np
import pytest
import torch

import audiotools_amd as A
from audiotools_amd import metrics
from tests import synth


def _pair(mod, seed=3, grad=False):
    x = synth.audio_batch(2, 1, 16000, seed=seed, gaps=False)
    y = synth.audio_batch(2, 1, 16000, seed=seed + 1, gaps=False)
    xs = mod.AudioSignal(x.clone().requires_grad_(grad), 44100)
    return xs, mod.AudioSignal(y.clone(), 44100)


CASES = [
    ("spectral", "MultiScaleSTFTLoss", {}),
    ("spectral", "MultiScaleSTFTLoss", {"window_lengths": [1024, 256], "pow": 1.0, "log_weight": 0.5}),
    ("spectral", "MelSpectrogramLoss", {}),
    ("spectral", "MelSpectrogramLoss", {"n_mels": [5, 10, 20, 40, 80, 160, 320], "window_lengths": [32, 64, 128, 256, 512, 1024, 2048],
                                        "mel_fmin": [0] * 7, "mel_fmax": [None] * 7, "pow": 1.0, "mag_weight": 0.0}),
    ("spectral", "PhaseLoss", {}),
    ("distance", "L1Loss", {}),
    ("distance", "SISDRLoss", {}),
    ("distance", "SISDRLoss", {"scaling": False, "zero_mean": False, "reduction": "sum", "clip_min": -30}),
]


@pytest.mark.parametrize("module,name,kw", CASES)
def test_losses_match_reference(reference, module, name, kw):
    ours = getattr(getattr(metrics, module), name)(**kw)
    theirs = getattr(getattr(reference.metrics, module), name)(**kw)
    xa, ya = _pair(A, grad=True)
    xr, yr = _pair(reference, grad=True)
    la, lr = ours(xa, ya), theirs(xr, yr)
    assert torch.allclose(la, lr, rtol=1e-5, atol=1e-6), (float(la), float(lr))
    (ga,) = torch.autograd.grad(la, xa.audio_data)
    (gr,) = torch.autograd.grad(lr, xr.audio_data)
    assert torch.allclose(ga, gr, rtol=1e-4, atol=1e-7 * float(gr.abs().max()) + 1e-12)


def test_loss_properties():
    """tests/metrics/test_spectral.py:7-83, test_distance.py: identity gives 0 (SI-SDR: -inf-ish),
    a different signal gives more."""
    x, y = _pair(A)
    for loss in (metrics.spectral.MultiScaleSTFTLoss(), metrics.spectral.MelSpectrogramLoss(), metrics.spectral.PhaseLoss(),
                 metrics.distance.L1Loss()):
        same = loss(x, x.deepcopy())
        assert np.allclose(float(same), 0, atol=1e-6)
        assert float(loss(x, y)) > float(same)
    sisdr = metrics.distance.SISDRLoss()
    assert float(sisdr(x, x.deepcopy())) < -70          # -10 log10(signal / ~0 + 1e-8)
    assert float(sisdr(x, y)) > float(sisdr(x, x.deepcopy()))
    assert float(metrics.spectral.MultiScaleSTFTLoss(loss_fn=metrics.distance.SISDRLoss())(x, y)) > -200
