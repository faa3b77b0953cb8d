"""Spectral training losses (reference ``audiotools/metrics/spectral.py``).

Each scale calls ``AudioSignal.stft`` / ``mel_spectrogram`` on both signals; for HIP tensors that
require gradients those run the fused forward kernels and their adjoints
(``spectral._NativeStft`` / ``_NativeStftMel``), so a multi-scale mel loss never materialises a
spectrum gradient in HBM."""
import typing

import numpy as np
from torch import nn

from ..spectral import STFTParams


def _scales(window_lengths, match_stride, window_type):
    return [STFTParams(window_length=w, hop_length=w // 4, match_stride=match_stride, window_type=window_type)
            for w in window_lengths]


def _log_and_linear(loss_fn, a, b, clamp_eps, power, log_weight, mag_weight):
    """log_weight * d(log10 clamp(a)^p, log10 clamp(b)^p) + mag_weight * d(a, b)
    (spectral.py:88-93, 188-193)."""
    la = a.clamp(clamp_eps).pow(power).log10()
    lb = b.clamp(clamp_eps).pow(power).log10()
    return log_weight * loss_fn(la, lb) + mag_weight * loss_fn(a, b)


class MultiScaleSTFTLoss(nn.Module):
    """Multi-resolution STFT magnitude distance (DDSP, Engel et al. 2019; spectral.py:11-94):
    for every window length w (hop w/4) the log-power and the linear magnitudes are compared with
    ``loss_fn``.  ``weight`` is carried for the trainer, not applied here."""

    def __init__(self, window_lengths: typing.List[int] = [2048, 512], loss_fn: typing.Callable = nn.L1Loss(),
                 clamp_eps: float = 1e-5, mag_weight: float = 1.0, log_weight: float = 1.0, pow: float = 2.0,
                 weight: float = 1.0, match_stride: bool = False, window_type: str = None):
        super().__init__()
        self.stft_params = _scales(window_lengths, match_stride, window_type)
        self.loss_fn, self.clamp_eps, self.pow = loss_fn, clamp_eps, pow
        self.log_weight, self.mag_weight, self.weight = log_weight, mag_weight, weight

    def forward(self, x, y):
        total = 0.0
        for s in self.stft_params:
            # as in the reference, match_stride of the scale is NOT forwarded to stft()
            x.stft(s.window_length, s.hop_length, s.window_type)
            y.stft(s.window_length, s.hop_length, s.window_type)
            total = total + _log_and_linear(self.loss_fn, x.magnitude, y.magnitude, self.clamp_eps, self.pow,
                                            self.log_weight, self.mag_weight)
        return total


class MelSpectrogramLoss(nn.Module):
    """Multi-resolution mel-spectrogram distance (spectral.py:97-194): scale i uses
    ``n_mels[i]`` bands between ``mel_fmin[i]`` and ``mel_fmax[i]`` on a window of
    ``window_lengths[i]`` (hop = window / 4)."""

    def __init__(self, n_mels: typing.List[int] = [150, 80], window_lengths: typing.List[int] = [2048, 512],
                 loss_fn: typing.Callable = nn.L1Loss(), clamp_eps: float = 1e-5, mag_weight: float = 1.0,
                 log_weight: float = 1.0, pow: float = 2.0, weight: float = 1.0, match_stride: bool = False,
                 mel_fmin: typing.List[float] = [0.0, 0.0], mel_fmax: typing.List[float] = [None, None],
                 window_type: str = None):
        super().__init__()
        self.stft_params = _scales(window_lengths, match_stride, window_type)
        self.n_mels, self.mel_fmin, self.mel_fmax = n_mels, mel_fmin, mel_fmax
        self.loss_fn, self.clamp_eps, self.pow = loss_fn, clamp_eps, pow
        self.log_weight, self.mag_weight, self.weight = log_weight, mag_weight, weight

    def forward(self, x, y):
        total = 0.0
        for n_mels, fmin, fmax, s in zip(self.n_mels, self.mel_fmin, self.mel_fmax, self.stft_params):
            kw = dict(window_length=s.window_length, hop_length=s.hop_length, window_type=s.window_type)
            mx = x.mel_spectrogram(n_mels, mel_fmin=fmin, mel_fmax=fmax, **kw)
            my = y.mel_spectrogram(n_mels, mel_fmin=fmin, mel_fmax=fmax, **kw)
            total = total + _log_and_linear(self.loss_fn, mx, my, self.clamp_eps, self.pow, self.log_weight,
                                            self.mag_weight)
        return total


class PhaseLoss(nn.Module):
    """Magnitude-weighted squared phase difference (spectral.py:197-247).

    Bug-compatible with the reference's wrap-around: differences below -pi get +2 pi, differences
    above +pi ALSO get +2 pi (the reference subtracts a negative), so only one side is wrapped."""

    def __init__(self, window_length: int = 2048, hop_length: int = 512, weight: float = 1.0):
        super().__init__()
        self.weight = weight
        self.stft_params = STFTParams(window_length, hop_length)

    def forward(self, x, y):
        s = self.stft_params
        x.stft(s.window_length, s.hop_length, s.window_type)
        y.stft(s.window_length, s.hop_length, s.window_type)
        diff = x.phase - y.phase
        diff[diff < -np.pi] += 2 * np.pi
        diff[diff > np.pi] += 2 * np.pi
        mag = x.magnitude
        lo, hi = mag.min(), mag.max()
        weights = (mag - lo) / (hi - lo)
        return ((weights * diff) ** 2).mean()
