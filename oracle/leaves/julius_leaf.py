"""Restatement of the parts of ``julius`` (upstream 0.2.7, unpinned in the
reference's ``setup.py:44``) that the audiotools hot path calls.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The julius source is NOT present in this image or under ``/root/reference``;
what follows restates its published algorithm.
Reference call sites that consume these functions:

* ``julius.resample_frac``      ``audiotools/core/audio_signal.py:732``
* ``julius.LowPassFilter``      ``audiotools/core/dsp.py:178``
* ``julius.HighPassFilter``     ``audiotools/core/dsp.py:210``
* ``julius.SplitBands``         ``audiotools/core/effects.py:400``
* ``julius.fftconv.fft_conv1d`` ``audiotools/core/loudness.py:94``
* ``julius.core.unfold``        ``audiotools/core/loudness.py:171``

All arithmetic is float32 torch, as upstream, so that filter lengths that sit
on an integer boundary (e.g. ``int(51 / f32(4000/48000) / 2)``) resolve the
same way.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------- core
def sinc(x: torch.Tensor):
    """julius.core.sinc: sin(x)/x with sinc(0) = 1 (NOT the normalised sinc)."""
    return torch.where(
        x == 0, torch.tensor(1.0, device=x.device, dtype=x.dtype), torch.sin(x) / x
    )


def unfold(input: torch.Tensor, kernel_size: int, stride: int):
    """julius.core.unfold (Appendix A.1): frames along the last axis, the
    tail is right-zero-padded so the last frame is complete."""
    shape = list(input.shape)
    length = shape.pop(-1)
    n_frames = math.ceil((max(length, kernel_size) - kernel_size) / stride) + 1
    tgt_length = (n_frames - 1) * stride + kernel_size
    padded = F.pad(input, (0, tgt_length - length)).contiguous()
    strides = [padded.stride(d) for d in range(padded.dim())]
    assert strides.pop(-1) == 1
    strides = strides + [stride, 1]
    return padded.as_strided(shape + [n_frames, kernel_size], strides)


def hz_to_mel(f):
    return 2595 * np.log10(1 + f / 700)


def mel_to_hz(m):
    return 700 * (10 ** (m / 2595) - 1)


def mel_frequencies(n_mels: int, fmin: float, fmax: float):
    """julius.core.mel_frequencies -- HTK mel scale, linspace in mel."""
    mels = np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels)
    return mel_to_hz(mels)


# ------------------------------------------------------------------ fftconv
def fft_conv1d(input, weight, bias=None, stride: int = 1, padding: int = 0,
               block_ratio: float = 5):
    """julius.fftconv.fft_conv1d (Appendix A.2).

    Upstream evaluates the cross-correlation block-wise with FFTs; any exact
    conv1d is an admissible restatement (results agree to f32 round-off), so
    the direct form is used."""
    return F.conv1d(input, weight, bias, stride=stride, padding=padding)


# ------------------------------------------------------------------ lowpass
class LowPassFilters(torch.nn.Module):
    """julius.lowpass.LowPassFilters (Appendix A.3): bank of Hann-windowed
    sinc low-pass FIRs sharing one length, unit DC gain, replicate padding."""

    def __init__(self, cutoffs, stride: int = 1, pad: bool = True,
                 zeros: float = 8, fft=None):
        super().__init__()
        self.cutoffs = list(cutoffs)
        if min(self.cutoffs) < 0:
            raise ValueError("Minimum cutoff must be larger than zero.")
        if max(self.cutoffs) > 0.5:
            raise ValueError("A cutoff above 0.5 does not make sense.")
        self.stride = stride
        self.pad = pad
        self.zeros = zeros
        self.half_size = int(zeros / min([c for c in self.cutoffs if c > 0]) / 2)
        if fft is None:
            fft = self.half_size > 32
        self.fft = fft
        window = torch.hann_window(2 * self.half_size + 1, periodic=False)
        time = torch.arange(-self.half_size, self.half_size + 1)
        filters = []
        for cutoff in cutoffs:
            if cutoff == 0:
                filter_ = torch.zeros_like(time)
            else:
                filter_ = 2 * cutoff * window * sinc(2 * cutoff * math.pi * time)
                # unit DC gain
                filter_ /= filter_.sum()
            filters.append(filter_)
        self.register_buffer("filters", torch.stack(filters)[:, None])

    def forward(self, input):
        shape = list(input.shape)
        input = input.reshape(-1, 1, shape[-1])
        if self.pad:
            input = F.pad(input, (self.half_size, self.half_size), mode="replicate")
        if self.fft:
            out = fft_conv1d(input, self.filters, stride=self.stride)
        else:
            out = F.conv1d(input, self.filters, stride=self.stride)
        shape.insert(0, len(self.cutoffs))
        shape[-1] = out.shape[-1]
        return out.permute(1, 0, 2).reshape(shape)


class LowPassFilter(torch.nn.Module):
    def __init__(self, cutoff, stride: int = 1, pad: bool = True,
                 zeros: float = 8, fft=None):
        super().__init__()
        self._lowpasses = LowPassFilters([cutoff], stride, pad, zeros, fft)

    def forward(self, input):
        return self._lowpasses(input)[0]


def lowpass_filters(input, cutoffs, stride=1, pad=True, zeros=8, fft=None):
    return LowPassFilters(cutoffs, stride, pad, zeros, fft).to(input)(input)


def lowpass_filter(input, cutoff, stride=1, pad=True, zeros=8, fft=None):
    return lowpass_filters(input, [cutoff], stride, pad, zeros, fft)[0]


# ----------------------------------------------------------------- highpass
class HighPassFilters(torch.nn.Module):
    """julius.highpass.HighPassFilters: x - lowpass(x)."""

    def __init__(self, cutoffs, stride: int = 1, pad: bool = True,
                 zeros: float = 8, fft=None):
        super().__init__()
        self._lowpasses = LowPassFilters(cutoffs, stride, pad, zeros, fft)

    @property
    def cutoffs(self):
        return self._lowpasses.cutoffs

    def forward(self, input):
        lows = self._lowpasses(input)
        if self._lowpasses.pad:
            start, end = 0, input.shape[-1]
        else:
            start = self._lowpasses.half_size
            end = -start
        input = input[..., start:end:self._lowpasses.stride]
        highs = input - lows
        return highs


class HighPassFilter(torch.nn.Module):
    def __init__(self, cutoff, stride: int = 1, pad: bool = True,
                 zeros: float = 8, fft=None):
        super().__init__()
        self._highpasses = HighPassFilters([cutoff], stride, pad, zeros, fft)

    def forward(self, input):
        return self._highpasses(input)[0]


# -------------------------------------------------------------------- bands
class SplitBands(torch.nn.Module):
    """julius.bands.SplitBands (Appendix A.4): successive differences of a
    LowPassFilters bank with HTK-mel-spaced cutoffs; bands sum to the input."""

    def __init__(self, sample_rate: float, n_bands=None, cutoffs=None,
                 pad: bool = True, zeros: float = 8, fft=None):
        super().__init__()
        if (cutoffs is None) + (n_bands is None) != 1:
            raise ValueError("You must provide either n_bands, or cutoffs, but not boths.")
        self.sample_rate = sample_rate
        self.n_bands = n_bands
        self._cutoffs = list(cutoffs) if cutoffs is not None else None
        self.pad = pad
        self.zeros = zeros
        self.fft = fft
        if cutoffs is None:
            if n_bands is None:
                raise ValueError("You must provide one of n_bands or cutoffs.")
            if not n_bands >= 1:
                raise ValueError(f"n_bands must be greater than one (got {n_bands})")
            cutoffs = mel_frequencies(n_bands + 1, 0, sample_rate / 2)[1:-1]
        else:
            if max(cutoffs) > 0.5 * sample_rate:
                raise ValueError("A cutoff above sample_rate/2 does not make sense.")
        if len(cutoffs) > 0:
            self.lowpass = LowPassFilters(
                [c / sample_rate for c in cutoffs], pad=pad, zeros=zeros, fft=fft)
        else:
            self.lowpass = None

    def forward(self, input):
        if self.lowpass is None:
            return input[None]
        lows = self.lowpass(input)
        low = lows[0]
        bands = [low]
        for low_and_band in lows[1:]:
            band = low_and_band - low
            bands.append(band)
            low = low_and_band
        bands.append(input - low)
        return torch.stack(bands)

    @property
    def cutoffs(self):
        if self._cutoffs is not None:
            return self._cutoffs
        elif self.lowpass is not None:
            return [c * self.sample_rate for c in self.lowpass.cutoffs]
        else:
            return []


def split_bands(signal, sample_rate, n_bands=None, cutoffs=None, pad=True,
                zeros=8, fft=None):
    return SplitBands(sample_rate, n_bands, cutoffs, pad, zeros, fft).to(signal)(signal)


# ----------------------------------------------------------------- resample
class ResampleFrac(torch.nn.Module):
    """julius.resample.ResampleFrac (Appendix A.5): windowed-sinc polyphase
    resampling by the reduced ratio new_sr/old_sr."""

    def __init__(self, old_sr: int, new_sr: int, zeros: int = 24,
                 rolloff: float = 0.945):
        super().__init__()
        if not isinstance(old_sr, int) or not isinstance(new_sr, int):
            raise ValueError("old_sr and new_sr should be integers")
        gcd = math.gcd(old_sr, new_sr)
        self.old_sr = old_sr // gcd
        self.new_sr = new_sr // gcd
        self.zeros = zeros
        self.rolloff = rolloff
        self._init_kernels()

    def _init_kernels(self):
        if self.old_sr == self.new_sr:
            return
        kernels = []
        sr = min(self.new_sr, self.old_sr)
        sr *= self.rolloff
        self._width = math.ceil(self.zeros * self.old_sr / sr)
        idx = torch.arange(-self._width, self._width + self.old_sr).float()
        for i in range(self.new_sr):
            t = (-i / self.new_sr + idx / self.old_sr) * sr
            t = t.clamp_(-self.zeros, self.zeros)
            t *= math.pi
            window = torch.cos(t / self.zeros / 2) ** 2
            kernel = sinc(t) * window
            kernel.div_(kernel.sum())
            kernels.append(kernel)
        self.register_buffer(
            "kernel", torch.stack(kernels).view(self.new_sr, 1, -1))

    def forward(self, x, output_length=None, full: bool = False):
        if self.old_sr == self.new_sr:
            return x
        shape = x.shape
        length = x.shape[-1]
        x = x.reshape(-1, length)
        x = F.pad(x[:, None], (self._width, self._width + self.old_sr),
                  mode="replicate")
        ys = F.conv1d(x, self.kernel, stride=self.old_sr)
        y = ys.transpose(1, 2).reshape(list(shape[:-1]) + [-1])
        float_output_length = torch.as_tensor(self.new_sr * length / self.old_sr)
        max_output_length = torch.ceil(float_output_length).long()
        default_output_length = torch.floor(float_output_length).long()
        if output_length is None:
            applied_output_length = max_output_length if full else default_output_length
        elif output_length < 0 or output_length > max_output_length:
            raise ValueError(f"output_length must be between 0 and {max_output_length.item()}")
        else:
            applied_output_length = torch.tensor(output_length)
            if full:
                raise ValueError("You cannot pass both full=True and output_length")
        return y[..., :applied_output_length]


def resample_frac(x, old_sr: int, new_sr: int, zeros: int = 24,
                  rolloff: float = 0.945, output_length=None, full: bool = False):
    return ResampleFrac(old_sr, new_sr, zeros, rolloff).to(x)(x, output_length, full)
