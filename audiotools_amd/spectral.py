"""STFT-domain methods of ``AudioSignal`` (reference
``audiotools/core/audio_signal.py:28-50, 1009-1516``).

HIP tensors (float32, no autograd) run the fused kernel in ``csrc/stft.hip``;
tensors that need autograd, or CPU tensors, run the same maths through torch
ops (the reference's own formulation) so that ``tests/core/test_grad.py``-style
use keeps working.
"""
import collections
import functools
import math

import numpy as np
import torch

from . import _native, kernels, tables

STFTParams = collections.namedtuple(
    "STFTParams", ["window_length", "hop_length", "window_type", "match_stride", "padding_type"])
STFTParams.__new__.__defaults__ = (None, None, None, None, None)
STFTParams.__doc__ = """STFT configuration of an AudioSignal (audio_signal.py:28-50).

window_length : int   default 2 ** ceil(log2(0.032 * sample_rate))
hop_length    : int   default window_length // 4
window_type   : str   scipy window name, "sqrt_hann" or "average"; default "hann"
match_stride  : bool  pad so that frames * hop == samples (conv-stride alignment); default False
padding_type  : str   torch pad mode for the match_stride padding; default "reflect"
"""


class _NativeStft(torch.autograd.Function):
    """stft() with the HIP kernels in both directions: forward = fused STFT kernel, backward = its
    adjoint (the fused inverse kernel in adjoint mode).  This is what lets the reference's training
    losses (metrics/spectral.py: MultiScaleSTFTLoss, MelSpectrogramLoss, PhaseLoss) stay on the
    native path; everything after the STFT (abs, log, mel matmul) is ordinary differentiable torch."""

    @staticmethod
    def forward(ctx, audio, window, n_fft, hop):
        ctx.save_for_backward(window)
        ctx.cfg = (n_fft, hop, audio.shape[-1])
        X, _ = kernels.stft_mel(audio.detach(), window, n_fft, hop)
        return X

    @staticmethod
    def backward(ctx, grad):
        (window,) = ctx.saved_tensors
        n_fft, hop, T = ctx.cfg
        return kernels.stft_adjoint(grad, window, n_fft, hop, T), None, None, None


class _NativeStftGeneral(torch.autograd.Function):
    """stft() under autograd for every other native transform: the run-time sizes (n_fft 4096 / 8192, 400 / 1200 /
    1920 ...: the window lengths metrics/spectral.py:9-247 accepts) and ``match_stride`` (outer padding of any
    ``padding_type``, two frames dropped per side).  Forward = the kernel the no-grad path runs; backward =
    ``kernels.stft_adjoint_general`` (inverse kernels with the envelope division undone, or the dedicated adjoint kernel
    for the fused sizes, plus the folds of the two paddings)."""

    @staticmethod
    def forward(ctx, audio, window, n_fft, hop, pad, right_pad, padding_type, match_stride):
        ctx.save_for_backward(window)
        ctx.cfg = (n_fft, hop, audio.shape[-1], pad, right_pad, padding_type, match_stride)
        X, _ = kernels.stft_mel(audio.detach(), window, n_fft, hop, pad=pad, right_pad=right_pad,
                                padding_type=padding_type, match_stride=match_stride)
        return X

    @staticmethod
    def backward(ctx, grad):
        (window,) = ctx.saved_tensors
        n_fft, hop, T, pad, right_pad, padding_type, match_stride = ctx.cfg
        g = kernels.stft_adjoint_general(grad, window, n_fft, hop, T, pad, right_pad, padding_type, match_stride)
        return (g,) + (None,) * 7


class _NativeStftMel(torch.autograd.Function):
    """mel_spectrogram() under autograd: ONE fused forward kernel (STFT + |X| + banded mel), and a
    backward that maps dL/dmel to dL/dX with the dense basis (``g_mag = g_mel @ basis``,
    ``g_X += g_mag * X / |X|``, the derivative torch's abs() uses) before the native adjoint."""

    @staticmethod
    def forward(ctx, audio, window, n_fft, hop, units, basis, bin_table):
        info, w, n_mels = units
        X, mel = kernels.stft_mel(audio.detach(), window, n_fft, hop, mel=(info, w, n_mels))
        ctx.save_for_backward(window, X, basis)
        ctx.cfg = (n_fft, hop, audio.shape[-1])
        ctx.bin_table = bin_table
        ctx.set_materialize_grads(False)   # an unused stft_data output must arrive as None, not as 7 GB of zeros
        return X, mel

    @staticmethod
    def backward(ctx, gX, gmel):
        window, X, basis = ctx.saved_tensors
        n_fft, hop, T = ctx.cfg
        if gX is None and gmel is not None and ctx.bin_table is not None:
            # the usual case (a loss on the mel output only): one kernel, dL/dX never materialised
            return (kernels.stft_mel_adjoint(X, gmel, ctx.bin_table, window, n_fft, hop, T),) + (None,) * 6
        g = gX
        if gmel is not None:
            # mel[b,c,m,n] = sum_f basis[m,f] |X[b,c,f,n]|
            g_mag = torch.matmul(gmel.transpose(2, 3), basis).transpose(2, 3)      # (B, C, F, N)
            mag = X.abs()
            unit = torch.where(mag > 0, X / mag.clamp_min(torch.finfo(mag.dtype).tiny), torch.zeros_like(X))
            gm = g_mag * unit
            g = gm if g is None else g + gm
        if g is None:
            return (None,) * 7
        return (kernels.stft_adjoint(g, window, n_fft, hop, T),) + (None,) * 6


class _NativeIstft(torch.autograd.Function):
    """istft() with the HIP kernels in both directions: forward = fused inverse kernel, backward =
    the forward STFT kernel applied to grad / envelope (``kernels.istft_adjoint``)."""

    @staticmethod
    def forward(ctx, X, window, n_fft, hop, length):
        ctx.save_for_backward(window)
        ctx.cfg = (n_fft, hop, X.shape[-1])
        return kernels.istft(X.detach(), window, n_fft, hop, length)

    @staticmethod
    def backward(ctx, grad):
        (window,) = ctx.saved_tensors
        n_fft, hop, n_frames = ctx.cfg
        return kernels.istft_adjoint(grad.contiguous(), window, n_fft, hop, n_frames), None, None, None, None


def _native_autograd_ok(audio: torch.Tensor, n_fft: int, hop: int, match_stride: bool) -> bool:
    """A HIP float32 tensor that needs gradients can use the native forward + adjoint pair."""
    return (audio.is_cuda and audio.dtype == torch.float32 and audio.requires_grad and torch.is_grad_enabled()
            and not match_stride and kernels.stft_fused_supported(n_fft) and kernels.istft_fused_supported(n_fft, hop)
            and audio.shape[-1] > n_fft // 2)


def _native_autograd_general_ok(audio: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, pad: int, right_pad: int,
                                padding_type: str) -> bool:
    """The same for the transforms ``_native_autograd_ok`` leaves out (run-time sizes, match_stride)."""
    T2 = audio.shape[-1] + 2 * pad + right_pad
    return (audio.is_cuda and audio.dtype == torch.float32 and audio.requires_grad and torch.is_grad_enabled()
            and padding_type in kernels.PAD_MODES and T2 > n_fft // 2 + 1
            and (padding_type != "reflect" or pad + right_pad < audio.shape[-1])
            and kernels.stft_adjoint_general_supported(window, n_fft, hop))


class SpectralMixin:
    # ---------------------------------------------------------------- params
    @staticmethod
    @functools.lru_cache(None)
    def get_window(window_type: str, window_length: int, device: str):
        """Analysis window as a float32 tensor on ``device`` (cached)."""
        return tables.window(window_type, int(window_length), torch.device(device))

    @property
    def stft_params(self):
        return self._stft_params

    @stft_params.setter
    def stft_params(self, value: STFTParams):
        win = int(2 ** (np.ceil(np.log2(0.032 * self.sample_rate))))
        defaults = dict(window_length=win, hop_length=win // 4, window_type="hann",
                        match_stride=False, padding_type="reflect")
        given = value._asdict() if value else defaults
        merged = {k: (defaults[k] if given[k] is None else given[k]) for k in defaults}
        self._stft_params = STFTParams(**merged)
        self.stft_data = None

    def compute_stft_padding(self, window_length: int, hop_length: int, match_stride: bool):
        """(right_pad, pad) added around the audio before torch.stft-style
        centring when ``match_stride`` (audio_signal.py:1089-1121)."""
        if not match_stride:
            return 0, 0
        assert hop_length == window_length // 4, "For match_stride, hop must equal n_fft // 4"
        length = self.signal_length
        right_pad = math.ceil(length / hop_length) * hop_length - length
        pad = (window_length - hop_length) // 2
        return right_pad, pad

    def _resolve(self, window_length, hop_length, window_type, match_stride, padding_type=None):
        p = self.stft_params
        return (p.window_length if window_length is None else int(window_length),
                p.hop_length if hop_length is None else int(hop_length),
                p.window_type if window_type is None else window_type,
                p.match_stride if match_stride is None else match_stride,
                p.padding_type if padding_type is None else padding_type)

    # ------------------------------------------------------------------ STFT
    def _torch_stft(self, n_fft, hop, window, match_stride, padding_type, right_pad, pad):
        x = torch.nn.functional.pad(self.audio_data, (pad, pad + right_pad), padding_type)
        X = torch.stft(x.reshape(-1, x.shape[-1]), n_fft=n_fft, hop_length=hop, window=window,
                       return_complex=True, center=True)
        X = X.reshape(self.batch_size, self.num_channels, X.shape[1], X.shape[2])
        return X[..., 2:-2] if match_stride else X

    def _release_stft_data(self):
        """Drop the spectrum a native stft() / mel_spectrogram() is about to replace (and any edit pending on it) and return
        its shape.  The call WILL assign a new one; releasing the old tensor first lets its storage be reused for the new
        result when nobody else holds it -- what the caching allocator does for a temporary, one call earlier."""
        old = self._stft_data
        self._pending_edit = None
        self._stft_data = None
        return None if old is None else tuple(old.shape)

    @staticmethod
    def _warn_if_shape_changed(old_shape, new):
        if old_shape is not None and old_shape != tuple(new.shape):
            import warnings
            warnings.warn("stft_data changed shape")       # (audio_signal.py:942-943; the setter no longer sees the old tensor)

    def stft(self, window_length: int = None, hop_length: int = None, window_type: str = None,
             match_stride: bool = None, padding_type: str = None):
        """Short-time Fourier transform -> complex64 (B, C, F, N); also stored
        in ``stft_data`` (audio_signal.py:1123-1212)."""
        n_fft, hop, wtype, match_stride, padding_type = self._resolve(
            window_length, hop_length, window_type, match_stride, padding_type)
        audio = self.audio_data
        window = self.get_window(wtype, n_fft, str(audio.device))
        right_pad, pad = self.compute_stft_padding(n_fft, hop, match_stride)
        if kernels.is_native(audio) and kernels.stft_native_supported(n_fft):
            # the spectrum this call replaces is released BEFORE the kernel runs: when nothing else references it, the
            # placement-aware pool (kernels._PlacedOutputs) can hand its buffer -- the fastest one it knows -- out again
            old_shape = self._release_stft_data()
            X, _ = kernels.stft_mel(audio, window, n_fft, hop, pad=pad, right_pad=right_pad,
                                    padding_type=padding_type, match_stride=match_stride)
            self._warn_if_shape_changed(old_shape, X)
        elif _native_autograd_ok(audio, n_fft, hop, match_stride):
            X = _NativeStft.apply(audio, window, n_fft, hop)
        elif _native_autograd_general_ok(audio, window, n_fft, hop, pad, right_pad, padding_type):
            X = _NativeStftGeneral.apply(audio, window, n_fft, hop, pad, right_pad, padding_type, bool(match_stride))
        else:
            X = self._torch_stft(n_fft, hop, window, match_stride, padding_type, right_pad, pad)
        self.stft_data = X
        return X

    def istft(self, window_length: int = None, hop_length: int = None, window_type: str = None,
              match_stride: bool = None, length: int = None):
        """Inverse STFT of ``stft_data`` into ``audio_data`` (audio_signal.py:1214-1296)."""
        if self._stft_data is None:
            raise RuntimeError("Cannot do inverse STFT without self.stft_data!")
        n_fft, hop, wtype, match_stride, _ = self._resolve(window_length, hop_length, window_type, match_stride)
        # a pending STFT-domain edit rides on the inverse kernel's spectrum load when that kernel takes this transform;
        # otherwise reading stft_data below materialises it
        pending = self._pending_edit
        Xraw = self._stft_data
        fuse = (pending is not None and kernels.is_native(torch.view_as_real(Xraw)) and not Xraw.requires_grad
                and Xraw.shape[-2] == n_fft // 2 + 1 and kernels.istft_edit_supported(n_fft, hop))
        if fuse:
            window = self.get_window(wtype, n_fft, str(Xraw.device))
            nb, nch, nf, nt = Xraw.shape
            right_pad, pad = self.compute_stft_padding(n_fft, hop, match_stride)
            if length is None:
                length = self.original_signal_length + 2 * pad + right_pad
            edge = 2 if match_stride else 0
            x = kernels.istft(Xraw, window, n_fft, hop, int(length), lead=edge, trail=edge, edit=pending.fused)
            x = x.reshape(nb, nch, -1)
            if match_stride:
                x = x[..., pad: -(pad + right_pad)]
            self.audio_data = x
            return self
        window = self.get_window(wtype, n_fft, str(self.stft_data.device))
        nb, nch, nf, nt = self.stft_data.shape
        X = self.stft_data.reshape(nb * nch, nf, nt)
        right_pad, pad = self.compute_stft_padding(n_fft, hop, match_stride)
        if length is None:
            length = self.original_signal_length + 2 * pad + right_pad
        Xd = self.stft_data
        if (Xd.is_cuda and Xd.dtype == torch.complex64 and Xd.requires_grad and torch.is_grad_enabled()
                and not match_stride and kernels.istft_adjoint_supported(n_fft, hop)
               ):
            x = _NativeIstft.apply(Xd, window, n_fft, hop, int(length))
        elif kernels.is_native(torch.view_as_real(self.stft_data)) and kernels.stft_native_supported(n_fft):
            edge = 2 if match_stride else 0   # the two frames per side the forward transform dropped
            x = kernels.istft(self.stft_data, window, n_fft, hop, int(length), lead=edge, trail=edge)
        else:
            if match_stride:
                X = torch.nn.functional.pad(X, (2, 2))
            x = torch.istft(X, n_fft=n_fft, hop_length=hop, window=window, length=length, center=True)
        x = x.reshape(nb, nch, -1)
        if match_stride:
            x = x[..., pad: -(pad + right_pad)]
        self.audio_data = x
        return self

    # ------------------------------------------------------------- mel, mfcc
    @staticmethod
    @functools.lru_cache(None)
    def get_mel_filters(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax: float = None):
        """(n_mels, 1 + n_fft/2) float32 Slaney mel basis (audio_signal.py:1298-1331)."""
        return tables.mel_filters_np(sr, n_fft, n_mels, fmin, fmax)

    def mel_spectrogram(self, n_mels: int = 80, mel_fmin: float = 0.0, mel_fmax: float = None, **kwargs):
        """|STFT| projected on a mel basis -> (B, C, n_mels, N)
        (audio_signal.py:1333-1369).  Also refreshes ``stft_data`` exactly as
        the reference's internal ``self.stft(**kwargs)`` does."""
        n_fft, hop, wtype, match_stride, padding_type = self._resolve(
            kwargs.get("window_length"), kwargs.get("hop_length"), kwargs.get("window_type"),
            kwargs.get("match_stride"), kwargs.get("padding_type"))
        audio = self.audio_data
        units = None
        if kernels.stft_fused_supported(n_fft) and audio.is_cuda:
            units = tables.mel_units_or_none(self.sample_rate, n_fft, n_mels, mel_fmin, mel_fmax, audio.device)
        banded = False
        if (units is None and kernels.is_native(audio) and not kernels.stft_fused_supported(n_fft)
                and kernels.stft_native_supported(n_fft) and n_fft <= 8192):
            # generic sizes (4096 @ 96 kHz, 8192 @ 192 kHz, 400 / 1200 / 1920 ...): banded mel fused into the
            # mixed-radix kernel -- no second pass over stft_data, no dense matmul
            units = tables.mel_bands(self.sample_rate, n_fft, n_mels, mel_fmin, mel_fmax, audio.device)
            banded = True
        if kernels.is_native(audio) and units is not None:
            dev = audio.device
            info, w = units
            window = self.get_window(wtype, n_fft, str(dev))
            right_pad, pad = self.compute_stft_padding(n_fft, hop, match_stride)
            try:
                old_shape = self._release_stft_data()        # (released before the kernel runs: see stft())
                X, mel = kernels.stft_mel(audio, window, n_fft, hop, pad=pad, right_pad=right_pad,
                                          padding_type=padding_type, match_stride=match_stride,
                                          mel=(info, w, n_mels))
                self._warn_if_shape_changed(old_shape, X)
                self.stft_data = X
                return mel
            except _native.NativeUnsupported:
                # the TILED generic kernel is the only one with a fused mel stage; inputs it does not take (a clip
                # shorter than its tile: T < n_fft + (frames per tile - 1) hop, fewer frames than a tile holds, chunk
                # tables past 160 KB of LDS for very many bands at 8192) keep the route they had before the fusion:
                # native stft() + the dense basis on the native spectrum, below
                if not banded:
                    raise
                units = None
        if units is not None and _native_autograd_ok(audio, n_fft, hop, match_stride) and not kwargs.get("padding_type"):
            dev = audio.device
            units = tuple(units) + (n_mels,)
            basis = torch.from_numpy(self.get_mel_filters(self.sample_rate, n_fft, n_mels, mel_fmin, mel_fmax)).to(dev)
            bin_table = None
            if kernels.stft_mel_adjoint_supported(n_fft, hop, n_mels):
                bin_table = tables.mel_bin_table(self.sample_rate, n_fft, n_mels, mel_fmin, mel_fmax, dev)
            X, mel = _NativeStftMel.apply(audio, self.get_window(wtype, n_fft, str(dev)), n_fft, hop, units, basis,
                                          bin_table)
            self.stft_data = X
            return mel
        X = self.stft(**kwargs)
        magnitude = torch.abs(X)
        nf = magnitude.shape[2]
        basis = self.get_mel_filters(sr=self.sample_rate, n_fft=2 * (nf - 1), n_mels=n_mels,
                                     fmin=mel_fmin, fmax=mel_fmax)
        basis = torch.from_numpy(basis).to(self.device)
        mel = magnitude.transpose(2, -1) @ basis.T
        return mel.transpose(-1, 2)

    @staticmethod
    @functools.lru_cache(None)
    def get_dct(n_mfcc: int, n_mels: int, norm: str = "ortho", device: str = None):
        """DCT-II matrix (n_mels, n_mfcc) (audio_signal.py:1371-1396)."""
        return torch.from_numpy(tables.dct_np(n_mfcc, n_mels, norm)).to(device)

    def mfcc(self, n_mfcc: int = 40, n_mels: int = 80, log_offset: float = 1e-6, **kwargs):
        """log-mel -> DCT (audio_signal.py:1398-1426)."""
        mel = self.mel_spectrogram(n_mels, **kwargs)
        mel = torch.log(mel + log_offset)
        dct = self.get_dct(n_mfcc, n_mels, "ortho", self.device)
        return (mel.transpose(-1, -2) @ dct).transpose(-1, -2)

    # -------------------------------------------------------- magnitude/phase
    @property
    def magnitude(self):
        if self.stft_data is None:
            self.stft()
        return torch.abs(self.stft_data)

    @magnitude.setter
    def magnitude(self, value):
        self.stft_data = value * torch.exp(1j * self.phase)

    def log_magnitude(self, ref_value: float = 1.0, amin: float = 1e-5, top_db: float = 80.0):
        """10 log10(|X|^2) relative to ``ref_value`` with a floor ``top_db``
        below the global peak (audio_signal.py:1457-1487)."""
        mag = self.magnitude
        amin = amin ** 2
        log_spec = 10.0 * torch.log10(mag.pow(2).clamp(min=amin))
        log_spec -= 10.0 * np.log10(np.maximum(amin, ref_value))
        if top_db is not None:
            log_spec = torch.maximum(log_spec, log_spec.max() - top_db)
        return log_spec

    @property
    def phase(self):
        if self.stft_data is None:
            self.stft()
        return torch.angle(self.stft_data)

    @phase.setter
    def phase(self, value):
        self.stft_data = self.magnitude * torch.exp(1j * value)
