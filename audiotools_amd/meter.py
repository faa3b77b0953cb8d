"""BS.1770 loudness meter with the reference's ``Meter`` / ``LoudnessMixin``
interface (``audiotools/core/loudness.py:11-320``), computed by the HIP
kernels in ``csrc/loudness.hip`` for HIP tensors.

Differences from the reference, by design (SURVEY.md 5.9 / north star): on a
GPU the reference silently switches to a 512-tap FIR *approximation* of the
weighting filters (``loudness.py:143-146``); here the GPU path evaluates the
exact IIR cascade, i.e. it reproduces the reference's CPU branch.
``use_fir=True`` on HIP tensors runs the reference's FIR approximation on the block-FFT FIR kernel (round 4).
"""
import copy

import numpy as np
import scipy.signal
import torch
import torch.nn.functional as F

from . import kernels, tables


class Meter(torch.nn.Module):
    """Tensorised ITU-R BS.1770-4 meter.

    Parameters mirror ``loudness.py:34-41``: ``rate``, ``filter_class``
    ("K-weighting", "Fenton/Lee 1", "Fenton/Lee 2", "Dash et al."),
    ``block_size`` (s), ``zeros`` (FIR length of the approximation),
    ``use_fir``.
    """

    def __init__(self, rate: int, filter_class: str = "K-weighting", block_size: float = 0.400,
                 zeros: int = 512, use_fir: bool = False):
        super().__init__()
        self.rate = rate
        self.filter_class = filter_class
        self.block_size = block_size
        self.use_fir = use_fir
        self.zeros = zeros
        self.register_buffer("G", torch.from_numpy(np.array([1.0, 1.0, 1.0, 1.41, 1.41])))
        self._firs = None  # built lazily: only the torch FIR path needs them

    # -- filter design ------------------------------------------------------
    @property
    def filter_class(self):
        return self._filter_class

    @filter_class.setter
    def filter_class(self, value):
        self._sos, self._gains = tables.weighting_sos(int(self.rate), value)
        self._filter_class = value
        self._firs = None

    def _fir_bank(self, device):
        if self._firs is None:
            impulse = np.zeros((self.zeros,))
            impulse[0] = 1.0
            firs = np.stack([scipy.signal.lfilter(s[:3], s[3:], impulse) for s in self._sos])
            self._firs = torch.from_numpy(firs[:, None, ::-1].copy()).float()
        return self._firs.to(device)

    # -- torch (CPU / autograd-free) paths ---------------------------------
    def apply_filter_gpu(self, data: torch.Tensor):
        """FIR approximation (loudness.py:69-100); data (nb, nt, nch)."""
        nb, nt, nch = data.shape
        x = data.permute(0, 2, 1).reshape(nb * nch, 1, nt)
        firs = self._fir_bank(data.device)
        L = firs.shape[-1]
        for i in range(firs.shape[0]):
            x = F.pad(x, (L, L))
            x = F.conv1d(x, firs[i: i + 1])
            x = float(self._gains[i]) * x
            x = x[..., 1: nt + 1]
        return x.reshape(nb, nch, nt).permute(0, 2, 1)[:, :nt, :]

    def apply_filter_cpu(self, data: torch.Tensor):
        """Exact float32 IIR cascade on the host (loudness.py:102-126)."""
        x = data.permute(0, 2, 1)
        xn = x.detach().cpu().numpy().astype(np.float32)
        for s, g in zip(self._sos, self._gains):
            b = s[:3].astype(np.float32)
            a = s[3:].astype(np.float32)
            xn = (np.float32(g) * scipy.signal.lfilter(b, a, xn, axis=-1)).astype(np.float32)
        return torch.from_numpy(xn).to(data.device).permute(0, 2, 1)

    def apply_filter(self, data: torch.Tensor):
        if self.use_fir:
            return self.apply_filter_gpu(data)
        return self.apply_filter_cpu(data)

    def forward(self, data: torch.Tensor):
        return self.integrated_loudness(data)

    def _unfold(self, input_data):
        K, S = kernels.lufs_block_params(self.rate, self.block_size)
        x = input_data.permute(0, 2, 1)
        T = x.shape[-1]
        n_frames = int(np.ceil((max(T, K) - K) / S)) + 1
        x = F.pad(x, (0, (n_frames - 1) * S + K - T))
        return x.unfold(-1, K, S).transpose(-1, -2)  # (nb, nch, K, nblk)

    def integrated_loudness(self, data: torch.Tensor):
        """Integrated gated loudness of ``data`` (nb, nt, nch) -> (nb,) LUFS
        (loudness.py:176-247)."""
        if not torch.is_tensor(data):
            data = torch.from_numpy(data).float()
        else:
            data = data.float()
        x = copy.copy(data)
        if x.ndim < 2:
            x = x.unsqueeze(-1)
        if x.ndim < 3:
            x = x.unsqueeze(0)

        if x.is_cuda and not self.use_fir:
            # HIP path: (nb, nt, nch) -> contiguous (nb, nch, nt) rows for the kernel
            return kernels.integrated_loudness(x.detach().permute(0, 2, 1).contiguous(), self.rate,
                                               self.filter_class, self.block_size)
        if self.use_fir and kernels.is_native(x) and x.shape[-1] <= 5:
            # (a tensor that needs grad keeps the differentiable conv1d formulation below, as the reference's FIR branch is)
            # the reference's FIR approximation (loudness.py:69-100) on the block-FFT FIR kernel + native gating
            firs = self._fir_bank("cpu")[:, 0].flip(-1).double().numpy()        # impulse responses, un-reversed
            return kernels.integrated_loudness_fir(x.permute(0, 2, 1).contiguous(), self.rate, firs,
                                                   [float(g) for g in self._gains], self.block_size)

        nb, nt, nch = x.shape
        x = self.apply_filter(x)
        G = self.G.to(x.device)
        T_g = self.block_size
        Gamma_a = -70.0
        unfolded = self._unfold(x)
        z = (1.0 / (T_g * self.rate)) * unfolded.square().sum(2)
        l = -0.691 + 10.0 * torch.log10((G[None, :nch, None] * z).sum(1, keepdim=True))
        l = l.expand_as(z)

        z_abs = torch.where(l > Gamma_a, z, torch.zeros_like(z))
        n_abs = (l > Gamma_a).sum(2)
        z_avg = z_abs.sum(2) / n_abs
        Gamma_r = -0.691 + 10.0 * torch.log10((z_avg * G[None, :nch]).sum(-1)) - 10.0
        Gamma_r = Gamma_r[:, None, None].expand(nb, nch, l.shape[-1])

        keep = (l > Gamma_a) & (l > Gamma_r)
        z_rel = torch.where(keep, z, torch.zeros_like(z))
        z_avg = z_rel.sum(2) / keep.sum(2)
        z_avg = torch.where(z_avg.isnan(), torch.zeros_like(z_avg), z_avg)
        fmax = float(np.finfo(np.float32).max)
        z_avg = torch.where(z_avg == float("inf"), torch.full_like(z_avg, fmax), z_avg)
        z_avg = torch.where(z_avg == -float("inf"), torch.full_like(z_avg, -fmax), z_avg)
        LUFS = -0.691 + 10.0 * torch.log10((G[None, :nch] * z_avg).sum(1))
        return LUFS.float()


class LoudnessMixin:
    _loudness = None
    MIN_LOUDNESS = -70
    """Minimum loudness possible."""

    def loudness(self, filter_class: str = "K-weighting", block_size: float = 0.400, **kwargs):
        """Integrated loudness (LUFS) per batch item, cached in ``_loudness``
        until ``audio_data`` is reassigned (loudness.py:268-320).  Signals
        shorter than 0.5 s are zero-padded to 0.5 s for the measurement."""
        if self._loudness is not None:
            return self._loudness.to(self.device)
        original_length = self.signal_length
        if self.signal_duration < 0.5:
            pad_len = int((0.5 - self.signal_duration) * self.sample_rate)
            self.zero_pad(0, pad_len)

        audio = self.audio_data
        if kernels.is_native(audio.detach()) and not kwargs.get("use_fir", False):
            loud = kernels.integrated_loudness(audio.detach(), self.sample_rate, filter_class, block_size,
                                               floor_db=float(self.MIN_LOUDNESS))
        else:
            meter = Meter(self.sample_rate, filter_class=filter_class, block_size=block_size, **kwargs)
            meter = meter.to(self.device)
            loud = meter.integrated_loudness(audio.permute(0, 2, 1))
            loud = torch.maximum(loud, torch.ones_like(loud) * self.MIN_LOUDNESS)
        self.truncate_samples(original_length)
        self._loudness = loud
        return self._loudness.to(self.device)
