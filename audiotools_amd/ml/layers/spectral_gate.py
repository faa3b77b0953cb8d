"""Spectral gating noise reduction (reference ``audiotools/ml/layers/spectral_gate.py:10-127``,
after Tim Sainburg's *noisereduce* / the Audacity effect).

Per frequency bin the noise clip gives a threshold (mean + n_std * std of its dB magnitudes over
time); bins of the signal below it are gated; the binary gate is smoothed over frequency and time
with a small tent filter, scaled by ``denoise_amount`` and applied to the complex spectrum, which is
inverted.  ``stft`` / ``istft`` run the HIP kernels for device tensors (sqrt-Hann, hop = n_fft/4:
the fused inverse path); the noise statistics (a small clip) are torch ops, the gate itself -- dB compare, 2-D
tent smoothing, scaling and the complex product -- is ONE native pass over the spectrum (``at_spec_gate_f32``)."""
import torch
import torch.nn.functional as F
from torch import nn

from ... import util
from ...spectral import STFTParams


def _tent(n: int) -> torch.Tensor:
    """0 < ... < 1 > ... > 0 ramp with ``n`` interior points per side: linspace(0, 1, n + 2) up to but
    excluding 1, then linspace(1, 0, n + 2), with the two outer zeros dropped (2 n + 1 taps)."""
    up = torch.linspace(0, 1, n + 2)[:-1]
    down = torch.linspace(1, 0, n + 2)
    return torch.cat([up, down])[1:-1]


class SpectralGate(nn.Module):
    """``SpectralGate(n_freq=3, n_time=5)(signal, noise, denoise_amount=1.0, n_std=3.0,
    win_length=2048, hop_length=512) -> AudioSignal`` (a denoised copy)."""

    def __init__(self, n_freq: int = 3, n_time: int = 5):
        super().__init__()
        tf, tt = _tent(n_freq), _tent(n_time)
        kernel = torch.outer(tf, tt)
        norm = kernel.sum()
        kernel = kernel / norm
        self.register_buffer("smoothing_filter", kernel[None, None])
        # the two 1-D factors of the normalised filter (the native gate kernel applies them separably)
        self.register_buffer("tent_f", tf / norm, persistent=False)
        self.register_buffer("tent_t", tt.clone(), persistent=False)

    def forward(self, audio_signal, nz_signal, denoise_amount: float = 1.0, n_std: float = 3.0, win_length: int = 2048,
                hop_length: int = 512):
        params = STFTParams(win_length, hop_length, "sqrt_hann")
        sig = audio_signal.clone()
        sig.stft_data = None
        sig.stft_params = params
        noise = nz_signal.clone()
        noise.stft_params = params

        # per-bin threshold from the noise clip (dB magnitudes over time)
        noise_db = 20 * noise.magnitude.clamp(1e-4).log10()
        thresh = noise_db.mean(dim=-1, keepdim=True) + n_std * noise_db.std(dim=-1, keepdim=True)

        from ... import kernels
        sig.stft()
        amount = util.ensure_tensor(denoise_amount)
        if (kernels.spec_native(sig.stft_data) and amount.numel() in (1, sig.batch_size) and thresh.shape[0] in (1, sig.batch_size)
                and thresh.shape[1] == sig.num_channels and max(self.tent_f.numel(), self.tent_t.numel()) <= 17):
            # one pass: gate bits, separable tent smoothing and the product (csrc/specedit.hip spec_gate_kernel)
            sig.stft_data = kernels.spec_gate(sig.stft_data, thresh[..., 0], amount, self.tent_f, self.tent_t)
            sig.istft()
            return sig
        sig_db = 20 * sig.magnitude.clamp(1e-4).log10()
        nb, nc, nf, nt = sig_db.shape
        gate = (sig_db < thresh.expand(nb, nc, -1, nt)).float()
        kf, kt = self.smoothing_filter.shape[-2:]
        gate = F.conv2d(gate.reshape(nb * nc, 1, nf, nt), self.smoothing_filter, padding=(kf // 2, kt // 2))
        gate = gate.reshape(nb, nc, nf, nt)
        gate = gate * util.ensure_tensor(denoise_amount, ndim=gate.ndim).to(sig.device)

        sig.stft_data = sig.stft_data * (1 - gate)
        sig.istft()
        return sig
