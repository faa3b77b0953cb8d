"""Time-domain filtering, resampling, overlap-add windows and STFT-domain
masks of ``AudioSignal`` (reference ``audiotools/core/dsp.py:9-390`` and
``audio_signal.py:716-736``).

The FIR work (per-item windowed-sinc low/high-pass, polyphase resampling) runs
on the HIP kernels of ``csrc/fir.hip`` for HIP float32 tensors; tensors that
need autograd, or live on the CPU, use the equivalent torch formulation below.
"""
import typing

import numpy as np
import torch
import torch.nn.functional as F

from . import kernels, tables, util


def _conv_rows_replicate(x: torch.Tensor, taps: torch.Tensor, half: int) -> torch.Tensor:
    """Cross-correlate every row of x (R, T) with one FIR (L,), replicate padding."""
    y = F.pad(x[:, None], (half, half), mode="replicate")
    return F.conv1d(y, taps.to(x)[None, None])[:, 0]


def lowpass_torch(audio: torch.Tensor, cutoffs: torch.Tensor, zeros: float, highpass: bool) -> torch.Tensor:
    """Torch formulation of the reference's per-item julius low/high-pass
    (dsp.py:173-179 / 205-211); items sharing a cutoff are filtered together."""
    B, C, T = audio.shape
    cut = cutoffs.detach().reshape(B).cpu()
    out = torch.empty_like(audio)
    for c in torch.unique(cut):
        idx = (cut == c).nonzero()[:, 0].to(audio.device)
        taps = tables.lowpass_taps(c, zeros)
        half = (taps.numel() - 1) // 2
        rows = audio[idx].reshape(-1, T)
        low = _conv_rows_replicate(rows, taps, half).reshape(len(idx), C, T)
        out[idx] = (audio[idx] - low) if highpass else low
    return out


def needs_native_grad(audio: torch.Tensor) -> bool:
    """A HIP float32 tensor whose gradient is wanted: the FIR family keeps it on the kernels (forward kernel + the
    adjoint below) instead of the torch formulation."""
    return audio.is_cuda and audio.dtype == torch.float32 and audio.requires_grad and torch.is_grad_enabled()


class _NativeFir(torch.autograd.Function):
    """``kernels.fir_per_item`` (per-item FIR h of odd length L = 2 H + 1, replicate padding, optionally x - FIR(x)) with
    its exact adjoint with respect to the audio, evaluated by the SAME kernel (reference: low_pass / high_pass /
    equalizer must stay differentiable, tests/core/test_grad.py:44-66).
    Forward  y[n] = sum_j h[j] x[clamp(n - H + j)].  With g = dL/dy zero-extended and u[m] = sum_j h[j] g[m + H - j]:
        dL/dx[m] = u[m] (0 < m < T-1),   dL/dx[0] = sum_{m' <= 0} u[m'],   dL/dx[T-1] = sum_{m' >= T-1} u[m'].
    u on [0, T) is the kernel run on g with the taps flipped, minus what its replicate padding adds within H samples of
    either end (g[0] times a suffix sum of h, g[T-1] times a prefix sum); the two folds are dot products of the first /
    last H gradient samples with prefix / suffix sums of h.  All tap sums come from one cumsum of the (B, L) table."""

    @staticmethod
    def forward(ctx, audio, tp, L, highpass):
        ctx.save_for_backward(tp)
        ctx.L, ctx.highpass = int(L), bool(highpass)
        return kernels.fir_per_item(audio.detach(), tp, highpass=bool(highpass), L=int(L))

    @staticmethod
    def backward(ctx, g):
        (tp,) = ctx.saved_tensors
        return fir_adjoint(g, tp, ctx.L, ctx.highpass, lambda a, t: kernels.fir_per_item(a, t, highpass=False, L=ctx.L)), None, None, None


def fir_adjoint(g: torch.Tensor, tp: torch.Tensor, L: int, highpass: bool, fir) -> torch.Tensor:
    """dL/dx of y = FIR_h(x) (replicate padding; ``highpass``: y = x - FIR_h(x)) from g = dL/dy (B, C, T), T >= L.
    ``tp`` (rows, >= L) tap table with rows in (1, B); ``fir(a, table)`` applies a table of this shape with replicate
    padding (the kernel; the CPU test passes a torch formulation)."""
    H = (L - 1) // 2
    g = g.contiguous()
    T = g.shape[-1]
    h = tp[:, :L]
    flipped = torch.zeros_like(tp)
    flipped[:, :L] = h.flip(-1)
    z = fir(g, flipped.contiguous())
    if H > 0:
        pre = torch.cumsum(h.double(), -1)                      # pre[j] = h[0] + .. + h[j]
        tot = pre[:, -1:]
        m = torch.arange(H, device=g.device)
        # replicate padding of g inside the kernel: output m < H got g[0] * sum_{j > m + H} h[j], output T - H + i got
        # g[T-1] * sum_{j <= i} h[j]
        A = (tot - pre[:, H + m]).to(g.dtype)                   # (rows, H)
        Bp = pre[:, m].to(g.dtype)
        z[..., :H] -= g[..., :1] * A[:, None, :]
        z[..., T - H:] -= g[..., -1:] * Bp[:, None, :]
        # folds of the virtual positions outside [0, T) onto the clamped end samples
        P = pre[:, (H - 1 - m)].to(g.dtype)                     # n = 0 .. H-1: sum_{j <= H-1-n} h[j]
        S = (tot - pre[:, (2 * H - 1 - m)]).to(g.dtype)         # n = T-H+i: sum_{j >= 2H-i} h[j]
        z[..., 0] += (g[..., :H] * P[:, None, :]).sum(-1)
        z[..., T - 1] += (g[..., T - H:] * S[:, None, :]).sum(-1)
    return (g - z) if highpass else z


class _NativeResample(torch.autograd.Function):
    """``kernels.resample`` with its exact adjoint with respect to the audio on the same family of kernels: the transposed
    polyphase sum is again a polyphase sum with the two rates swapped (``tables.resample_adjoint_bank``), followed by the
    fold of the replicate padding (reference: ``resample`` stays differentiable, tests/core/test_grad.py:62)."""

    @staticmethod
    def forward(ctx, audio, old_sr, new_sr):
        ctx.rates = (int(old_sr), int(new_sr), int(audio.shape[-1]))
        return kernels.resample(audio.detach(), int(old_sr), int(new_sr))

    @staticmethod
    def backward(ctx, g):
        old_sr, new_sr, T = ctx.rates
        return kernels.resample_adjoint(g.contiguous(), old_sr, new_sr, T), None, None


def fir_native_or_grad(audio: torch.Tensor, tp: torch.Tensor, L: int, highpass: bool = False) -> torch.Tensor:
    """The FIR kernels on a padded tap table (kernels.sinc_taps_native / eq_taps_native), differentiable in ``audio``."""
    if needs_native_grad(audio):
        return _NativeFir.apply(audio, tp, L, highpass)
    return kernels.fir_per_item(audio, tp, highpass=highpass, L=L)


def resample_torch(audio: torch.Tensor, old_sr: int, new_sr: int) -> torch.Tensor:
    """Torch formulation of julius.resample_frac (audio_signal.py:732)."""
    plan = tables.resample_bank(int(old_sr), int(new_sr))
    if plan is None:
        return audio
    bank, old, new, width = plan
    shape = audio.shape
    T = shape[-1]
    x = audio.reshape(-1, T)
    x = F.pad(x[:, None], (width, width + old), mode="replicate")
    ys = F.conv1d(x, bank.to(audio)[:, None], stride=old)
    y = ys.transpose(1, 2).reshape(list(shape[:-1]) + [-1])
    out_len = int(np.floor(new * T / old))
    return y[..., :out_len]


def _broadcasts_to(t, shape) -> bool:
    t = util.ensure_tensor(t)
    if t.ndim == 0 or t.ndim > len(shape):
        return False
    try:
        return tuple(torch.broadcast_shapes(tuple(t.shape), tuple(shape))) == tuple(shape)
    except RuntimeError:
        return False


class SpecEdit:
    """A STFT-domain edit that has been requested but not yet applied: ``apply(X)`` runs the eager kernel (a new
    spectrum), ``fused`` holds the arguments with which ``at_istft_edit_f32`` applies the same edit while the inverse
    transform reads ``X``."""
    __slots__ = ("apply", "fused")

    def __init__(self, apply, fused):
        self.apply, self.fused = apply, fused


def _per_item_native(X, *params) -> bool:
    """The in-place stft_data kernels apply: native spectrum and every parameter is a scalar or
    has one value per item (shape (), (1,), (B,), (B,1,1,1))."""
    if not (kernels.spec_native(X)):
        return False
    B = X.shape[0]
    for p in params:
        t = util.ensure_tensor(p)
        if t.numel() not in (1, B) or (t.numel() == B and t.ndim >= 1 and t.shape[0] != B):
            return False
    return True


class DSPMixin:
    _original_batch_size = None
    _original_num_channels = None
    _padded_signal_length = None

    # ------------------------------------------------------------- resample
    def resample(self, sample_rate: int):
        """Windowed-sinc resampling to ``sample_rate`` (no-op if equal).  As in
        the reference, ``stft_params`` are NOT re-derived for the new rate."""
        if sample_rate == self.sample_rate:
            return self
        audio = self.audio_data
        if kernels.is_native(audio) and kernels.resample_supported(self.sample_rate, int(sample_rate)):
            self.audio_data = kernels.resample(audio, self.sample_rate, int(sample_rate))
        elif (needs_native_grad(audio) and audio.ndim == 3
              and kernels.resample_adjoint_supported(self.sample_rate, int(sample_rate))):
            self.audio_data = _NativeResample.apply(audio, self.sample_rate, int(sample_rate))
        else:
            self.audio_data = resample_torch(audio, self.sample_rate, int(sample_rate))
        self.sample_rate = sample_rate
        return self

    # ------------------------------------------------- windows / overlap-add
    def _preprocess_signal_for_windowing(self, window_duration, hop_duration):
        self._original_batch_size = self.batch_size
        self._original_num_channels = self.num_channels
        window_length = int(window_duration * self.sample_rate)
        hop_length = int(hop_duration * self.sample_rate)
        if window_length % hop_length != 0:
            window_length = (window_length // hop_length) * hop_length
        self.zero_pad(hop_length, hop_length)
        self._padded_signal_length = self.signal_length
        return window_length, hop_length

    def windows(self, window_duration: float, hop_duration: float, preprocess: bool = True):
        """Generator over windows (dsp.py:37-68)."""
        if preprocess:
            window_length, hop_length = self._preprocess_signal_for_windowing(window_duration, hop_duration)
        self.audio_data = self.audio_data.reshape(-1, 1, self.signal_length)
        for b in range(self.batch_size):
            start = 0
            while start + window_length <= self.signal_length:
                yield self[b, ..., start: start + window_length]
                start += hop_length

    def collect_windows(self, window_duration: float, hop_duration: float, preprocess: bool = True):
        """Reshape into a batch of windows (dsp.py:70-108)."""
        if preprocess:
            window_length, hop_length = self._preprocess_signal_for_windowing(window_duration, hop_duration)
        rows = self.audio_data.reshape(-1, self.signal_length)
        if kernels.is_native(rows) and rows.numel() < (1 << 31) and window_length <= self.signal_length:
            # (a window longer than the signal goes on to unfold, which raises as the reference does)
            self.audio_data = kernels.collect_windows(rows, window_length, hop_length)
            return self
        frames = rows.unfold(-1, window_length, hop_length)  # (rows, n, L)
        self.audio_data = frames.reshape(-1, 1, window_length)
        return self

    def overlap_and_add(self, hop_duration: float):
        """Inverse of collect_windows: overlap-add, divide by the overlap count (dsp.py:110-151)."""
        hop_length = int(hop_duration * self.sample_rate)
        window_length = self.signal_length
        nb, nch = self._original_batch_size, self._original_num_channels
        a = self.audio_data
        # the gather takes exactly what fold accepts (single-channel windows, the window count of the padded length);
        # every other shape goes on to the fold formulation, which raises as the reference does
        if (kernels.is_native(a) and a.shape[1] == 1 and a.shape[0] % (nb * nch) == 0 and hop_length > 0
                and window_length <= self._padded_signal_length
                and a.shape[0] // (nb * nch) == (self._padded_signal_length - window_length) // hop_length + 1
                and 2 * hop_length <= self._padded_signal_length):
            # one gather instead of fold + fold(ones) + division + trim; trim(hop, hop) as the reference's last step
            y = kernels.overlap_add(a, nb * nch, hop_length, self._padded_signal_length, hop_length)
            self.audio_data = y.reshape(nb, nch, -1)
            return self
        unfolded = self.audio_data.reshape(nb * nch, -1, window_length).permute(0, 2, 1)
        size = dict(output_size=(1, self._padded_signal_length), kernel_size=(1, window_length),
                    stride=(1, hop_length))
        folded = F.fold(unfolded, **size)
        norm = F.fold(torch.ones_like(unfolded), **size)
        self.audio_data = (folded / norm).reshape(nb, nch, -1)
        self.trim(hop_length, hop_length)
        return self

    # -------------------------------------------------------- low/high pass
    def _sinc_filter(self, cutoffs, zeros, highpass):
        cutoffs = util.ensure_tensor(cutoffs, 2, self.batch_size)
        host = util.host_copy(cutoffs)
        cutoffs = cutoffs / self.sample_rate
        audio = self.audio_data
        if kernels.is_native(audio):
            # the host twin goes through the same float32 division as the device tensor
            host = None if host is None else host.to(cutoffs.dtype) / self.sample_rate
            filtered = kernels.sinc_filter(audio, cutoffs, zeros, highpass, host_cutoffs=host)
        elif needs_native_grad(audio) and not cutoffs.requires_grad:
            # gradient with respect to the audio wanted: forward kernel + its adjoint on the same kernel (_NativeFir)
            host = None if host is None else host.to(cutoffs.dtype) / self.sample_rate
            tp, L = kernels.sinc_taps_native(cutoffs.detach().to(audio.device).reshape(self.batch_size), zeros,
                                             None if host is None else host.reshape(self.batch_size))
            if L <= audio.shape[-1]:
                filtered = _NativeFir.apply(audio, tp, L, highpass)
            else:
                filtered = lowpass_torch(audio, cutoffs, zeros, highpass)
        else:
            filtered = lowpass_torch(audio, cutoffs, zeros, highpass)
        self.audio_data = filtered
        self.stft_data = None
        return self

    def low_pass(self, cutoffs: typing.Union[torch.Tensor, np.ndarray, float], zeros: int = 51):
        """Per-item windowed-sinc low-pass, cutoffs in Hz (dsp.py:153-183)."""
        return self._sinc_filter(cutoffs, zeros, highpass=False)

    def high_pass(self, cutoffs: typing.Union[torch.Tensor, np.ndarray, float], zeros: int = 51):
        """Per-item high-pass = signal - low-pass (dsp.py:185-215)."""
        return self._sinc_filter(cutoffs, zeros, highpass=True)

    # ----------------------------------------------------------- STFT masks
    def _spec_edit(self, eager, fused_args):
        """Apply a per-item edit of a native spectrum -- now, or, inside a SpectralTransform (``_defer_edits``) whose
        inverse transform can take it (hop = n_fft / 4, fused sizes), as a pending edit that ``istft()`` folds into
        its spectrum load (transforms.py:274-286: stft -> edit -> istft; the edit's read + write pass of stft_data
        disappears).  Reading ``stft_data`` before that materialises it."""
        X = self.stft_data                      # (materialises an earlier pending edit: edits compose in order)
        if self._defer_edits:
            n_fft = 2 * (X.shape[-2] - 1)
            p = self.stft_params
            if p.window_length == n_fft and kernels.istft_edit_supported(n_fft, p.hop_length):
                self._pending_edit = SpecEdit(eager, fused_args(X))
                return self
        self.stft_data = eager(X)
        return self

    def mask_frequencies(self, fmin_hz, fmax_hz, val: float = 0.0):
        """Set magnitude AND phase to ``val`` for fmin <= f < fmax (dsp.py:217-261)."""
        if self.stft_data is None:
            self.stft()
        if _per_item_native(self.stft_data, fmin_hz, fmax_hz):
            lo, hi = util.ensure_tensor(fmin_hz, ndim=1), util.ensure_tensor(fmax_hz, ndim=1)
            assert torch.all(lo < hi)
            grid = torch.linspace(0, self.sample_rate / 2, self.stft_data.shape[-2], device=self.device)
            return self._spec_edit(lambda X: kernels.spec_mask(X, 0, lo, hi, grid, val),
                                   lambda X: kernels.mask_edit(1, lo, hi, grid, val, X.shape[0], X.device))
        mag, phase = self.magnitude, self.phase
        fmin_hz = util.ensure_tensor(fmin_hz, ndim=mag.ndim)
        fmax_hz = util.ensure_tensor(fmax_hz, ndim=mag.ndim)
        assert torch.all(fmin_hz < fmax_hz)
        nbins = mag.shape[-2]
        bins_hz = torch.linspace(0, self.sample_rate / 2, nbins, device=self.device)
        bins_hz = bins_hz[None, None, :, None].repeat(self.batch_size, 1, 1, mag.shape[-1])
        mask = (fmin_hz.to(self.device) <= bins_hz) & (bins_hz < fmax_hz.to(self.device))
        mag = mag.masked_fill(mask, val)
        phase = phase.masked_fill(mask, val)
        self.stft_data = mag * torch.exp(1j * phase)
        return self

    def mask_timesteps(self, tmin_s, tmax_s, val: float = 0.0):
        """Set magnitude AND phase to ``val`` for tmin <= t < tmax (dsp.py:263-306)."""
        if self.stft_data is None:
            self.stft()
        if _per_item_native(self.stft_data, tmin_s, tmax_s):
            lo, hi = util.ensure_tensor(tmin_s, ndim=1), util.ensure_tensor(tmax_s, ndim=1)
            assert torch.all(lo < hi)
            grid = torch.linspace(0, self.signal_duration, self.stft_data.shape[-1], device=self.device)
            return self._spec_edit(lambda X: kernels.spec_mask(X, 1, lo, hi, grid, val),
                                   lambda X: kernels.mask_edit(2, lo, hi, grid, val, X.shape[0], X.device))
        mag, phase = self.magnitude, self.phase
        tmin_s = util.ensure_tensor(tmin_s, ndim=mag.ndim)
        tmax_s = util.ensure_tensor(tmax_s, ndim=mag.ndim)
        assert torch.all(tmin_s < tmax_s)
        nt = mag.shape[-1]
        bins_t = torch.linspace(0, self.signal_duration, nt, device=self.device)
        bins_t = bins_t[None, None, None, :].repeat(self.batch_size, 1, mag.shape[-2], 1)
        mask = (tmin_s.to(self.device) <= bins_t) & (bins_t < tmax_s.to(self.device))
        mag = mag.masked_fill(mask, val)
        phase = phase.masked_fill(mask, val)
        self.stft_data = mag * torch.exp(1j * phase)
        return self

    def mask_low_magnitudes(self, db_cutoff, val: float = 0.0):
        """Mask bins whose log-magnitude is below ``db_cutoff`` (dsp.py:308-334)."""
        if self.stft_data is None:
            self.stft()
        if _per_item_native(self.stft_data, db_cutoff):
            cut = util.ensure_tensor(db_cutoff, ndim=1)

            def fused(X):
                B = X.shape[0]
                return {"kind": 4, "cut": cut.reshape(-1).to(X.device, torch.float64).expand(B).contiguous(),
                        "maxpow": kernels.spec_maxpow(X), "top_db": 80.0, "use_top": 1, "val": float(val)}

            return self._spec_edit(lambda X: kernels.spec_mask_lowmag(X, cut, val), fused)
        mag = self.magnitude
        log_mag = self.log_magnitude()
        db_cutoff = util.ensure_tensor(db_cutoff, ndim=mag.ndim).to(self.device)
        self.magnitude = mag.masked_fill(log_mag < db_cutoff, val)
        return self

    def shift_phase(self, shift):
        if self.stft_data is None:
            self.stft()
        if _per_item_native(self.stft_data, shift) and util.ensure_tensor(shift).dtype in (torch.float32, torch.int64):
            sh = util.ensure_tensor(shift, ndim=1)
            return self._spec_edit(lambda X: kernels.spec_phase_shift(X, sh),
                                   lambda X: {"kind": 3, "shift": sh.reshape(-1).to(X.device, torch.float32).expand(X.shape[0]).contiguous()})
        if kernels.spec_native(self.stft_data) and _broadcasts_to(shift, self.stft_data.shape):
            # a full (B, C, F, N) / (C, F, N) shift tensor (CorruptPhase): one tiled pass instead of
            # angle, add, abs, exp, mul
            sh = util.ensure_tensor(shift)
            if sh.dtype in (torch.float32, torch.float64, torch.int64):
                self.stft_data = kernels.spec_polar_elem(self.stft_data, sh)
                return self
        shift = util.ensure_tensor(shift, ndim=self.phase.ndim).to(self.device)
        self.phase = self.phase + shift
        return self

    def corrupt_phase(self, scale):
        if self.stft_data is None:
            self.stft()
        if kernels.spec_native(self.stft_data):
            # same draws as the reference (one torch.randn_like of the phase), applied in one pass
            scale = util.ensure_tensor(scale, ndim=4).to(self.device)
            noise = torch.randn(self.stft_data.shape, dtype=torch.float32, device=self.device)
            self.stft_data = kernels.spec_polar_elem(self.stft_data, scale * noise)
            return self
        scale = util.ensure_tensor(scale, ndim=self.phase.ndim).to(self.device)
        self.phase = self.phase + scale * torch.randn_like(self.phase)
        return self

    def preemphasis(self, coef: float = 0.85):
        """y[n] = x[n-1] - coef * x[n]  (the reference's 3-tap conv, dsp.py:372-390)."""
        kernel = torch.tensor([1, -coef, 0]).view(1, 1, -1).to(self.device)
        x = self.audio_data.reshape(-1, 1, self.signal_length)
        x = F.conv1d(x, kernel, padding=1)
        self.audio_data = x.reshape(*self.audio_data.shape)
        return self
