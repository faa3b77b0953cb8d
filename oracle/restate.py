"""Stand-alone CPU restatement of the audiotools hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): this is the checker
that travels to the GPU box (``/root/reference`` does not exist there).  Every
function cites the reference lines it follows and uses the same torch-CPU /
scipy primitives the reference calls, plus the leaf restatements in
``oracle/leaves`` for the third-party pieces.  It is validated against the
shim-imported unmodified reference in ``tests/test_oracle_vs_reference.py``
(build container) and against the committed fixtures in ``tests/golden``.

Functions take and return plain torch CPU tensors, shapes as in the reference:
audio (B, C, T) float32; spectra (B, C, F, N) complex64.
"""
import math

import numpy as np
import scipy.signal
import torch
import torch.nn.functional as F

from . import cport
from .leaves import julius_leaf, misc_leaves, pyloudnorm_leaf


# --------------------------------------------------------------------- STFT
def get_window(window_type: str, window_length: int) -> torch.Tensor:
    """audio_signal.py:1030-1038."""
    if window_type == "average":
        w = np.ones(window_length) / window_length
    elif window_type == "sqrt_hann":
        w = np.sqrt(scipy.signal.get_window("hann", window_length))
    else:
        w = scipy.signal.get_window(window_type, window_length)
    return torch.from_numpy(w).float()


def default_stft_params(sample_rate: int):
    """audio_signal.py:1066-1070."""
    win = int(2 ** (np.ceil(np.log2(0.032 * sample_rate))))
    return win, win // 4, "hann", False, "reflect"


def stft_padding(T: int, window_length: int, hop_length: int, match_stride: bool):
    """audio_signal.py:1109-1121 -> (right_pad, pad)."""
    if match_stride:
        assert hop_length == window_length // 4
        return math.ceil(T / hop_length) * hop_length - T, (window_length - hop_length) // 2
    return 0, 0


def stft(audio: torch.Tensor, window_length: int, hop_length: int, window_type: str = "hann",
         match_stride: bool = False, padding_type: str = "reflect") -> torch.Tensor:
    """audio_signal.py:1185-1210."""
    B, C, T = audio.shape
    window = get_window(window_type, window_length)
    right_pad, pad = stft_padding(T, window_length, hop_length, match_stride)
    x = F.pad(audio, (pad, pad + right_pad), padding_type)
    X = torch.stft(x.reshape(-1, x.shape[-1]), n_fft=window_length, hop_length=hop_length, window=window,
                   return_complex=True, center=True)
    X = X.reshape(B, C, X.shape[1], X.shape[2])
    if match_stride:
        X = X[..., 2:-2]
    return X


def stft_f64_direct(audio: np.ndarray, window_length: int, hop_length: int, window_type: str = "hann"):
    """Independent float64 O(N^2)-free KAT: reflect pad n_fft/2, frame t covers
    padded samples [t*hop, t*hop+n_fft), X[f,t] = sum_n w[n] x[t*hop+n] e^{-2 pi i f n/n_fft}.
    Uses numpy's float64 rfft on explicitly built frames (no torch.stft involved)."""
    x = np.asarray(audio, dtype=np.float64)
    B, C, T = x.shape
    n = window_length
    w = get_window(window_type, n).numpy().astype(np.float64)
    xp = np.pad(x, ((0, 0), (0, 0), (n // 2, n // 2)), mode="reflect")
    N = 1 + T // hop_length
    frames = np.stack([xp[..., t * hop_length: t * hop_length + n] for t in range(N)], axis=-2)
    return np.fft.rfft(frames * w, axis=-1).transpose(0, 1, 3, 2)  # (B, C, F, N)


def istft(X: torch.Tensor, window_length: int, hop_length: int, window_type: str, match_stride: bool,
          original_length: int, length: int = None) -> torch.Tensor:
    """audio_signal.py:1266-1294."""
    window = get_window(window_type, window_length)
    nb, nch, nf, nt = X.shape
    Xr = X.reshape(nb * nch, nf, nt)
    right_pad, pad = stft_padding(original_length, window_length, hop_length, match_stride)
    if length is None:
        length = original_length + 2 * pad + right_pad
    if match_stride:
        Xr = F.pad(Xr, (2, 2))
    x = torch.istft(Xr, n_fft=window_length, hop_length=hop_length, window=window, length=length, center=True)
    x = x.reshape(nb, nch, -1)
    if match_stride:
        x = x[..., pad: -(pad + right_pad)]
    return x


def mel_basis(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax: float = None) -> np.ndarray:
    """audio_signal.py:1323-1331 via the librosa leaf."""
    return misc_leaves.librosa_mel(sr=sr, n_fft=n_fft, n_mels=n_mels, fmin=fmin, fmax=fmax)


def mel_spectrogram(X: torch.Tensor, sample_rate: int, n_mels: int = 80, mel_fmin: float = 0.0,
                    mel_fmax: float = None) -> torch.Tensor:
    """audio_signal.py:1355-1369 (from an STFT)."""
    magnitude = torch.abs(X)
    nf = magnitude.shape[2]
    basis = torch.from_numpy(mel_basis(sample_rate, 2 * (nf - 1), n_mels, mel_fmin, mel_fmax))
    mel = magnitude.transpose(2, -1) @ basis.T
    return mel.transpose(-1, 2)


def mfcc(mel: torch.Tensor, n_mfcc: int = 40, log_offset: float = 1e-6) -> torch.Tensor:
    """audio_signal.py:1420-1426."""
    n_mels = mel.shape[2]
    dct = misc_leaves.create_dct(n_mfcc, n_mels, "ortho")
    return (torch.log(mel + log_offset).transpose(-1, -2) @ dct).transpose(-1, -2)


# ----------------------------------------------------------------- loudness
def weighting_filters(rate: int, filter_class: str = "K-weighting"):
    """loudness.py:253-260 -> list of (b[3], a[3], passband_gain), float64."""
    m = pyloudnorm_leaf.Meter(rate, filter_class)
    return [(f.b, f.a, f.passband_gain) for f in m._filters.values()]


def apply_weighting(audio: torch.Tensor, rate: int, filter_class: str = "K-weighting") -> torch.Tensor:
    """loudness.py:102-126 (the CPU/IIR branch): float32 DF-I biquad cascade on (B, C, T)."""
    x = audio.float()
    for b, a, g in weighting_filters(rate, filter_class):
        y = misc_leaves.lfilter(x, torch.from_numpy(a).float(), torch.from_numpy(b).float(), clamp=False)
        x = g * y
    return x


def integrated_loudness(audio: torch.Tensor, rate: int, filter_class: str = "K-weighting",
                        block_size: float = 0.400) -> torch.Tensor:
    """loudness.py:176-247 on (B, C, T) input (the reference permutes to (B, T, C) and back)."""
    B, C, T = audio.shape
    y = apply_weighting(audio, rate, filter_class)
    G = torch.from_numpy(np.array([1.0, 1.0, 1.0, 1.41, 1.41]))
    T_g = block_size
    Gamma_a = -70.0
    K = int(T_g * rate)
    S = int(T_g * rate * (1.0 - 0.75))
    unfolded = julius_leaf.unfold(y, K, S)                    # (B, C, nblk, K)
    z = (1.0 / (T_g * rate)) * unfolded.square().sum(3)       # (B, C, nblk) float32
    l = -0.691 + 10.0 * torch.log10((G[None, :C, None] * z).sum(1, keepdim=True))
    l = l.expand_as(z)
    z = z.clone()
    z[l <= Gamma_a] = 0
    masked = l > Gamma_a
    z_avg = z.sum(2) / masked.sum(2)
    Gamma_r = -0.691 + 10.0 * torch.log10((z_avg * G[None, :C]).sum(-1)) - 10.0
    Gamma_r = Gamma_r[:, None, None].expand(B, C, l.shape[-1])
    z[l <= Gamma_a] = 0
    z[l <= Gamma_r] = 0
    masked = (l > Gamma_a) * (l > Gamma_r)
    z_avg = z.sum(2) / masked.sum(2)
    z_avg = torch.where(z_avg.isnan(), torch.zeros_like(z_avg), z_avg)
    z_avg[z_avg == float("inf")] = float(np.finfo(np.float32).max)
    z_avg[z_avg == -float("inf")] = float(np.finfo(np.float32).min)
    LUFS = -0.691 + 10.0 * torch.log10((G[None, :C] * z_avg).sum(1))
    return LUFS.float()


def loudness(audio: torch.Tensor, rate: int, filter_class: str = "K-weighting",
             block_size: float = 0.400) -> torch.Tensor:
    """loudness.py:300-320: pad to 0.5 s, measure, clamp at -70."""
    T = audio.shape[-1]
    if T / rate < 0.5:
        audio = F.pad(audio, (0, int((0.5 - T / rate) * rate)))
    lufs = integrated_loudness(audio, rate, filter_class, block_size)
    return torch.maximum(lufs, torch.full_like(lufs, -70.0))


def loudness_f64(audio: np.ndarray, rate: int, filter_class: str = "K-weighting",
                 block_size: float = 0.400) -> np.ndarray:
    """Independent float64 meter with the reference's block semantics (ceil + zero-padded
    tail, loudness.py:164-174) -- scipy lfilter in float64, numpy gating."""
    x = np.asarray(audio, dtype=np.float64)
    B, C, T = x.shape
    y = x
    for b, a, g in weighting_filters(rate, filter_class):
        y = g * scipy.signal.lfilter(b, a, y, axis=-1)
    K = int(block_size * rate)
    S = int(block_size * rate * 0.25)
    nblk = math.ceil((max(T, K) - K) / S) + 1
    yp = np.pad(y, ((0, 0), (0, 0), (0, (nblk - 1) * S + K - T)))
    z = np.stack([np.square(yp[..., j * S: j * S + K]).sum(-1) for j in range(nblk)], -1) / (block_size * rate)
    G = np.array([1.0, 1.0, 1.0, 1.41, 1.41])[:C]
    out = np.zeros(B)
    with np.errstate(divide="ignore", invalid="ignore"):
        for i in range(B):
            l = -0.691 + 10 * np.log10((G[:, None] * z[i]).sum(0))
            ga = l > -70.0
            if not ga.any():
                out[i] = -np.inf
                continue
            gr = -0.691 + 10 * np.log10((G * z[i][:, ga].mean(1)).sum()) - 10.0
            keep = ga & (l > gr)
            zz = z[i][:, keep].mean(1) if keep.any() else np.zeros(C)
            out[i] = -0.691 + 10 * np.log10((G * zz).sum())
    return out


# ------------------------------------------------------- filters / resample
def resample(audio: torch.Tensor, old_sr: int, new_sr: int) -> torch.Tensor:
    """audio_signal.py:730-734."""
    if old_sr == new_sr:
        return audio
    return julius_leaf.resample_frac(audio, int(old_sr), int(new_sr))


def low_pass(audio: torch.Tensor, cutoffs, sample_rate: int, zeros: int = 51) -> torch.Tensor:
    """dsp.py:173-181 (per-item loop, float32 cutoff tensor)."""
    B = audio.shape[0]
    c = _ensure(cutoffs, 2, B) / sample_rate
    out = torch.empty_like(audio)
    for i, cutoff in enumerate(c):
        out[i] = julius_leaf.LowPassFilter(cutoff.cpu(), zeros=zeros)(audio[i])
    return out


def high_pass(audio: torch.Tensor, cutoffs, sample_rate: int, zeros: int = 51) -> torch.Tensor:
    """dsp.py:205-213."""
    B = audio.shape[0]
    c = _ensure(cutoffs, 2, B) / sample_rate
    out = torch.empty_like(audio)
    for i, cutoff in enumerate(c):
        out[i] = julius_leaf.HighPassFilter(cutoff.cpu(), zeros=zeros)(audio[i])
    return out


def mel_filterbank(audio: torch.Tensor, sample_rate: int, n_bands: int) -> torch.Tensor:
    """effects.py:399-403 -> (B, C, T, n_bands)."""
    fb = julius_leaf.SplitBands(sample_rate, n_bands).float()
    return fb(audio).permute(1, 2, 3, 0)


def equalizer(audio: torch.Tensor, sample_rate: int, db: torch.Tensor) -> torch.Tensor:
    """effects.py:420-432."""
    db = _ensure(db)
    n_bands = db.shape[-1]
    fbank = mel_filterbank(audio, sample_rate, n_bands)
    if db.ndim == 2:
        if db.shape[0] != 1:
            assert db.shape[0] == fbank.shape[0]
    else:
        db = db.unsqueeze(0)
    weights = (10 ** db).float()
    return (fbank * weights[:, None, None, :]).sum(-1)


def convolve(audio: torch.Tensor, ir: torch.Tensor, start_at_max: bool = True) -> torch.Tensor:
    """effects.py:85-121 (literal, including the delta FFTs)."""
    T = audio.shape[-1]
    pad_len = T - ir.shape[-1]
    ir = F.pad(ir, (0, pad_len)) if pad_len > 0 else ir[..., :T]
    if start_at_max:
        idx = ir.abs().argmax(axis=-1)
        rolled = torch.zeros_like(ir)
        for i in range(ir.shape[0]):
            rolled[i] = torch.roll(ir[i], -idx[i].item(), -1)
        ir = rolled
    delta = torch.zeros_like(ir)
    delta[..., 0] = 1
    delta_fft = torch.fft.rfft(delta, T)
    other_fft = torch.fft.rfft(ir, T)
    self_fft = torch.fft.rfft(audio, T)
    convolved = torch.fft.irfft(other_fft * self_fft, T)
    delta_audio = torch.fft.irfft(other_fft * delta_fft, T)
    delta_max = delta_audio.abs().max(dim=-1, keepdims=True)[0]
    return convolved * (1 / delta_max.clamp(1e-5))


def alter_drr(ir: torch.Tensor, sample_rate: int, drr) -> torch.Tensor:
    """effects.py:540-647 for mono impulse responses (B, 1, T): decompose_ir (early span = +-2.5 ms
    around the arg-max; the "Hann window over the early span" is get_window("hann", n) with
    n = window_idx.shape[-1] == 1 of the (n, 1) nonzero() result, i.e. all ones -- reproduced as
    written, effects.py:569-573), solve_alpha (eq. 5), recombination, ensure_max_of_audio."""
    B = ir.shape[0]
    drr = _ensure(drr, 2, B).to(torch.float32)
    td = torch.argmax(ir, dim=-1, keepdim=True)
    t0 = int(sample_rate * 0.0025)
    idx = torch.arange(ir.shape[-1])[None, None, :].expand(B, -1, -1)
    early_idx = (idx >= td - t0) * (idx <= td + t0)
    early = torch.zeros_like(ir)
    early[early_idx] = ir[early_idx]
    late = torch.zeros_like(ir)
    late[~early_idx] = ir[~early_idx]
    window = torch.zeros_like(ir)
    for i in range(B):
        window_idx = early_idx[i, 0].nonzero()
        window[i, ..., window_idx] = get_window("hann", window_idx.shape[-1])
    wd = window
    e_sq, l_sq = early ** 2, late ** 2
    a = ((wd ** 2) * e_sq).sum(dim=-1)
    b = (2 * (1 - wd) * wd * e_sq).sum(dim=-1)
    c = (((1 - wd) ** 2) * e_sq).sum(dim=-1) - torch.pow(10, drr / 10) * l_sq.sum(dim=-1)
    expr = ((b ** 2) - 4 * a * c).sqrt()
    alpha = torch.maximum((-b - expr) / (2 * a), (-b + expr) / (2 * a))
    min_alpha = late.abs().max(dim=-1)[0] / early.abs().max(dim=-1)[0]
    alpha = torch.maximum(alpha, min_alpha)[..., None]
    out = alpha * window * early + ((1 - window) * early) + late
    peak = out.abs().max(dim=-1, keepdim=True)[0]          # ensure_max_of_audio, effects.py:213-238
    return out * torch.where(peak > 1.0, 1.0 / peak, torch.ones_like(peak))


def apply_ir(audio: torch.Tensor, ir: torch.Tensor, sample_rate: int, drr=None, ir_eq=None) -> torch.Tensor:
    """effects.py:155-179 with use_original_phase=False: IR EQ -> DRR -> convolve -> restore the
    input peak."""
    if ir_eq is not None:
        ir = equalizer(ir, sample_rate, ir_eq)
    if drr is not None:
        ir = alter_drr(ir, sample_rate, drr)
    max_spk = audio.abs().max(dim=-1, keepdims=True).values
    y = convolve(audio, ir)
    max_tr = y.abs().max(dim=-1, keepdims=True).values
    return y * (max_spk.clamp(1e-8) / max_tr.clamp(1e-8))


def log_magnitude(X: torch.Tensor, ref_value: float = 1.0, amin: float = 1e-5, top_db: float = 80.0) -> torch.Tensor:
    """audio_signal.py:1457-1487 (the top_db floor uses the maximum over the WHOLE batch tensor)."""
    magnitude = X.abs()
    amin = amin ** 2
    log_spec = 10.0 * torch.log10(magnitude.pow(2).clamp(min=amin))
    log_spec -= 10.0 * np.log10(np.maximum(amin, ref_value))
    if top_db is not None:
        log_spec = torch.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def mix(audio: torch.Tensor, other: torch.Tensor, sample_rate: int, snr=10, other_eq=None) -> torch.Tensor:
    """effects.py:27-64: pad/truncate ``other`` to the signal length, optional EQ, bring it to
    ``loudness(self) - snr`` LUFS, add."""
    B = audio.shape[0]
    snr = _ensure(snr).to(torch.float32)
    T = audio.shape[-1]
    pad_len = max(0, T - other.shape[-1])
    other = F.pad(other, (0, pad_len))[..., :T]
    if other_eq is not None:
        other = equalizer(other, sample_rate, other_eq)
    tgt = loudness(audio, sample_rate) - snr
    gain = torch.exp((tgt - loudness(other, sample_rate)) * (np.log(10) / 20))
    return audio + other * gain[:, None, None]


def phase_vocoder_f64(X: np.ndarray, p: int, q: int, hop: int) -> np.ndarray:
    """Checker for the device phase vocoder (csrc/vocoder.hip): float64, explicit loops over output
    frames.  PARITY UNPINNED against the reference: effects.py:247-309 pipes the audio through CPU
    libsox ("tempo" / "pitch"), whose output is not reproducible here (no sox, no fixtures; the
    reference's own tests only compare batched with single, tests/core/test_effects.py:156-181).
    The algorithm restated is the published phase vocoder as torchaudio.functional.phase_vocoder
    formulates it (X: (..., F, N) complex), with the rate as the rational p/q."""
    X = np.asarray(X, dtype=np.complex128)
    F_bins, N = X.shape[-2], X.shape[-1]
    n_out = (N * q + p - 1) // p
    Xp = np.concatenate([X, np.zeros(X.shape[:-1] + (2,), X.dtype)], axis=-1)
    adv = np.pi * hop * np.arange(F_bins) / (F_bins - 1)
    out = np.zeros(X.shape[:-1] + (n_out,), np.complex128)
    acc = np.angle(X[..., 0])
    for k in range(n_out):
        j, rem = divmod(k * p, q)
        alpha = rem / q
        x0, x1 = Xp[..., j], Xp[..., j + 1]
        mag = alpha * np.abs(x1) + (1 - alpha) * np.abs(x0)
        out[..., k] = mag * np.exp(1j * acc)
        d = np.angle(x1) - np.angle(x0) - adv
        d = d - 2 * np.pi * np.round(d / (2 * np.pi))
        acc = acc + d + adv
    return out


def _ensure(x, ndim=None, batch_size=None):
    """core/util.py:56-89."""
    if not torch.is_tensor(x):
        x = torch.as_tensor(x)
    if ndim is not None:
        while x.ndim < ndim:
            x = x.unsqueeze(-1)
    if batch_size is not None and x.shape[0] != batch_size:
        shape = list(x.shape)
        shape[0] = batch_size
        x = x.expand(*shape)
    return x
