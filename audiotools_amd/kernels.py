"""Thin Python launchers for the ``at_*`` C-ABI entry points.

Each function takes torch tensors that already live on a HIP device,
allocates the outputs/workspaces with ``torch.empty`` (torch owns all memory,
SURVEY.md 8(b) "Ownership"), and launches on torch's current stream.  No
function here computes anything on the host beyond shapes.
"""
import math

import torch

from . import _native, tables

PAD_MODES = {"reflect": 0, "constant": 1, "replicate": 2, "circular": 3}


def is_native(t: torch.Tensor) -> bool:
    """Dispatch rule: HIP tensor, float32, no autograd."""
    return t.is_cuda and t.dtype == torch.float32 and not (t.requires_grad and torch.is_grad_enabled())


def _require_native_ok(t):
    if not t.is_cuda:
        raise _native.NativeError("native kernels need a HIP device tensor")


def stft_frames(T: int, n_fft: int, hop: int, pad: int, right_pad: int, match_stride: bool):
    """(first computed frame, number of output frames) of the reference's
    stft(): torch.stft(center=True) on the outer-padded signal gives
    1 + T2//hop frames; match_stride drops 2 on each side
    (audio_signal.py:1203-1209)."""
    T2 = T + 2 * pad + right_pad
    n_total = 1 + T2 // hop
    if match_stride:
        return 2, max(n_total - 4, 0)
    return 0, n_total


def stft_native_supported(n_fft: int) -> bool:
    return bool(_native.lib().at_stft_native_supported(int(n_fft)))


def stft_mel(audio: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, *, pad: int = 0,
             right_pad: int = 0, padding_type: str = "reflect", match_stride: bool = False,
             want_stft: bool = True, mel=None):
    """Fused STFT (+ mel).  ``audio`` (B, C, T) float32 HIP tensor.

    ``mel`` is ``None`` or a tuple ``(unit_info, unit_w, n_mels)`` of
    device tables from :func:`tables.mel_units`.

    Returns ``(stft, mel_spec)``: ``stft`` is a complex64 tensor of logical
    shape (B, C, F, N) whose memory is bin-contiguous (B, C, N, F) -- the same
    physical layout torch.stft hands the reference -- and ``mel_spec`` is the
    (B, C, n_mels, N) transposed view of a (B, C, N, n_mels) buffer
    (audio_signal.py:1367-1368).
    """
    _require_native_ok(audio)
    if padding_type not in PAD_MODES:
        raise NotImplementedError(f"Unrecognised padding mode {padding_type}")
    B, C, T = audio.shape
    audio = audio.contiguous()
    F = n_fft // 2 + 1
    frame_lo, n_out = stft_frames(T, n_fft, hop, pad, right_pad, match_stride)
    dev = audio.device
    tw = tables.stft_twiddles(n_fft, dev)
    if not want_stft:
        raise NotImplementedError("the fused kernel always produces stft_data")
    stft_buf = torch.empty((B, C, n_out, F), dtype=torch.complex64, device=dev)
    mel_buf = None
    info = w = None
    n_units = n_mels = 0
    if mel is not None:
        info, w, n_mels = mel
        n_units = int(info.shape[0])
        mel_buf = torch.empty((B, C, n_out, n_mels), dtype=torch.float32, device=dev)
    code = _native.lib().at_stft_mel_f32(
        _native.ptr(audio), B * C, T, _native.ptr(window), _native.ptr(tw), n_fft, hop, pad, right_pad,
        PAD_MODES[padding_type], frame_lo, n_out, _native.ptr(stft_buf), _native.ptr(info), _native.ptr(w),
        n_units, n_mels, _native.ptr(mel_buf), _native.current_stream(dev))
    _native.check(code, "at_stft_mel_f32")
    stft = stft_buf.transpose(2, 3) if stft_buf is not None else None
    mel_spec = mel_buf.transpose(2, 3) if mel_buf is not None else None
    return stft, mel_spec


def lufs_block_params(rate: int, block_size: float):
    """(K, S) exactly as loudness.py:165-170 computes them (Python floats)."""
    overlap = 0.75
    step = 1.0 - overlap
    K = int(block_size * rate)
    S = int(block_size * rate * step)
    return K, S


_ws_cache = {}


def _workspace(nbytes: int, device):
    key = (device.type, device.index)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def integrated_loudness(audio_bct: torch.Tensor, rate: int, filter_class: str = "K-weighting",
                        block_size: float = 0.400, floor_db: float = float("nan")) -> torch.Tensor:
    """BS.1770 integrated loudness of a (B, C, T) float32 HIP tensor -> (B,) float32."""
    _require_native_ok(audio_bct)
    B, C, T = audio_bct.shape
    if C > 5:
        raise RuntimeError("BS.1770 channel gains are defined for at most 5 channels (loudness.py:49)")
    audio_bct = audio_bct.contiguous()
    dev = audio_bct.device
    sos, gains = tables.weighting_sos(int(rate), filter_class)
    K, S = lufs_block_params(rate, block_size)
    if K <= 0 or S <= 0:
        raise ValueError("block_size * rate must be at least 4 samples")
    warm = tables.lufs_warmup(sos)
    inv_norm = 1.0 / (block_size * rate)
    lib = _native.lib()
    need = lib.at_lufs_workspace_bytes(B, C, T, K, S)
    if need < 0:
        _native.check(int(need), "at_lufs_workspace_bytes")
    ws = _workspace(int(need), dev)
    out = torch.empty((B,), dtype=torch.float32, device=dev)
    code = lib.at_lufs_f32(_native.ptr(audio_bct), B, C, T, sos.ctypes.data, gains.ctypes.data, len(sos), K, S,
                           inv_norm, floor_db, min(warm, 1 << 30), _native.ptr(out), _native.ptr(ws),
                           ws.numel(), _native.current_stream(dev))
    _native.check(code, "at_lufs_f32")
    return out


def have(symbol: str) -> bool:
    """True when the loaded library exports ``symbol`` (kernels land incrementally)."""
    return hasattr(_native.lib(), symbol)
