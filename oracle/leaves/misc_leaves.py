"""Restatements of the remaining small third-party leaves.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

* ``librosa.filters.mel`` (librosa 0.10, unpinned ``setup.py:50``) --
  consumed at ``audiotools/core/audio_signal.py:1323-1331``.  Slaney mel
  scale, ``norm="slaney"``, float32.  **Values are parity-unpinned by the
  reference's own tests** (shape only, ``tests/core/test_audio_signal.py:
  470-486``); pinned here by closed-form properties.
* ``torchaudio.functional.lfilter`` (torchaudio 2.x, unpinned) -- consumed at
  ``audiotools/core/loudness.py:122-124``.  Direct-form-I in float32.
* ``torchaudio.functional.create_dct`` -- ``audio_signal.py:1394``.
* ``flatten_dict.flatten / unflatten`` -- ``core/util.py``, ``data/transforms.py``.
"""
import math

import numpy as np
import torch


# ------------------------------------------------------------ librosa.filters
def _hz_to_mel_slaney(frequencies):
    frequencies = np.asanyarray(frequencies, dtype=np.float64)
    f_min = 0.0
    f_sp = 200.0 / 3
    mels = (frequencies - f_min) / f_sp
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if frequencies.ndim:
        log_t = frequencies >= min_log_hz
        mels[log_t] = min_log_mel + np.log(frequencies[log_t] / min_log_hz) / logstep
    elif frequencies >= min_log_hz:
        mels = min_log_mel + np.log(frequencies / min_log_hz) / logstep
    return mels


def _mel_to_hz_slaney(mels):
    mels = np.asanyarray(mels, dtype=np.float64)
    f_min = 0.0
    f_sp = 200.0 / 3
    freqs = f_min + f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        log_t = mels >= min_log_mel
        freqs[log_t] = min_log_hz * np.exp(logstep * (mels[log_t] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (mels - min_log_mel))
    return freqs


def librosa_mel(*, sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False,
                norm="slaney", dtype=np.float32):
    """librosa.filters.mel (Appendix A.8)."""
    if htk:
        raise NotImplementedError("audiotools never passes htk=True")
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=dtype)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    min_mel = _hz_to_mel_slaney(fmin)
    max_mel = _hz_to_mel_slaney(fmax)
    mel_f = _mel_to_hz_slaney(np.linspace(min_mel, max_mel, n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    if norm == "slaney":
        enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
        weights *= enorm[:, np.newaxis]
    return weights


# ------------------------------------------------------- torchaudio.functional
def librosa_amplitude_to_db(S, ref=1.0, amin=1e-5, top_db=80.0):
    """librosa.core.spectrum.amplitude_to_db (the comparator of the reference's
    tests/core/test_audio_signal.py:459-467): 20 log10(max(amin, |S|)) - 20 log10(max(amin, ref)),
    floored at (max - top_db).  numpy in, numpy out."""
    import numpy as np
    mag = np.abs(np.asarray(S))
    ref_value = ref(mag) if callable(ref) else np.abs(ref)
    power = np.square(mag, out=np.empty_like(mag))
    out = 10.0 * np.log10(np.maximum(amin ** 2, power))
    out -= 10.0 * np.log10(np.maximum(amin ** 2, ref_value ** 2))
    if top_db is not None:
        out = np.maximum(out, out.max() - top_db)
    return out


def lfilter(waveform: torch.Tensor, a_coeffs: torch.Tensor, b_coeffs: torch.Tensor,
            clamp: bool = True, batching: bool = True) -> torch.Tensor:
    """torchaudio.functional.lfilter (Appendix A.7): Direct-Form-I IIR along
    the last axis, zero initial state, in the input dtype.

    Uses the C restatement ``oracle/c/lfilter.c`` when it has been built
    (same DF-I recursion order as torchaudio's ``cpu_lfilter_core_loop``),
    else ``scipy.signal.lfilter`` on the float32 data (DF-II-T; equal to
    ~1e-6 relative)."""
    from .. import cport

    shape = waveform.shape
    x = waveform.detach().reshape(-1, shape[-1]).contiguous()
    a = a_coeffs.detach().to(torch.float64).cpu().numpy()
    b = b_coeffs.detach().to(torch.float64).cpu().numpy()
    if x.dtype == torch.float32 and len(a) == 3 and len(b) == 3 and cport.available():
        y = torch.from_numpy(cport.lfilter_df1_f32(x.cpu().numpy(), b, a))
    else:
        import scipy.signal

        xn = x.cpu().numpy()
        y = scipy.signal.lfilter(b.astype(xn.dtype), a.astype(xn.dtype), xn, axis=-1)
        y = torch.from_numpy(np.ascontiguousarray(y)).to(x.dtype)
    if clamp:
        y = torch.clamp(y, min=-1.0, max=1.0)
    return y.reshape(shape).to(waveform.device)


def create_dct(n_mfcc: int, n_mels: int, norm) -> torch.Tensor:
    """torchaudio.functional.create_dct (Appendix A.9): DCT-II, (n_mels, n_mfcc)."""
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    if norm is None:
        dct *= 2.0
    else:
        assert norm == "ortho"
        dct[0] *= 1.0 / math.sqrt(2.0)
        dct *= math.sqrt(2.0 / float(n_mels))
    return dct.t()


# ---------------------------------------------------------------- flatten_dict
def flatten(d, reducer="tuple", _parent=()):
    out = {}
    for k, v in d.items():
        key = _parent + (k,)
        if isinstance(v, dict) and len(v):
            out.update(flatten(v, reducer, key))
        else:
            out[key] = v
    if not _parent and reducer != "tuple":
        raise NotImplementedError(reducer)
    return out


def unflatten(d, splitter="tuple"):
    out = {}
    for key, v in d.items():
        cur = out
        for k in key[:-1]:
            cur = cur.setdefault(k, {})
        cur[key[-1]] = v
    return out
