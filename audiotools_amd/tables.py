"""Host-side design of the small read-only tables the HIP kernels consume.

These are one-off float64 numpy computations (cached), exactly the values the
reference obtains from scipy / librosa / pyloudnorm / julius at
``audio_signal.py:1030-1037`` (window), ``:1323-1331`` (mel basis),
``loudness.py:253-260`` (weighting biquads).  Device copies are cached per
(device, parameters) in ``_device_cache`` -- the counterpart of the
reference's ``functools.lru_cache`` on ``get_window`` / ``get_mel_filters``.
"""
import collections
import functools
import math

import numpy as np
import torch

from . import _native

# ------------------------------------------------------------------- windows


@functools.lru_cache(None)
def window_np(window_type: str, window_length: int) -> np.ndarray:
    """``AudioSignal.get_window`` semantics (audio_signal.py:1030-1038):
    scipy periodic windows, plus "average" and "sqrt_hann"; float32."""
    from scipy import signal

    if window_type == "average":
        w = np.ones(window_length) / window_length
    elif window_type == "sqrt_hann":
        w = np.sqrt(signal.get_window("hann", window_length))
    else:
        w = signal.get_window(window_type, window_length)
    return np.ascontiguousarray(w, dtype=np.float64).astype(np.float32)


# ----------------------------------------------------------------- mel basis
def _slaney_hz_to_mel(f):
    f = np.atleast_1d(np.asarray(f, dtype=np.float64))
    lin = f / (200.0 / 3.0)
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log = 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) / logstep
    return np.where(f >= 1000.0, log, lin)


def _slaney_mel_to_hz(m):
    m = np.atleast_1d(np.asarray(m, dtype=np.float64))
    lin = m * (200.0 / 3.0)
    logstep = np.log(6.4) / 27.0
    log = 1000.0 * np.exp(logstep * (m - 15.0))
    return np.where(m >= 15.0, log, lin)


@functools.lru_cache(None)
def mel_filters_np(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax: float = None) -> np.ndarray:
    """Slaney-scale, slaney-normalised triangular filterbank, float32
    ``(n_mels, 1 + n_fft//2)`` -- what ``librosa.filters.mel(sr=, n_fft=,
    n_mels=, fmin=, fmax=)`` returns for the reference (audio_signal.py:1325)."""
    if fmax is None:
        fmax = float(sr) / 2
    n_bins = 1 + n_fft // 2
    freqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    edges_mel = np.linspace(_slaney_hz_to_mel(fmin)[0], _slaney_hz_to_mel(fmax)[0], n_mels + 2)
    edges = _slaney_mel_to_hz(edges_mel)
    widths = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        rising = -ramps[:-2] / widths[:-1, None]
        falling = ramps[2:] / widths[1:, None]
    tri = np.maximum(0.0, np.minimum(rising, falling)).astype(np.float32)
    enorm = 2.0 / (edges[2: n_mels + 2] - edges[:n_mels])
    # librosa multiplies the float32 triangles by the float64 norm in place
    tri = (tri.astype(np.float64) * enorm[:, None]).astype(np.float32)
    assert tri.shape == (n_mels, n_bins)
    return tri


MEL_UNIT = 16  # bins per mel work unit (matches csrc/stft.hip)


def mel_units_np(basis: np.ndarray):
    """Compress a banded (n_mels, F) filterbank into the unit tables of
    ``at_stft_mel_f32`` with the library's host helper ``at_mel_units_host``:
    ``(unit_info int32 (n_units, 2), unit_w float32 (n_units, 16))``.  A unit is
    one (row of 16 bins, band) pair; zero weights inside a band are kept, so
    any banded real matrix is represented exactly."""
    basis = np.ascontiguousarray(basis, dtype=np.float32)
    n_mels, n_bins = basis.shape
    lib = _native.lib()
    n = lib.at_mel_units_host(basis.ctypes.data, n_mels, n_bins, None, None)
    if n <= 0:
        _native.check(int(n) if n < 0 else -1, "at_mel_units_host")
    info = np.zeros((n, 2), dtype=np.int32)
    w = np.zeros((n, 16), dtype=np.float32)
    n2 = lib.at_mel_units_host(basis.ctypes.data, n_mels, n_bins, info.ctypes.data, w.ctypes.data)
    assert n2 == n
    return info, w


def dct_np(n_mfcc: int, n_mels: int, norm="ortho") -> np.ndarray:
    """DCT-II basis (n_mels, n_mfcc) as ``torchaudio.functional.create_dct``
    (audio_signal.py:1394)."""
    n = np.arange(n_mels, dtype=np.float32)
    k = np.arange(n_mfcc, dtype=np.float32)[:, None]
    dct = np.cos(np.float32(math.pi / n_mels) * (n + np.float32(0.5)) * k).astype(np.float32)
    if norm is None:
        dct *= 2.0
    else:
        dct[0] *= np.float32(1.0 / math.sqrt(2.0))
        dct *= np.float32(math.sqrt(2.0 / n_mels))
    return np.ascontiguousarray(dct.T)


# ------------------------------------------------------- weighting biquads
def _rbj(kind, G, Q, fc, rate):
    """RBJ cookbook biquad as pyloudnorm's IIRfilter designs it; returns
    (b0,b1,b2,a0,a1,a2) normalised by a0."""
    A = 10 ** (G / 40.0)
    w0 = 2.0 * np.pi * (fc / rate)
    al = np.sin(w0) / (2.0 * Q)
    c = np.cos(w0)
    rA = np.sqrt(A)
    if kind == "high_shelf":
        b = [A * ((A + 1) + (A - 1) * c + 2 * rA * al), -2 * A * ((A - 1) + (A + 1) * c),
             A * ((A + 1) + (A - 1) * c - 2 * rA * al)]
        a = [(A + 1) - (A - 1) * c + 2 * rA * al, 2 * ((A - 1) - (A + 1) * c),
             (A + 1) - (A - 1) * c - 2 * rA * al]
    elif kind == "high_pass":
        b = [(1 + c) / 2, -(1 + c), (1 + c) / 2]
        a = [1 + al, -2 * c, 1 - al]
    elif kind == "peaking":
        b = [1 + al * A, -2 * c, 1 - al * A]
        a = [1 + al / A, -2 * c, 1 - al / A]
    else:  # pragma: no cover
        raise ValueError(kind)
    b = np.asarray(b, dtype=np.float64) / a[0]
    a = np.asarray(a, dtype=np.float64) / a[0]
    return np.concatenate([b, a])


_FILTER_CLASSES = {
    # name -> list of (kind, G, Q, fc); order = application order (loudness.py:115)
    "K-weighting": [("high_shelf", 4.0, 1 / np.sqrt(2), 1500.0), ("high_pass", 0.0, 0.5, 38.0)],
    "Fenton/Lee 1": [("high_shelf", 5.0, 1 / np.sqrt(2), 1500.0), ("high_pass", 0.0, 0.5, 130.0),
                     ("peaking", 0.0, 1 / np.sqrt(2), 500.0)],
    "Fenton/Lee 2": [("high_shelf", 4.0, 1 / np.sqrt(2), 1500.0), ("high_pass", 0.0, 0.5, 38.0)],
    "Dash et al.": [("high_pass", 0.0, 0.375, 149.0), ("peaking", -2.93820927, 1.68878655, 1000.0)],
}


@functools.lru_cache(None)
def weighting_sos(rate: int, filter_class: str = "K-weighting"):
    """(sos[nstage,6] float64, gains[nstage] float64) for a BS.1770 meter."""
    if filter_class not in _FILTER_CLASSES:
        raise ValueError("Invalid filter class:", filter_class)
    sos = np.stack([_rbj(k, G, Q, fc, rate) for (k, G, Q, fc) in _FILTER_CLASSES[filter_class]])
    gains = np.ones(len(sos), dtype=np.float64)
    return sos, gains


def lufs_warmup(sos: np.ndarray, tol: float = 1e-9, granule: int = 4) -> int:
    """Samples a filter segment must be started early (from zero state) so
    the discarded transient has decayed below ``tol``: max pole radius ** n
    < tol, rounded up to the 16-byte granule the kernels align a segment's first load to.  (Until round 5 this was
    rounded up to a whole 2048-sample super-block -- 6144 instead of 4136 samples for K-weighting at 44.1 kHz -- although
    the kernels start their super-blocks AT the segment's first sample, not on a 2048 grid: at 64 items, 32 segments of
    13.8 k samples per row, a third of the redundant warm-up filtering for nothing.)"""
    rmax = 0.0
    for row in sos:
        a = np.asarray(row[3:], dtype=np.float32).astype(np.float64)
        rmax = max(rmax, float(np.max(np.abs(np.roots(a)))))
    if rmax <= 0.0:
        return 0
    if rmax >= 1.0:
        return 1 << 30  # unstable/marginal filter: never split rows (start clamps to 0)
    n = math.log(tol) / math.log(rmax)
    return int(math.ceil(n / granule) * granule)


# --------------------------------------------------------------- device cache
_device_cache = {}


def _dev_key(device):
    d = torch.device(device)
    return (d.type, d.index if d.index is not None else (torch.cuda.current_device() if d.type == "cuda" else -1))


def device_table(key, device, builder):
    """Cache ``builder()`` (numpy array or tuple of arrays) on ``device``."""
    k = (key, _dev_key(device))
    hit = _device_cache.get(k)
    if hit is None:
        val = builder()
        if isinstance(val, tuple):
            hit = tuple(torch.from_numpy(np.ascontiguousarray(v)).to(device) for v in val)
        else:
            hit = torch.from_numpy(np.ascontiguousarray(val)).to(device)
        _device_cache[k] = hit
        # per-ratio resampler banks are MB-sized and pitch_shift feeds arbitrary ratios: evict the oldest of them
        if isinstance(key, tuple) and key and key[0] in _EVICTABLE:
            mine = [q for q in _device_cache if isinstance(q[0], tuple) and q[0] and q[0][0] == key[0]]
            for q in mine[:-_EVICT_KEEP]:
                _device_cache.pop(q, None)
    return hit


_EVICTABLE = ("resample_mfma", "resample_grouped", "resample_f16")
_EVICT_KEEP = 16


def stft_twiddles(n_fft: int, device):
    def make():
        out = np.empty(2 * n_fft, dtype=np.float32)
        _native.check(_native.lib().at_stft_twiddles_host(n_fft, out.ctypes.data), "at_stft_twiddles_host")
        return out

    return device_table(("stft_tw", n_fft), device, make)


_LONGCONV_LRU = collections.OrderedDict()
_LONGCONV_MAX = 16


def longconv_tables(T: int, device):
    """Twiddle tables of the four-step convolution of length ``T`` (``at_longconv_tables_host``,
    evaluated in double), on ``device``.  Signal lengths vary from batch to batch, so this cache
    is a bounded LRU (each entry is ~100 KB), unlike the parameter-keyed tables above."""
    k = (int(T), _dev_key(device))
    hit = _LONGCONV_LRU.get(k)
    if hit is None:
        lib = _native.lib()
        n = int(lib.at_longconv_table_floats(int(T)))
        if n < 0:
            _native.check(n, "at_longconv_table_floats")
        host = np.empty(n, dtype=np.float32)
        _native.check(lib.at_longconv_tables_host(int(T), host.ctypes.data, n), "at_longconv_tables_host")
        hit = torch.from_numpy(host).to(device)
        _LONGCONV_LRU[k] = hit
        while len(_LONGCONV_LRU) > _LONGCONV_MAX:
            _LONGCONV_LRU.popitem(last=False)
    else:
        _LONGCONV_LRU.move_to_end(k)
    return hit


def window(window_type: str, window_length: int, device):
    return device_table(("window", window_type, window_length), device,
                        lambda: window_np(window_type, window_length))


def mel_filters(sr, n_fft, n_mels, fmin, fmax, device):
    return device_table(("mel", sr, n_fft, n_mels, fmin, fmax), device,
                        lambda: mel_filters_np(sr, n_fft, n_mels, fmin, fmax))


def mel_units(sr, n_fft, n_mels, fmin, fmax, device):
    return device_table(("mel_units", sr, n_fft, n_mels, fmin, fmax), device,
                        lambda: mel_units_np(mel_filters_np(sr, n_fft, n_mels, fmin, fmax)))


def mel_bands_np(basis: np.ndarray):
    """BANDED form of a dense (n_mels, F) filterbank for the generic-size fused mel stage (``at_mel_bands_host``):
    every row cut to its non-zero span, the span into chunks of 16 bins.  Returns ``(info int32 (n_chunks + 2 n_mels,),
    w float32 (n_chunks, 16))``: the first bin of every chunk, then ``{first chunk, chunk count}`` per band."""
    basis = np.ascontiguousarray(basis, dtype=np.float32)
    n_mels, F = basis.shape
    lib = _native.lib()
    n = int(lib.at_mel_bands_host(basis.ctypes.data, n_mels, F, None, None))
    if n < 0:
        _native.check(n, "at_mel_bands_host")
    info = np.zeros(n + 2 * n_mels, dtype=np.int32)
    w = np.zeros((max(n, 1), 16), dtype=np.float32)
    got = int(lib.at_mel_bands_host(basis.ctypes.data, n_mels, F, info.ctypes.data, w.ctypes.data))
    assert got == n
    return info, w


def mel_bands(sr, n_fft, n_mels, fmin, fmax, device):
    """Device copy of :func:`mel_bands_np` of the Slaney basis (cached)."""
    return device_table(("mel_bands", sr, n_fft, n_mels, fmin, fmax), device,
                        lambda: mel_bands_np(mel_filters_np(sr, n_fft, n_mels, fmin, fmax)))


def mel_bin_table_np(basis: np.ndarray):
    """Per-bin view of a triangular filterbank for the mel backward kernel: bin k is covered by at
    most two bands.  Returns ``(bands int32 (F,) = lo | hi << 16, weights float32 (F, 2))`` or None
    when some bin has more than two non-zero bands (not a shared-edge triangular bank)."""
    n_mels, F = basis.shape
    bands = np.zeros(F, dtype=np.int64)
    w = np.zeros((F, 2), dtype=np.float32)
    for k in range(F):
        nz = np.nonzero(basis[:, k])[0]
        if len(nz) > 2:
            return None
        for j, m in enumerate(nz):
            bands[k] |= int(m) << (16 * j)
            w[k, j] = basis[m, k]
    return bands.astype(np.int32), w


def mel_bin_table(sr, n_fft, n_mels, fmin, fmax, device):
    """Device copy of :func:`mel_bin_table_np` (cached), or None."""
    key = ("mel_bins", sr, n_fft, n_mels, fmin, fmax)
    if key in _mel_units_unsupported:
        return None
    t = mel_bin_table_np(mel_filters_np(sr, n_fft, n_mels, fmin, fmax))
    if t is None or n_mels >= 0xffff:
        _mel_units_unsupported.add(key)
        return None
    return device_table(key, device, lambda: t)


_mel_units_unsupported = set()


def mel_units_or_none(sr, n_fft, n_mels, fmin, fmax, device):
    """Unit tables of the fused mel stage, or None when the filterbank does not fit its layout
    (a band wider than 16 rows of 16 bins, e.g. 5 mels at n_fft 2048, or more than 384 units):
    the caller then applies the dense basis to |X| of the native STFT instead."""
    key = (sr, n_fft, n_mels, fmin, fmax)
    # n_fft = 32: 64 frames per wave, their padded magnitude rows do not fit the wave's LDS slab
    # (at_stft_mel_f32 answers AT_ERR_UNSUPPORTED for the mel stage)
    if key in _mel_units_unsupported or n_fft < 64:
        return None
    try:
        return mel_units(sr, n_fft, n_mels, fmin, fmax, device)
    except _native.NativeError as e:
        if "unsupported" not in str(e):
            raise
        _mel_units_unsupported.add(key)
        return None


def install_table(key, device, tensors):
    """Install an already-on-device table (used by the multi-GPU broadcast)."""
    _device_cache[(key, _dev_key(device))] = tensors


def drop_table(key, device):
    """Forget a cached / installed table (its memory returns to torch's allocator once nothing else holds it)."""
    _device_cache.pop((key, _dev_key(device)), None)


# ------------------------------------------------------- sinc FIR design
def _sinc_t(x: torch.Tensor) -> torch.Tensor:
    one = torch.ones((), dtype=x.dtype)
    return torch.where(x == 0, one, torch.sin(x) / x)


def lowpass_half_size(cutoff, zeros: float) -> int:
    """Half length of the windowed-sinc low-pass the reference builds through
    julius (``dsp.py:178``): ``int(zeros / cutoff / 2)`` evaluated in the
    dtype of ``cutoff`` (a float32 0-dim tensor when it comes from
    ``low_pass``), so boundary cases round identically."""
    return int(zeros / cutoff / 2)


def lowpass_taps(cutoff, zeros: float, half_size: int = None) -> torch.Tensor:
    """Float32 taps (2*half+1,) of a Hann-windowed sinc low-pass with
    normalised cutoff ``cutoff`` (cycles/sample), unit DC gain.  ``cutoff == 0``
    gives the all-zero filter.  Arithmetic is float32 torch, as upstream."""
    if cutoff < 0:
        raise ValueError("Minimum cutoff must be larger than zero.")
    if cutoff > 0.5:
        raise ValueError("A cutoff above 0.5 does not make sense.")
    if half_size is None:
        half_size = lowpass_half_size(cutoff, zeros)
    win = torch.hann_window(2 * half_size + 1, periodic=False)
    n = torch.arange(-half_size, half_size + 1)
    if cutoff == 0:
        return torch.zeros(2 * half_size + 1)
    h = 2 * cutoff * win * _sinc_t(2 * cutoff * math.pi * n)
    return (h / h.sum()).float()


def htk_band_edges(sample_rate: float, n_bands: int) -> np.ndarray:
    """Interior band edges (Hz) of the ``n_bands``-way mel split used by
    ``equalizer`` / ``mel_filterbank`` (effects.py:400): HTK-mel-spaced
    points between 0 and Nyquist, end points dropped."""
    to_mel = lambda f: 2595 * np.log10(1 + f / 700)
    to_hz = lambda m: 700 * (10 ** (m / 2595) - 1)
    mels = np.linspace(to_mel(0.0), to_mel(sample_rate / 2), n_bands + 1)
    return to_hz(mels)[1:-1]


@functools.lru_cache(None)
def band_split_bank(sample_rate: int, n_bands: int, zeros: float = 8):
    """Low-pass bank (n_bands-1, 2*half+1) float32 + half size for the mel
    band split; all filters share the length set by the LOWEST edge."""
    edges = htk_band_edges(sample_rate, n_bands)
    if len(edges) == 0:
        return None, 0
    cut = [c / sample_rate for c in edges]
    half = int(zeros / min(c for c in cut if c > 0) / 2)
    return torch.stack([lowpass_taps(c, zeros, half) for c in cut]), half


@functools.lru_cache(32)      # pitch_shift feeds arbitrary p/q: bounded (MB-sized banks)
def resample_bank(old_sr: int, new_sr: int, zeros: int = 24, rolloff: float = 0.945):
    """Polyphase windowed-sinc bank of the reference's resampler
    (``audio_signal.py:732`` -> julius.resample_frac): returns
    ``(bank[new, 2*width+old] float32, old, new, width)`` for the REDUCED
    ratio, or ``None`` when the rates are equal."""
    g = math.gcd(old_sr, new_sr)
    old, new = old_sr // g, new_sr // g
    if old == new:
        return None
    sr = min(new, old) * rolloff
    width = math.ceil(zeros * old / sr)
    idx = torch.arange(-width, width + old).float()
    rows = []
    for i in range(new):
        t = (-i / new + idx / old) * sr
        t = t.clamp_(-zeros, zeros)
        t *= math.pi
        win = torch.cos(t / zeros / 2) ** 2
        k = _sinc_t(t) * win
        k.div_(k.sum())
        rows.append(k)
    return torch.stack(rows), old, new, width


@functools.lru_cache(32)      # pitch_shift feeds arbitrary p/q: bounded (MB-sized banks)
def resample_sparse_bank(old_sr: int, new_sr: int, zeros: int = 24, rolloff: float = 0.945):
    """Sparse, transposed form of :func:`resample_bank` for ``at_resample_f32``: per output
    phase only the contiguous run of taps with |t| < zeros is non-negligible (the cos^2 window
    is ~1e-17 at the clamp), so the (new, 2w+old) bank is stored as ``ws[Wd, new]`` with the
    run of phase i starting at dense index ``k0[i]``.  Returns
    ``(ws float32, k0 int32, old, new, width, Wd)`` or None for equal rates."""
    plan = resample_bank(old_sr, new_sr, zeros, rolloff)
    if plan is None:
        return None
    bank, old, new, width = plan
    b = bank.numpy()
    thr = 1e-12 * np.abs(b).max()
    k0 = np.zeros(new, dtype=np.int32)
    k1 = np.zeros(new, dtype=np.int32)
    for i in range(new):
        nz = np.nonzero(np.abs(b[i]) > thr)[0]
        k0[i], k1[i] = nz[0], nz[-1] + 1
    Wd = int((k1 - k0).max())
    k0 = np.minimum(k0, b.shape[1] - Wd).astype(np.int32)   # keep k0 + Wd inside the dense row
    ws = np.zeros((Wd, new), dtype=np.float32)
    for i in range(new):
        ws[:, i] = b[i, k0[i]: k0[i] + Wd]
    return ws, k0, old, new, width, Wd


@functools.lru_cache(32)      # pitch_shift feeds arbitrary p/q: bounded (MB-sized banks)
def resample_grouped_bank(old_sr: int, new_sr: int, zeros: int = 24, rolloff: float = 0.945):
    """Bank layout of ``at_resample_f32``: output phases grouped by 4; group G stores, for every
    tap of the union of its phases' support windows, one float4 (4 phases), zero filled.
    Tap-major: adjacent groups are adjacent in memory, so the lanes of a wave (adjacent groups)
    read one contiguous run per tap.
    Returns ``(wg float32 (LG, NG, 4), base int32 (NG,), old, new, width, NG, LG)`` or None."""
    sp = resample_sparse_bank(old_sr, new_sr, zeros, rolloff)
    if sp is None:
        return None
    ws, k0, old, new, width, Wd = sp
    dense = resample_bank(old_sr, new_sr, zeros, rolloff)[0].numpy()
    K = dense.shape[1]
    NG = (new + 3) // 4
    base = np.zeros(NG, dtype=np.int32)
    spans = []
    for G in range(NG):
        ph = list(range(4 * G, min(4 * G + 4, new)))
        lo = int(min(k0[i] for i in ph))
        hi = int(max(k0[i] + Wd for i in ph))
        base[G] = lo
        spans.append(hi - lo)
    LG = (int(max(spans)) + 3) // 4 * 4   # the kernel consumes taps in blocks of 4 (zero padded)
    base = np.minimum(base, K - LG).astype(np.int32)
    wg = np.zeros((NG, LG, 4), dtype=np.float32)
    thr = 1e-12 * np.abs(dense).max()
    for G in range(NG):
        for p in range(4):
            i = 4 * G + p
            if i < new:
                seg = dense[i, base[G]: base[G] + LG].copy()
                seg[np.abs(seg) <= thr] = 0.0
                wg[G, :, p] = seg
    return np.ascontiguousarray(wg.transpose(1, 0, 2)), base, old, new, width, NG, LG


def group_dense_bank(dense: np.ndarray):
    """Any banded polyphase bank (phases, taps) in the layout of ``at_resample_f32``: phases grouped by 4, group G keeps
    the taps base[G] .. base[G] + LG - 1 of the union support of its phases as float4 rows, tap-major.
    Returns ``(wg float32 (LG, NG, 4), base int32 (NG,), NG, LG)``."""
    P, K = dense.shape
    thr = 1e-12 * np.abs(dense).max()
    nzm = np.abs(dense) > thr
    NG = (P + 3) // 4
    lo = np.zeros(NG, dtype=np.int64)
    hi = np.zeros(NG, dtype=np.int64)
    for G in range(NG):
        cols = np.nonzero(nzm[4 * G: 4 * G + 4].any(0))[0]
        lo[G], hi[G] = (int(cols[0]), int(cols[-1]) + 1) if len(cols) else (0, 1)
    LG = (int((hi - lo).max()) + 3) // 4 * 4
    if LG > K:                                   # tiny banks: zero columns up to one block of 4
        dense = np.concatenate([dense, np.zeros((P, LG - K), dtype=dense.dtype)], 1)
        K = LG
    base = np.minimum(lo, K - LG).astype(np.int32)
    wg = np.zeros((NG, LG, 4), dtype=np.float32)
    for G in range(NG):
        for p in range(4):
            i = 4 * G + p
            if i < P:
                seg = dense[i, base[G]: base[G] + LG].copy()
                seg[np.abs(seg) <= thr] = 0.0
                wg[G, :, p] = seg
    return np.ascontiguousarray(wg.transpose(1, 0, 2)), base, NG, LG


@functools.lru_cache(32)
def resample_adjoint_bank(old_sr: int, new_sr: int, zeros: int = 24, rolloff: float = 0.945):
    """The TRANSPOSE of the resampler as a polyphase bank for the same kernel (``at_resample_f32`` with the roles of the
    two rates swapped).  Forward (audio_signal.py:732 -> julius.resample_frac), with xp the replicate-padded input and
    h = resample_bank (new, K):   y[f new + p] = sum_k h[p][k] xp[f old + k].
    Its adjoint with respect to xp, position i = g old + q:
        dxp[g old + q] = sum_j sum_p h[p][j old + q] dy[(g - j) new + p]  =  sum_m HT[q][m] dyz[g new + m],
        m = (J - j) new + p,   HT[q][m] = h[m % new][(J - m // new) old + q]  (0 where that tap does not exist),
        dyz[i] = dy[i - J new] (zero outside),  J >= (K - 1) // old.
    -- ``old`` output phases per frame, the input advancing by ``new`` per frame, (J + 1) new taps: the kernel's own form.
    Returns ``(wg, base, old, new, width, NG, LG, J)`` (reduced rates; ``width`` is the FORWARD padding) or None for
    equal rates."""
    plan = resample_bank(old_sr, new_sr, zeros, rolloff)
    if plan is None:
        return None
    bank, old, new, width = plan
    h = bank.numpy()
    K = h.shape[1]
    J = (K - 1) // old
    if J * new < 2:                 # the kernel needs a left padding >= 1 on top of the one zero sample in front of dy
        J += 1
    HT = np.zeros((old, (J + 1) * new), dtype=np.float32)
    for j in range(J + 1):
        ks = j * old + np.arange(old)
        ok = ks < K
        HT[ok, (J - j) * new: (J - j + 1) * new] = h[:, ks[ok]].T
    wg, base, NG, LG = group_dense_bank(HT)
    return wg, base, old, new, width, NG, LG, J


# K-slot -> tap offset inside a 32-tap chunk of the MFMA resampler: MFMA number s (0..7) of a chunk
# consumes taps s + MFMA_KOFF[k], k = lane // 16.  Offsets 0/16 (lanes 0-31) and 8/24 (lanes 32-63)
# make the 32 lanes of a ds_read_b32 group hit 32 distinct banks for every odd ``old``:
# lane (i, k) reads xs[i*old + off_k + const]; i*old mod 32 takes 16 distinct values and adding 16
# gives the other 16.
MFMA_KOFF = (0, 16, 8, 24)


@functools.lru_cache(32)      # pitch_shift feeds arbitrary p/q: bounded (MB-sized banks)
def resample_mfma_bank(old_sr: int, new_sr: int, zeros: int = 24, rolloff: float = 0.945):
    """Bank layout of ``at_resample_mfma_f32`` (v_mfma_f32_16x16x4_f32 form of the sparse polyphase
    resampler).  Output phases are cut into blocks of 16 (one MFMA column block); block P keeps the
    taps of the union support window of its phases, ``lo[P] .. lo[P] + LW`` (LW a common multiple
    of 32, zero filled).  The weights are stored in the order the kernel consumes them:
        W[P, c, h, lane, e]  =  bank[16 P + lane % 16,  lo[P] + 32 c + (4 h + e) + MFMA_KOFF[lane // 16]]
    i.e. one float4 per lane covers the B operands of 4 consecutive MFMAs, and a wave's load is 1 KB
    contiguous.  Returns ``(W float32 (NPB, NC, 2, 64, 4), lo int32 (NPB,), old, new, width, NPB, NC)``
    or None for equal rates."""
    plan = resample_bank(old_sr, new_sr, zeros, rolloff)
    if plan is None:
        return None
    bank, old, new, width = plan
    b = bank.numpy()
    K = b.shape[1]
    thr = 1e-12 * np.abs(b).max()
    NPB = (new + 15) // 16
    lo = np.zeros(NPB, dtype=np.int32)
    span = 0
    for P in range(NPB):
        rows = b[16 * P: min(16 * P + 16, new)]
        nz = np.nonzero((np.abs(rows) > thr).any(0))[0]
        lo[P] = nz[0]
        span = max(span, int(nz[-1]) + 1 - int(nz[0]))
    NC = (span + 31) // 32
    LW = 32 * NC
    W = np.zeros((NPB, NC, 2, 64, 4), dtype=np.float32)
    lane = np.arange(64)
    j, k = lane % 16, lane // 16
    koff = np.asarray(MFMA_KOFF)[k]
    for P in range(NPB):
        ph = 16 * P + j
        for c in range(NC):
            for h in range(2):
                for e in range(4):
                    tap = lo[P] + 32 * c + 4 * h + e + koff
                    ok = (ph < new) & (tap < K)
                    vals = np.where(ok, b[np.minimum(ph, new - 1), np.minimum(tap, K - 1)], 0.0)
                    vals = np.where(np.abs(vals) > thr, vals, 0.0)
                    W[P, c, h, :, e] = vals
    return W, lo, old, new, width, NPB, NC


@functools.lru_cache(32)
def resample_f16_bank(old_sr: int, new_sr: int, zeros: int = 24, rolloff: float = 0.945):
    """Bank layout of ``at_resample_f16s_f32`` (csrc/resample_f16.hip: the polyphase sum on
    v_mfma_f32_16x16x32_f16 with every tap split into an fp16 high and low half).  Phase blocks and windows
    as :func:`resample_mfma_bank`; the taps are scaled by ``2 ** wscale_log2`` (max |w| lands in [2^13, 2^14):
    the low halves of the significant taps stay normal fp16 numbers) and stored as the B operands the kernel
    consumes, eight fp16 per lane and chunk:
        half s of W[P, c, h, lane, :]  =  (hi if h == 0 else lo)(2^k bank[16 P + lane % 16, lo[P] + 32 c + s + MFMA_KOFF[lane // 16]])
    with hi = RN16(v), lo = RN16(v - hi) and half s in bits 16 (s % 2) .. of dword s // 2.
    Returns ``(W uint32 (NPB, NC, 2, 64, 4), lo int32 (NPB,), old, new, width, NPB, NC, wscale_log2)`` or None."""
    plan = resample_mfma_bank(old_sr, new_sr, zeros, rolloff)
    if plan is None:
        return None
    W32, lo, old, new, width, NPB, NC = plan
    # resample_mfma_bank holds slot s of a chunk at [h = s // 4, e = s % 4]: (NPB, NC, 2, 64, 4) -> (NPB, NC, 64, 8)
    taps = np.ascontiguousarray(np.transpose(W32, (0, 1, 3, 2, 4))).reshape(NPB, NC, 64, 8)
    wmax = float(np.abs(taps).max())
    k = 13 - int(math.floor(math.log2(wmax)))
    v = (taps * np.float32(2.0 ** k)).astype(np.float32)
    hi = v.astype(np.float16)
    lw = (v - hi.astype(np.float32)).astype(np.float16)
    out = np.zeros((NPB, NC, 2, 64, 4), dtype=np.uint32)
    for h, plane in enumerate((hi, lw)):
        u = plane.view(np.uint16).astype(np.uint32)          # (NPB, NC, 64, 8)
        out[:, :, h] = u[..., 0::2] | (u[..., 1::2] << 16)
    return out, lo, old, new, width, NPB, NC, k
