"""The C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/audiotools_amd.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from audiotools_amd import _native

    _native.build()
    return _native.lib()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "audiotools_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(at_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    syms = _declared_symbols()
    assert len(syms) >= 5
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/audiotools_amd.h but not exported"


def test_python_signatures_cover_header(lib):
    from audiotools_amd import _native

    assert sorted(_native.SIGNATURES) == _declared_symbols()


def test_host_only_entry_points(lib):
    out = np.empty(2 * 16, dtype=np.float32)
    assert lib.at_stft_twiddles_host(16, out.ctypes.data) == 0
    k = np.arange(16)
    assert np.allclose(out[0::2], np.cos(2 * np.pi * k / 16), atol=1e-7)
    assert np.allclose(out[1::2], -np.sin(2 * np.pi * k / 16), atol=1e-7)
    assert lib.at_stft_native_supported(2048) == 1
    assert lib.at_stft_fused_supported(2048) == 1 and lib.at_stft_fused_supported(4096) == 0
    for n in (4096, 8192, 16384, 400, 1200, 1920, 150, 6, 14, 882):   # generic mixed-radix sizes (n/2 = 2^a 3^b 5^c 7^d)
        assert lib.at_stft_native_supported(n) == 1, n
    for n in (401, 22, 2 * 143, 32768):                          # odd, or a prime factor > 7, or too long
        assert lib.at_stft_native_supported(n) == 0, n
    assert lib.at_lufs_workspace_bytes(512, 2, 441000, 17640, 4410) >= 512 * 2 * 100 * 8


def test_longconv_plan_and_tables(lib):
    """Host side of the four-step convolution (csrc/longconv.hip): the split of T/2 and the twiddle
    tables, evaluated in double."""
    import ctypes
    n1, n2 = ctypes.c_int(), ctypes.c_int()
    for T, want in [(240000, (60, 2000)), (16, (1, 8)), (44100 * 5, None), (48000 * 40, (480, 2000)), (44100 * 30, None)]:
        assert lib.at_longconv_supported(T) == 1
        assert lib.at_longconv_plan(T, ctypes.byref(n1), ctypes.byref(n2)) == 0
        assert n1.value * n2.value * 2 == T and n1.value <= 512 and n2.value <= 2048
        assert want is None or (n1.value, n2.value) == want
    for T in (10007, 15, 2 * 11 * 64, 2 * 513 * 2048 * 2, 0):            # odd, prime factor 11, too long
        assert lib.at_longconv_supported(T) == 0
        assert lib.at_longconv_plan(T, None, None) == -2
    T, N1, N2 = 9600, 3, 1600
    n = lib.at_longconv_table_floats(T)
    rt = 64 + (N2 + 63) // 64
    assert n == 2 * (N1 + N2 + N1 * rt + N1 + N2)
    tb = np.empty(n, dtype=np.float32)
    assert lib.at_longconv_tables_host(T, tb.ctypes.data, n) == 0
    assert lib.at_longconv_tables_host(T, tb.ctypes.data, n - 2) == -1
    z = tb[0::2] + 1j * tb[1::2]
    w = lambda num, den: np.exp(-2j * np.pi * (np.asarray(num) % den) / den)
    assert np.allclose(z[:N1], w(np.arange(N1), N1), atol=1e-7)
    assert np.allclose(z[N1:N1 + N2], w(np.arange(N2), N2), atol=1e-7)
    row = z[N1 + N2:N1 + N2 + N1 * rt].reshape(N1, rt)
    k1 = np.arange(N1)[:, None]
    assert np.allclose(row[:, :64], w(k1 * np.arange(64)[None], T // 2), atol=1e-7)
    assert np.allclose(row[:, 64:], w(k1 * 64 * np.arange(rt - 64)[None], T // 2), atol=1e-7)
    tail = z[N1 + N2 + N1 * rt:]
    assert np.allclose(tail[:N1], w(np.arange(N1), T), atol=1e-7)
    assert np.allclose(tail[N1:], w(N1 * np.arange(N2), T), atol=1e-7)
    assert lib.at_longconv_circ_f32(None, None, None, 1, 1, 1, T, None, None, None, 0, None) == -1


def test_argument_validation_without_gpu(lib):
    # NULL pointers / bad sizes are rejected before any HIP call
    assert lib.at_stft_mel_f32(None, 1, 100, None, None, 512, 128, 0, 0, 0, 0, 1, None, None, None, 0, 0,
                               None, None) == -1
    assert lib.at_lufs_f32(None, 1, 1, 100, None, None, 2, 400, 100, 1.0, 0.0, 0, None, None, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from audiotools_amd import _native

    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeError):
        _native.lib()
