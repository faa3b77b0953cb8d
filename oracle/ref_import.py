"""Import the UNMODIFIED reference package (``/root/reference/audiotools``) on
CPU torch, with its absent third-party leaves shimmed from ``oracle/leaves``.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Works only in the
build container -- the GPU box has no ``/root/reference`` -- so it is used by
``oracle/make_golden.py`` (fixture generation) and by CPU tests that are
skipped when the reference tree is absent.  Bytecode writing is disabled so
the read-only reference tree is never written to.
"""
import importlib
import importlib.resources
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "audiotools"))


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_shims():
    from .leaves import julius_leaf, misc_leaves, pyloudnorm_leaf

    if "julius" not in sys.modules:
        j = _module("julius", **{k: getattr(julius_leaf, k) for k in (
            "LowPassFilter", "LowPassFilters", "HighPassFilter", "HighPassFilters",
            "SplitBands", "ResampleFrac", "resample_frac", "lowpass_filter",
            "lowpass_filters", "split_bands")})
        j.fftconv = _module("julius.fftconv", fft_conv1d=julius_leaf.fft_conv1d)
        j.core = _module("julius.core", unfold=julius_leaf.unfold,
                         sinc=julius_leaf.sinc,
                         mel_frequencies=julius_leaf.mel_frequencies)
        j.lowpass = _module("julius.lowpass", LowPassFilter=julius_leaf.LowPassFilter,
                            LowPassFilters=julius_leaf.LowPassFilters)
        j.bands = _module("julius.bands", SplitBands=julius_leaf.SplitBands)
        j.resample = _module("julius.resample", ResampleFrac=julius_leaf.ResampleFrac,
                             resample_frac=julius_leaf.resample_frac)

    if "torchaudio" not in sys.modules:
        def _no_sox(*a, **k):
            raise RuntimeError("sox effects are out of scope (SURVEY.md 8(f))")

        def _info(path):
            raise RuntimeError("file I/O is out of scope for the oracle")

        ta = _module("torchaudio", __version__="2.0.0", info=_info)
        ta.functional = _module("torchaudio.functional", lfilter=misc_leaves.lfilter,
                                create_dct=misc_leaves.create_dct)
        ta.sox_effects = _module("torchaudio.sox_effects",
                                 apply_effects_tensor=_no_sox)
        ta.io = _module("torchaudio.io")

    if "pyloudnorm" not in sys.modules:
        _module("pyloudnorm", Meter=pyloudnorm_leaf.Meter,
                IIRfilter=pyloudnorm_leaf.IIRfilter)

    if "librosa" not in sys.modules:
        def _load(*a, **k):
            raise RuntimeError("file I/O is out of scope for the oracle")

        lb = _module("librosa", load=_load, amplitude_to_db=misc_leaves.librosa_amplitude_to_db)
        lb.filters = _module("librosa.filters", mel=misc_leaves.librosa_mel)

    if "flatten_dict" not in sys.modules:
        _module("flatten_dict", flatten=misc_leaves.flatten,
                unflatten=misc_leaves.unflatten)

    for name in ("soundfile", "ffmpy", "randomname", "markdown2", "argbind",
                 "pystoi", "torch_stoi", "pesq"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                _module(name)
    if "importlib_resources" not in sys.modules:
        sys.modules["importlib_resources"] = importlib.resources
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        class SummaryWriter:  # pragma: no cover - never used by the DSP path
            def __init__(self, *a, **k):
                pass
        _module("torch.utils.tensorboard", SummaryWriter=SummaryWriter)
    try:
        import IPython  # noqa: F401
    except Exception:
        ip = _module("IPython")
        ip.display = _module("IPython.display", HTML=object, Audio=object,
                             display=lambda *a, **k: None)


_ref = None


def import_reference():
    """Return the unmodified reference ``audiotools`` module (cached)."""
    global _ref
    if _ref is not None:
        return _ref
    if not reference_available():
        raise RuntimeError(f"{REFERENCE_ROOT} is not present (GPU box?)")
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    _install_shims()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import audiotools  # the reference package, unmodified

    assert audiotools.__file__.startswith(REFERENCE_ROOT), audiotools.__file__
    _ref = audiotools
    return audiotools
