"""Training losses on AudioSignal pairs (reference ``audiotools/metrics``): the spectral and
waveform distances whose hot path is ``stft()`` / ``mel_spectrogram()`` under autograd -- on HIP
tensors that is the native forward + adjoint kernel pair.  The perceptual
``quality`` metrics of the reference (PESQ / STOI / ViSQOL wrappers around CPU packages) are out
of scope."""
from . import distance, spectral  # noqa: F401
