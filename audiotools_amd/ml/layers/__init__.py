from .spectral_gate import SpectralGate  # noqa: F401
