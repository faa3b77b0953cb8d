"""Only the piece of the reference's ``audiotools.ml`` that sits on the STFT hot path:
``ml.layers.SpectralGate``.  Models, trainers and experiment tooling are
out of scope."""
from . import layers  # noqa: F401
