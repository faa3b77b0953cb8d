"""audiotools_amd: MI355X (gfx950) native implementation of the batched DSP
hot path of descriptinc/audiotools behind the same ``AudioSignal`` /
``STFTParams`` / ``Meter`` / ``transforms`` API.

    from audiotools_amd import AudioSignal, STFTParams
    sig = AudioSignal(torch.randn(8, 2, 441000, device="cuda"), 44100)
    mel = sig.mel_spectrogram(80)      # fused HIP STFT + mel kernel
    lufs = sig.loudness()              # HIP K-weighting + gated loudness
"""
__version__ = "0.1.0"

from . import util  # noqa: F401
from .meter import Meter  # noqa: F401
from .signal import AudioSignal, STFTParams  # noqa: F401
from . import transforms  # noqa: F401,E402
from . import core, data, metrics, ml  # noqa: F401,E402
from .data import datasets  # noqa: F401,E402  (audiotools.datasets, reference __init__.py:9)
