"""Import-path compatibility with the reference (``audiotools.core``)."""
from .. import util  # noqa: F401
from ..meter import Meter  # noqa: F401
from ..signal import AudioSignal, STFTParams  # noqa: F401
from . import audio_signal, dsp, effects, loudness  # noqa: F401,E402
