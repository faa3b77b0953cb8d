"""Import-path compatibility: ``audiotools.core.audio_signal`` (reference core/audio_signal.py)."""
from ..signal import AudioSignal, STFTParams  # noqa: F401
