"""Import-path compatibility: ``audiotools.core.loudness`` (reference core/loudness.py)."""
from ..meter import LoudnessMixin, Meter  # noqa: F401
