"""Import-path compatibility: ``audiotools.core.dsp`` (reference core/dsp.py)."""
from ..filters import DSPMixin  # noqa: F401
