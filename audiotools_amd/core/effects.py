"""Import-path compatibility: ``audiotools.core.effects`` (reference core/effects.py)."""
from ..fx import EffectMixin, ImpulseResponseMixin  # noqa: F401
