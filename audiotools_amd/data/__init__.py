"""Import-path compatibility with the reference (``audiotools.data.transforms``,
``audiotools.data.datasets.AudioLoader``)."""
from .. import transforms  # noqa: F401
from . import datasets  # noqa: F401
from .datasets import AudioLoader  # noqa: F401
from .staging import DeviceStager  # noqa: F401
