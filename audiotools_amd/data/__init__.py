"""Import-path compatibility with the reference (``audiotools.data.transforms``)."""
from .. import transforms  # noqa: F401
from .staging import DeviceStager  # noqa: F401
