import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
# (no AT_* switches here: the suite runs the shipped library exactly as production does; the A/B switches exist only in
#  the development build, AT_DEV_KNOBS=1, which the tools under tools/ load)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle_c():
    """The oracle's C restatement is test infrastructure; build it once."""
    from oracle import cport

    cport.build()
    # the product library normally arrives prebuilt (driver build() / gpurun snapshot); a fresh
    # checkout compiles it here once (hipcc cross-compiles gfx950 without a GPU)
    from audiotools_amd import _native

    _native.build()
    yield


@pytest.fixture(scope="session")
def reference():
    """The unmodified reference package (only in the build container)."""
    from oracle import ref_import

    if not ref_import.reference_available():
        pytest.skip("/root/reference not present (GPU box)")
    return ref_import.import_reference()
