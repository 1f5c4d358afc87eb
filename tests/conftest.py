import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def capi():
    """The C-ABI library; built here if the in-tree .so is missing (hipcc cross-compiles on CPU)."""
    from mp3rgain_amd import _capi

    if not _capi.LIB_PATH.exists():
        import __graft_entry__

        __graft_entry__.build()
    return _capi.load()


@pytest.fixture(scope="session")
def analyzer(capi):
    """GPU context; only gpu-marked tests may request it.  No fallback: failing to get a
    device is a hard error on the GPU box."""
    import mp3rgain_amd as rg

    an = rg.Analyzer(0)
    yield an
    an.close()
