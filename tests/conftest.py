import sys
from pathlib import Path

import os

import pytest

# the tests drive two seams of the shipped ABI that production refuses: a stand-in for librccl.so (rg_comm_library) and
# foreign node engines (rg_node_create_backend); subprocesses inherit the switch
os.environ.setdefault("MP3RGAIN_AMD_TEST_SEAMS", "1")

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
def _ctx(capi):
    """GPU context; only gpu-marked tests may request it.  No fallback: failing to get a
    device is a hard error on the GPU box."""
    import torch  # noqa: F401  (torch first: one HIP runtime per process, see mp3rgain_amd/_capi.py)

    import mp3rgain_amd as rg

    an = rg.Analyzer(0)
    yield an
    an.close()


# every GPU parity test runs against both kernel variants and several segment lengths of variant 2:
#   (1, 0)    order-faithful halo kernel
#   (2, 0)    transient-moment kernels, segment length chosen by the library
#   (2, -1)   transient-moment kernels, smallest admissible segment (many segments per window)
#   (2, -2)   transient-moment kernels, one segment per window
#   (2, -3)   transient-moment kernels, four windows per segment (moments over the first, plain energies for the rest)
#   (2, -4)   transient-moment kernels, sixteen windows per segment
#   (2, -6)   thirty-seven windows per segment (round 4: what a 1000-track batch runs at, one round of blocks; rows whose
#             track ends inside a lane's run re-read the track's first window)
@pytest.fixture(params=[(1, 0), (2, 0), (2, -1), (2, -2), (2, -3), (2, -4), (2, -6)],
                ids=["halo", "tm-auto", "tm-short", "tm-window", "tm-multi4", "tm-multi16", "tm-multi37"])
def analyzer(_ctx, request):
    variant, seg = request.param
    _ctx.set_kernel(variant)
    _ctx.set_tuning(1, 0)
    _ctx.set_tuning(2, 0)
    _ctx.set_tuning(4, 0)
    if seg == -1:
        _ctx.set_tuning(2, 1 << 40)  # unreachable lane target -> smallest admissible segment
    elif seg == -2:
        _ctx.set_tuning(2, 1)        # any lane count is enough -> largest segment (= the window) ...
        _ctx.set_tuning(4, 1)        # ... of one window
    elif seg == -3:
        _ctx.set_tuning(2, 1)
        _ctx.set_tuning(4, 4)
    elif seg == -4:
        _ctx.set_tuning(2, 1)
        _ctx.set_tuning(4, 16)
    elif seg == -6:
        _ctx.set_tuning(2, 1)
        _ctx.set_tuning(4, 37)
    yield _ctx
    _ctx.set_kernel(0)
    _ctx.set_tuning(1, 0)
    _ctx.set_tuning(2, 0)
    _ctx.set_tuning(4, 0)
