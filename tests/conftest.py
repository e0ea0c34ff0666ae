import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests never run on a box without a device: fail loudly there instead of silently skipping
    # is the job of the tests themselves (they call into the C ABI, which raises).  Nothing to do here.
    return


@pytest.fixture
def tuning_library():
    """Tests that force an alternative kernel (MV_FORCE_*, strip heights, ...) run against the -DMV_TUNING build of the
    same sources: the product library reads no environment variable at all."""
    from cpu_vision_amd import _lib
    with _lib.tuning_library() as lib:
        assert lib.mv_build_id().decode().endswith("+tuning")
        yield lib
