import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The HIP library is git-ignored (it travels with the working tree, not with history): a fresh checkout has to
    # compile it once.  hipcc cross-compiles gfx950 without a GPU; a no-op when the sources are unchanged.
    from brepgen_amd.build import build
    build(verbose=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
