import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: >= 20 s on the GPU box and redundant with a faster test of the same path; runs only "
                                       "with BG_RUN_SLOW=1 (the driver's -m gpu run has a 1200 s cap)")
    # The HIP library is git-ignored (it travels with the working tree, not with history): a fresh checkout has to
    # compile it once.  hipcc cross-compiles gfx950 without a GPU; a no-op when the sources are unchanged.
    from brepgen_amd.build import build
    build(verbose=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    if os.environ.get("BG_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow duplicate coverage: set BG_RUN_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)
