import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # soak runs: AVRF_TEST_SEED_SHIFT=k adds k to every integer seed the tests give random.Random, so the whole suite -- every
    # GPU-against-oracle comparison on seeded inputs -- runs on other inputs (the golden fixtures are data and do not move)
    shift = int(os.environ.get("AVRF_TEST_SEED_SHIFT", "0") or 0)
    if shift:
        import random
        base = random.Random

        class Shifted(base):
            def __init__(self, x=None):
                super().__init__(x + shift if isinstance(x, int) else x)
        random.Random = Shifted


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """libavrf.so is a build product (git-ignored): build it once if this checkout does not have it yet
    (hipcc cross-compiles without a GPU; the GPU box receives the built file with the snapshot)."""
    lib = os.path.join(ROOT, "ark_vrf_amd", "libavrf.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    yield
