import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the product library arms its one test hook (mustafar_compress_test_skip_publish) only in a process that asks for it before first use
os.environ.setdefault("MUSTAFAR_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the native pieces once if a fresh checkout lacks them (hipcc cross-compiles without a GPU; on the GPU box
    the prebuilt .so files travel with the snapshot)."""
    import glob
    lib = os.path.join(ROOT, "mustafar_amd", "lib", "libmustafar_hip.so")
    ext = glob.glob(os.path.join(ROOT, "mustafar_amd", "dropin", "mustafar_package*.so"))
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    import __graft_entry__

    def stale():
        if not (os.path.exists(lib) and ext and os.path.exists(orc)):
            return True
        # a stale binary must not be tested against newer sources: build() leaves the hash of its sources next to the library
        try:
            return open(__graft_entry__.STAMP).read().strip() != __graft_entry__.native_source_hash()
        except OSError:
            return True

    if stale():
        __graft_entry__.build()
    yield
