import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "ray-tracer-challenge-rust_amd")
for p in (REPO, PKG, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Every new view gets its tile table in the tests (a scene that sees three new views in a row stops building them until a view
    # repeats: csrc/rtc_hip.hip strip_table); the test of that rule takes the variable away again.
    os.environ.setdefault("RTC_STRIP_ALWAYS", "1")


def pytest_collection_modifyitems(config, items):
    """Parity first: tests of the bench / tooling contracts must never stand between `-x` and the oracle-parity
    suite (round 1 lost its whole GPU record to a contract test that sorted ahead of test_gpu_parity.py)."""
    def rank(item):
        name = item.fspath.basename
        if name == "test_gpu_parity.py":
            return 0
        if name.startswith("test_zz_"):
            return 2
        return 1
    items.sort(key=rank)  # stable: file order is kept inside each class


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """The native libraries are built in-tree (and shipped prebuilt to the GPU box); build if absent."""
    import __graft_entry__ as ge

    ge.ensure_built()


@pytest.fixture(scope="session")
def prod():
    import rtc_amd

    return rtc_amd.api


@pytest.fixture(scope="session")
def orc():
    import oracle_binding

    return oracle_binding.api


@pytest.fixture(scope="session")
def gpu_device():
    import rtc_amd

    n = rtc_amd.device_count()
    if n < 1:
        pytest.fail("marked gpu but no HIP device is visible: the product has no CPU fallback")
    return 0
