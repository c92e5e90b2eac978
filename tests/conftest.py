import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A test that hangs (a kernel whose waves never finish, a rank waiting for a peer) must end the run, not sit there until
    something outside gives up: 20 minutes per test -- the slowest takes one -- enforced from a watchdog thread, because a
    signal cannot interrupt a native call that never returns.  Tests that set their own limit keep it."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(1200, method="thread"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
