import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def pytest_collection_modifyitems(config, items):
    """GPU tests get a hard per-test limit (pytest-timeout, thread method: a wedged HIP call cannot be interrupted by a
    signal handler) so that a hung kernel ends the run as a failure instead of stalling it."""
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600, method="thread"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
