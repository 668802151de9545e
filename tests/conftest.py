import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as O
    O.build()
    return O


def pytest_collection_modifyitems(config, items):
    """GPU run order: the direct parity tests (vectors, differential ticks, WAL kernels) first, the long
    closed-loop replays last -- with `-x` a failure in the broadest test must not hide the focused ones."""
    late = [it for it in items if "test_cluster_safety" in it.nodeid and it.get_closest_marker("gpu")]
    if late:
        rest = [it for it in items if it not in late]
        items[:] = rest + late
