import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graphneuralnetworks.jl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (C restatement + numpy composition).  Checker only."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def oracle_np():
    from oracle import oracle_np
    return oracle_np
