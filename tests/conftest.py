import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `-m gpu`)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    oracle_lib.lib()
    return oracle_lib
