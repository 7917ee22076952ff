import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests fail loudly (not skip) when selected on a machine without a GPU: a silent skip would hide a broken box.
    pass


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
