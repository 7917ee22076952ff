import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A checkout without build artefacts (the .so is git-ignored): build the library the way __graft_entry__.build() does
    # before anything imports galois_amd.  This builds the product; it is not a fallback path -- without hipcc the import
    # below fails loudly.
    lib = os.path.join(ROOT, "galois_amd", "libgalois_amd.so")
    if not os.path.exists(lib):
        import subprocess

        subprocess.run([sys.executable, os.path.join(ROOT, "galois_amd", "build.py")], check=True)


def pytest_collection_modifyitems(config, items):
    # GPU tests fail loudly (not skip) when selected on a machine without a GPU: a silent skip would hide a broken box.
    pass


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
