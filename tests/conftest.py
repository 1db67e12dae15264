import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """Import the product package (its directory name has a hyphen)."""
    import importlib
    if "odr_dabmod_amd" in sys.modules:
        return sys.modules["odr_dabmod_amd"]
    mod = importlib.import_module("odr-dabmod_amd")
    sys.modules["odr_dabmod_amd"] = mod
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()
