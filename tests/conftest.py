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


def record_bound(name, measured, limit):
    """Log a measured worst case next to the bound it is held to (gpurun_out/measured_bounds.jsonl, one JSON
    object per line; copied to profiles/ when a bound is (re)derived from it) and return measured <= limit."""
    import json
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "measured_bounds.jsonl"), "a") as f:
            f.write(json.dumps({"name": name, "measured": float(measured), "limit": float(limit)}) + "\n")
    except OSError:
        pass
    return float(measured) <= float(limit)
