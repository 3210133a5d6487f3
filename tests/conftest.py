import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "2024-eumaster4hpc-student-challenge_amd"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def lam():
    """The product package (name is not a Python identifier, so import it by string)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        return json.load(f)
