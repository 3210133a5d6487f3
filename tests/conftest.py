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
    config.addinivalue_line("markers", "slow: the long form of the suite (minutes of CPU, or GPU cases beyond the default run's ~6 minutes); "
                                       "skipped unless LAM_RUN_SLOW=1")


def pytest_collection_modifyitems(config, items):
    """The default `-m gpu` run stays near six minutes (round 4: 363 s on the driver's box; a round-5 suite of 599 s was one second
    from the usual ten-minute limit): parametrisations marked `slow` repeat a shape the default run already covers at another size
    or rank count and run with LAM_RUN_SLOW=1 only."""
    if os.environ.get("LAM_RUN_SLOW", "0") not in ("", "0"):
        return
    skip = pytest.mark.skip(reason="long form of the suite: set LAM_RUN_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


def slow(*values):
    """A parametrisation that only the long form of the suite runs."""
    return pytest.param(*values, marks=pytest.mark.slow)


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


MOCK_DIR = os.path.join(ROOT, "tests", "mock_rccl")


@pytest.fixture(scope="session")
def mock_async():
    """Stream-ordered RCCL test double (tests/mock_rccl/mock_rccl_async.hip), built on demand."""
    import subprocess
    lib, src = os.path.join(MOCK_DIR, "libmock_rccl_async.so"), os.path.join(MOCK_DIR, "mock_rccl_async.hip")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", src,
                        "-o", lib, "-lrt"], check=True)
    return lib


@pytest.fixture(scope="session")
def mock_mp_lib():
    """Host-synchronous multi-process RCCL test double (tests/mock_rccl/mock_rccl_mp.cpp)."""
    import subprocess
    lib, src = os.path.join(MOCK_DIR, "libmock_rccl_mp.so"), os.path.join(MOCK_DIR, "mock_rccl_mp.cpp")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        # g++ and NOT linked against libamdhip64: the HIP symbols bind at first use to the runtime the
        # process already holds (liblam_hip.so's) instead of dragging in a second one
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        src, "-o", lib, "-lrt", "-lpthread"], check=True)
    return lib
