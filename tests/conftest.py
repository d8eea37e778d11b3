"""pytest configuration: the `gpu` marker and session fixtures that build the test-only libraries.

CPU tier  (-m "not gpu"): oracle vs golden vectors / reference KATs, host logic (LM driver on the CPU
test backend), C-ABI load + symbol export, multi-rank protocol over gloo.
GPU tier  (-m gpu): parity of the HIP path (through the C ABI) against the oracle.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _make(directory):
    subprocess.run(["make", "-s", "-C", directory], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    from tests import helpers

    _make(os.path.join(ROOT, "oracle"))
    return helpers.load_oracle()


@pytest.fixture(scope="session")
def hostmath():
    from tests import helpers

    _make(os.path.join(ROOT, "tests", "cpu_backend"))
    return helpers.load_hostmath()


@pytest.fixture(scope="session")
def lib():
    from calibration_amd import capi

    return capi.load_library()


@pytest.fixture(scope="session")
def gpu_lib(lib):
    if lib.cba_device_count() <= 0:
        pytest.fail("gpu test selected but no HIP device is visible (the engine has no CPU fallback)")
    return lib
