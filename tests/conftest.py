import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_kzg():
    return load_golden("kzg.json")


@pytest.fixture(scope="session")
def golden_msm():
    return load_golden("msm.json")


@pytest.fixture(scope="session")
def golden_ntt():
    return load_golden("ntt.json")


@pytest.fixture(scope="session")
def golden_constants():
    return load_golden("constants.json")


@pytest.fixture(scope="session")
def fr_kat():
    return load_golden("fr_kat.json")


@pytest.fixture(scope="session")
def oracle_cpu():
    from oracle import cpu

    cpu.build()
    return cpu


@pytest.fixture(scope="module")
def hip():
    """Factory of HipEngine contexts on GPU 0 (closed when the module is done).  Raises if the HIP library is missing or no
    gfx950 device works: the GPU tests never fall back to anything."""
    from zkp_subnet_amd import HipEngine

    engines = []

    def make(window=0):
        e = HipEngine(0, window=window)
        engines.append(e)
        return e

    yield make
    for e in engines:
        e.close()
