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
