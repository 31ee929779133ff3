import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    # development only: run the suite against a library built by tools/dev_variant.py instead of kmers_amd/libkmx.so
    parser.addoption("--kmx-lib", action="store", default=None, help="path of a development build of libkmx.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    path = config.getoption("--kmx-lib")
    if path:
        from kmers_amd import _lib

        _lib.LIB_PATH = os.path.abspath(path)


@pytest.fixture(scope="session")
def kats():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle

    oracle.lib()
    return oracle
