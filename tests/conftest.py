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


def _device_count():
    from cvx_amd import capi
    return capi.lib().cvx_device_count()


def pytest_collection_modifyitems(config, items):
    # gpu tests are selected with -m gpu; if they are collected without a device, skip (never fall back)
    skip = None
    for item in items:
        if "gpu" in item.keywords:
            if skip is None:
                skip = pytest.mark.skip(reason="no HIP device visible") if _device_count() == 0 else False
            if skip:
                item.add_marker(skip)


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle
