import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference/ZPAQSharp"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: reads /root/reference as text (skipped where it is absent)")


def pytest_collection_modifyitems(config, items):
    if not os.path.isdir(REFERENCE):
        skip = pytest.mark.skip(reason="/root/reference not present on this machine")
        for it in items:
            if "reference" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def ctx():
    import zpaqsharp_amd as z
    c = z.Context(0)
    yield c
    c.close()
