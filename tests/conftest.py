import os
import sys

import pytest

try:  # the bundled HIP runtime of torch must be the first one loaded (see strumpack_amd/__init__.py)
    import torch  # noqa: F401
except Exception:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "hss_seq_golden.json")) as f:
        return json.load(f)
