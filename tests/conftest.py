import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


if os.environ.get("ILQR_TEST_LIB"):  # A/B runs of the GPU tests against another build of the library (development aid)
    from ilqr_planner_amd import capi as _capi

    _capi.LIB_PATH = os.environ["ILQR_TEST_LIB"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold():
    from tests.helpers import golden

    return golden()
