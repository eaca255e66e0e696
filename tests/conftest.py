import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session', autouse=True)
def _debug_options_from_env():
    """FDSR_TEST_DEBUG_OPTION="name=value[,name=value]": a TEST-harness variable (the library itself never reads the
    environment) that child pytest sessions use to run a subset of tests under one launcher option (fdsr_debug_option)."""
    spec = os.environ.get('FDSR_TEST_DEBUG_OPTION')
    if spec:
        from fastdiffsr_amd import _lib
        for item in spec.split(','):
            k, v = item.split('=')
            _lib.debug_option(k.strip(), int(v))
    yield
