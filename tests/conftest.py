import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def gold_dir():
    return GOLD


@pytest.fixture(scope='session')
def hip_lib():
    """Build (if needed) and load the HIP library; GPU tests call through it."""
    from lct_vqa_amd import build
    build.build(verbose=False)
    from lct_vqa_amd._lib import lib
    lib.load()
    return lib
