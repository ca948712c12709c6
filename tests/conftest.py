import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the tests bind the in-tree library: (re)build it when it is missing or was built from other sources (a no-op
    # otherwise; hipcc cross-compiles gfx950 without a GPU), and the C oracle with it
    from ultrare_amd import build as lib_build
    lib_build.build()
    from oracle import build as oracle_build
    oracle_build.build()


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
