import os
import sys
import pytest

# every render / physics call of the suite checks that the wall grid it walks was built from the walls as they are
os.environ.setdefault('MEGASTEP_CHECK_GRID', '1')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as o
    o.build()
    return o
