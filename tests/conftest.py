import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_sessionstart(session):
    """The shared objects are build artefacts (git-ignored): if a checkout has none yet, build them once -- hipcc
    cross-compiles gfx950 without a GPU.  Nothing is rebuilt when they exist (the GPU box receives them prebuilt)."""
    so = os.path.join(ROOT, 'biseqt_amd', 'pwlib', 'pwlib.so')
    orc = os.path.join(ROOT, 'oracle', 'libpw_oracle.so')
    if not (os.path.exists(so) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as O
    O.lib()
    return O
