import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def bunny_path():
    import objfmt
    return objfmt.bunny_obj_path()


@pytest.fixture(scope="session")
def bunny_mesh(oracle, bunny_path):
    """(xyzw, idx) of the bunny as the oracle's A1 restatement loads it."""
    return oracle.load_obj(bunny_path)


@pytest.fixture(scope="session")
def gpu_ctx():
    import renderer_amd as R
    ctx = R.Context(0)
    yield ctx
    ctx.close()
