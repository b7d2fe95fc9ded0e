import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product: ctypes view of ro-map_amd/libmon_core.so (built in-tree by __graft_entry__.build())."""
    p = ge.load_package()
    if not os.path.exists(p.lib_path()):
        ge.build()
    return p


@pytest.fixture(scope="session")
def orc():
    """The checker: CPU restatement under oracle/ (test infrastructure)."""
    o = ge.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def ss():
    return ge.load_tools()


@pytest.fixture(scope="session")
def small_scene(ss):
    return ss.make_scene(n_views=12, H=120, W=160, f=130.0, seed=0)


C1 = dict(rays_per_batch=1024, n_levels=4, n_neurons=32, n_hidden_layers=2)          # BASELINE configs[0]
C2 = dict()                                                                              # base.json defaults, BASELINE configs[1]
