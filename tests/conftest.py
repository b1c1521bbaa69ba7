import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

graft.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sample_data():
    from sm64rt_legacy_renderer_amd import sample_scene
    return sample_scene.make_sample_scene()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py.lib()


@pytest.fixture(scope="session")
def rt64_lib():
    """The HIP library.  GPU tests fail loudly when it is missing -- there is no CPU fallback to pass on."""
    from sm64rt_legacy_renderer_amd import rt64
    return rt64.Library()
