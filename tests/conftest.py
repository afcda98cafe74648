import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-semantic-segmentation_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture
def heavy_threshold():
    """Setter for VP_OPT_HEAVY_THRESHOLD on every workspace of both fronts (None = the library's default, min(256 + 64*B*V, 2048));
    the default is restored afterwards.  Replaces the environment variable of ABI v2: options belong to the workspace."""
    import voxproj_host

    def set_(value):
        voxproj_host.set_default_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, value)
        try:
            import torch  # noqa: F401
            import project_features_cuda as m
            m.set_workspace_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, -1 if value is None else int(value))
        except ImportError:
            pass
    yield set_
    set_(None)


@pytest.fixture
def workspace_option():
    """Setter for any VP_OPT_* on every workspace of both fronts (None = the library's default); restored afterwards."""
    import voxproj_host
    touched = set()

    def set_(option, value):
        touched.add(int(option))
        voxproj_host.set_default_option(int(option), value)
        try:
            import torch  # noqa: F401
            import project_features_cuda as m
            m.set_workspace_option(int(option), -1 if value is None else int(value))
        except ImportError:
            pass
    yield set_
    for opt in touched:
        set_(opt, None)
