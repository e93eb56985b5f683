import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # hiprtc instances are cached on disk (VX355_CACHE_DIR): a fresh directory per test session, so
    # that no test depends on what an earlier session left behind
    if "VX355_CACHE_DIR" not in os.environ:
        import tempfile
        os.environ["VX355_CACHE_DIR"] = tempfile.mkdtemp(prefix="vx355_test_cache_")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def vx():
    """The product library on a GPU box. Fails loudly when the HIP extension is
    missing or no device is visible."""
    # torch bundles its own HIP runtime; bench.py loads torch first, and the one
    # GPU test that needs torch.distributed must see the same arrangement, so
    # load torch before libvx355 pulls in /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    from velox_amd import ops
    ops.init(0)
    return ops
