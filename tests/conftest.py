import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm device (run on the MI355X box with -m gpu)")


@pytest.fixture(scope="session")
def scpose():
    import scpose as pkg  # alias module for the hyphenated package directory
    return pkg


@pytest.fixture(scope="session")
def gpu_ops():
    """The torch-facing wrappers of the C ABI.  GPU tests must exercise the HIP library:
    a missing build or a missing device is a FAILURE, never a skip."""
    import torch
    import scpose  # noqa: F401
    from importlib import import_module
    ops = import_module("spacecraft-pose-estimation_amd.ops")
    ops.nat.lib()  # raises NativeError when the extension is not built
    assert torch.cuda.is_available(), "GPU test selected but no ROCm device is visible"
    return ops
