"""Test configuration: registers the `gpu` marker and exposes the package (its directory name contains
hyphens, so it is imported through importlib)."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "semi-supervised-segmentation-cyclegan_amd"


def load_pkg():
    return importlib.import_module(PKG_NAME)


def load_sub(name):
    return importlib.import_module(PKG_NAME + "." + name)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The fp64 / fp32 references of these tests are torch CPU ops on small maps: on a 128-core (256-thread) GPU host they lose time
    # beyond ~32 threads (bench.py's cpu_baseline: 52 s per oracle step on 128 threads against 10.5 s on 64).  Tests that pin a thread
    # count for bit-exactness (test_oracle_golden.py) set their own.
    try:
        import torch
        torch.set_num_threads(min(torch.get_num_threads(), 32))
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs (a kernel that never returns) must not take the box with it: with pytest-timeout installed every
    gpu test gets a 20-minute ceiling, enforced from a watchdog thread (a blocked HIP call never returns to a signal handler)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(1200, method="thread"))


@pytest.fixture(scope="session")
def F():
    return load_sub("functional")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _drain_the_device_between_gpu_tests(request):
    """A test may return while its model's side lanes (operand-copy refresh, an overlapped discriminator step) are still running; the
    next test's allocations must not meet kernels of a model that no longer exists.  Launch what is queued, drain, collect."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import gc
    import torch
    if torch.cuda.is_available():
        F = load_sub("functional")
        F.flush_side_work()
        torch.cuda.synchronize()
        # models hold reference cycles (autograd nodes <-> closures): collect them where models are built - the kernel-level files
        # (500 tests) build none, and a full collection costs ~0.25 s per test (130 s of the round-3 suite)
        if os.path.basename(str(request.node.fspath)) not in ("test_kernels_gpu.py", "test_abi.py", "test_schedule_gpu.py"):
            gc.collect()
            torch.cuda.synchronize()
