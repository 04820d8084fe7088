import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu")
    config.addinivalue_line("markers", "slow: minutes of CPU oracle work; runs only with PMD_RUN_SLOW=1")


@pytest.fixture(scope="session", autouse=True)
def _poison_all_allocations():
    """PMD_TEST_POISON=1: the whole session runs with every torch.empty device buffer pre-filled with NaN patterns (a read of
    uninitialised memory then shows up as a NaN or a failed comparison in whatever test exercises it)."""
    if not os.environ.get("PMD_TEST_POISON"):
        yield
        return
    from tests.test_gpu_poison import poisoned_allocations

    with poisoned_allocations():
        yield


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from localmd_amd._lib import Context

    ctx = Context(0)
    yield ctx
    ctx.close()
