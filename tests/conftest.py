import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu")
    config.addinivalue_line("markers", "slow: minutes of CPU oracle work; runs only with PMD_RUN_SLOW=1")


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from localmd_amd._lib import Context

    ctx = Context(0)
    yield ctx
    ctx.close()
