import os
import sys

import pytest

TESTS = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (TESTS, ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _poisoned_allocator(request):
    """VC_POISON=1: before every GPU test, fill the memory the caching allocator will hand out next with NaN
    patterns (one large block and a spread of small ones, allocated, filled and freed).  A kernel that reads a
    buffer it was supposed to have written -- and only ever saw the zeros of fresh device memory -- then fails
    loudly instead of once in a while."""
    if os.environ.get("VC_POISON", "0") == "1" and request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            big = torch.full((int(os.environ.get("VC_POISON_MB", "6144")) << 18,), float("nan"), device="cuda")
            small = [torch.full((n,), float("nan"), device="cuda") for n in (64, 256, 1024, 4096, 16384, 65536) * 8]
            torch.cuda.synchronize()
            del big, small
    yield
