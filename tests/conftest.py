import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference importable (this container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- -m gpu hardening (round 6): what a kernel reads must not depend on what the allocator's free blocks happen to hold -----------
# A fresh process hands out zero pages, so an out-of-bounds or uninitialised device read is invisible in a test run alone and
# shows up late in a long session (round 6: thx_vec_gather behind a mis-sized delta, a memory fault after ~250 tests).  With
# THX_TEST_POLLUTE=1 every GPU test starts with the caching allocator's free blocks filled with NaN patterns (as floats; huge
# indices as int32): such reads then misbehave in the test that makes them.
def _pollute_allocator():
    import torch
    keep = []
    for size_mb, count in ((1, 24), (2, 16), (8, 12), (32, 8), (128, 4), (512, 2)):
        for _ in range(count):
            keep.append(torch.full((size_mb * 1024 * 1024 // 4,), 0x7FC00000, dtype=torch.int32, device="cuda"))
    for small in (64, 512, 4096, 65536):
        for _ in range(64):
            keep.append(torch.full((small,), 0x7FC00000, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    del keep


@pytest.fixture(autouse=True)
def _polluted_free_blocks(request):
    if os.environ.get("THX_TEST_POLLUTE", "0") == "1" and request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            _pollute_allocator()
    yield
