import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand from oracle/*.c."""
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def gpu():
    """The product: libmtsamd.so through mitsuba2_amd.render.  No CPU fallback: the fixture fails
    loudly when the HIP extension is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from mitsuba2_amd import render, _lib
    _lib.lib()
    return render


@pytest.fixture(scope="session", autouse=True)
def _dirty_device_memory(request):
    """GPU runs start from device memory that holds junk, not the zero pages a fresh process gets from the driver: a kernel that reads
    a workspace or scene buffer before anything was written to it (round 2: the empty scene's BVH4 root) then fails here instead of
    only after other scenes have used the memory."""
    markexpr = request.config.getoption("-m") or ""
    if "not gpu" in markexpr:
        return
    import torch
    if not torch.cuda.is_available():
        return
    free, _ = torch.cuda.mem_get_info()
    n = int(min(free // 2, 32 << 30)) // 4
    junk = torch.full((n,), 0x7f7f7f7f, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    del junk
    torch.cuda.empty_cache()
