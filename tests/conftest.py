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
