"""Host and device evaluate the elementary functions with the same explicit operation sequence (oracle/mo_libm.h,
mitsuba2_amd/csrc/device_libm.h): the results must be equal bit for bit -- this is what makes per-sample radiance comparable
exactly instead of "to a few ulp" (include/mitsuba/core/warp.h:54-90, include/mitsuba/render/microfacet.h:187-493 call sites)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RANGES = {"sin": (-30.0, 30.0), "cos": (-30.0, 30.0), "tan": (-7.0, 7.0), "exp": (-100.0, 95.0), "erf": (-12.0, 12.0), "acos": (-1.0, 1.0),
          "atanh": (-1.0, 1.0), "cosh": (-20.0, 20.0)}


@pytest.mark.parametrize("name", ["sin", "cos", "tan", "exp", "log", "erf", "acos", "atan2", "atanh", "cosh"])
def test_device_equals_host_bitwise(gpu, oracle, name):
    rng = np.random.RandomState(11)
    n = 1 << 22
    if name == "log":
        x = np.ldexp(rng.uniform(0.5, 1.0, n), rng.randint(-126, 127, n)).astype(np.float32)
    elif name == "atan2":
        x = rng.uniform(-2, 2, n).astype(np.float32)
    else:
        lo, hi = RANGES[name]
        x = rng.uniform(lo, hi, n).astype(np.float32)
    y = rng.uniform(-2, 2, n).astype(np.float32)
    # specials ride along: zeros, infinities, NaN, the ends of the domain, subnormals
    special = np.float32([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 88.7, 88.8, -87.0, -87.1, 0.5, -0.5, 2.0, 10.0, 10.1])
    x[:special.size] = special
    y[:special.size] = special[::-1]
    got = gpu.libm_eval(name, torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda() if name == "atan2" else None).cpu().numpy()
    want = oracle.libm_eval(name, x, y if name == "atan2" else None)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), (name, int((~same).sum()), x[~same][:8], got[~same][:8], want[~same][:8])
