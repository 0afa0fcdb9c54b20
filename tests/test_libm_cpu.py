"""oracle/mo_libm.h: the explicit-fma Cephes restatements that the oracle (and, bit for bit, the HIP kernels through
csrc/device_libm.h) use instead of libm / OCML.  The reference takes these functions from Enoki, which is absent from
/root/reference: "parity unpinned" against Enoki's own coefficients; pinned here against float64 numpy to a few ulp."""
import numpy as np
import pytest


def _ulp(got, want64):
    want32 = want64.astype(np.float32)
    spacing = np.spacing(np.abs(want32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - want64) / spacing


@pytest.mark.parametrize("name,lo,hi,ref,max_ulp", [
    ("sin", -20.0, 20.0, np.sin, 2.0), ("cos", -20.0, 20.0, np.cos, 2.0), ("tan", -1.5, 1.5, np.tan, 3.0),
    ("exp", -80.0, 80.0, np.exp, 1.5), ("acos", -1.0, 1.0, np.arccos, 2.0),
    ("atanh", -0.999, 0.999, np.arctanh, 3.0), ("cosh", -10.0, 10.0, np.cosh, 2.5)])
def test_against_float64(oracle, name, lo, hi, ref, max_ulp):
    rng = np.random.RandomState(7)
    x = rng.uniform(lo, hi, 400000).astype(np.float32)
    got = oracle.libm_eval(name, x)
    want = ref(x.astype(np.float64))
    keep = np.abs(want) > 1e-3                              # ulp is meaningless next to a zero crossing
    assert _ulp(got[keep], want[keep]).max() <= max_ulp
    assert keep.all() or np.abs(got[~keep] - want[~keep]).max() < 2e-7


def test_log_erf_atan2(oracle):
    rng = np.random.RandomState(9)
    x = np.ldexp(rng.uniform(0.5, 1.0, 400000), rng.randint(-40, 40, 400000)).astype(np.float32)
    got, want = oracle.libm_eval("log", x), np.log(x.astype(np.float64))
    keep = np.abs(want) > 1e-2
    assert _ulp(got[keep], want[keep]).max() <= 1.5
    import math
    xe = rng.uniform(-6.0, 6.0, 200000).astype(np.float32)
    want = np.array([math.erf(float(v)) for v in xe])
    assert np.abs(oracle.libm_eval("erf", xe) - want).max() < 2.5e-7
    yy, xx = rng.uniform(-1, 1, 400000).astype(np.float32), rng.uniform(-1, 1, 400000).astype(np.float32)
    assert np.abs(oracle.libm_eval("atan2", yy, xx) - np.arctan2(yy.astype(np.float64), xx.astype(np.float64))).max() < 6e-7


def test_special_values(oracle):
    f = np.float32
    assert oracle.libm_eval("exp", f([0.0, -100.0, 89.0]))[0] == 1.0
    e = oracle.libm_eval("exp", f([0.0, -100.0, 89.0, np.nan]))
    assert e[1] == 0.0 and np.isinf(e[2]) and np.isnan(e[3])
    l = oracle.libm_eval("log", f([1.0, 0.0, -1.0, np.inf]))
    assert l[0] == 0.0 and l[1] == -np.inf and np.isnan(l[2]) and l[3] == np.inf
    assert (oracle.libm_eval("erf", f([0.0, 20.0, -20.0])) == f([0.0, 1.0, -1.0])).all()
    a = oracle.libm_eval("atan2", f([0.0, -0.0, 1.0, -1.0]), f([-1.0, -1.0, 0.0, 0.0]))
    assert np.allclose(a, [np.pi, -np.pi, np.pi / 2, -np.pi / 2], rtol=1e-7)
    assert (oracle.libm_eval("acos", f([1.0, -1.0])) == f([0.0, np.pi])).all()
    s, c = oracle.libm_eval("sin", f([0.0])), oracle.libm_eval("cos", f([0.0]))
    assert s[0] == 0.0 and c[0] == 1.0
    # the concentric disk warp's range: phi in [-pi/4, 3 pi/4] (include/mitsuba/core/warp.h:54-90)
    phi = np.linspace(-np.pi / 4, 3 * np.pi / 4, 100001).astype(np.float32)
    assert np.abs(oracle.libm_eval("sin", phi) ** 2 + oracle.libm_eval("cos", phi) ** 2 - 1.0).max() < 3e-7
