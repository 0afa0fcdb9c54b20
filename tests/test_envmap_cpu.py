"""Hierarchical2D0 and the `envmap` emitter of the oracle (oracle/mo_envmap.c).  Pinned by the reference's spot checks against
Mathematica (src/libcore/tests/test_distr_2d.py:8-60) and by consistency of sample / eval / pdf on a random map."""
import math

import numpy as np
import pytest

import oracle_binding as ob


@pytest.mark.parametrize("normalize", [True, False])
def test_hierarchical2d_spot_checks(normalize):
    """test_distr_2d.py:8-60 (test01_sample_inverse_discrete, Hierarchical2D0): mismatched X/Y resolution, odd column count"""
    ref = np.array([[1, 2, 5], [9, 7, 2]], np.float32)
    intg = np.array([19, 16]) / 35
    s = 35 / 8.0 if not normalize else 1

    def close(a, b):
        return np.allclose(a, b, atol=1e-6)

    sample = lambda p: ob.hier2d(ref, "sample", [p], normalize)[0]
    invert = lambda p: ob.hier2d(ref, "invert", [p], normalize)[0]
    evalf = lambda p: ob.hier2d(ref, "eval", [p], normalize)[0][2]
    assert close(sample([0, 0]), [0, 0, s * 8.0 / 35.0])
    assert close(sample([1, 1]), [1, 1, s * 16.0 / 35.0])
    assert close(sample([intg[0], 0]), [0.5, 0, s * 16.0 / 35.0])
    assert close(invert([0, 0]), [0, 0, s * 8.0 / 35.0])
    assert close(invert([1, 1]), [1, 1, s * 16.0 / 35.0])
    assert close(invert([0.5, 0]), [intg[0], 0, s * 16.0 / 35.0])
    sx, sy, pdf = ob.bilinear_to_square(1, 2, 9, 7, 0.4, 0.3)
    sx *= intg[0]; pdf *= 8.0 / 35.0 * s
    assert close(sample([sx, sy]), [0.2, 0.3, pdf]) and close(invert([0.2, 0.3]), [sx, sy, pdf]) and close(evalf([0.2, 0.3]), pdf)
    sx, sy, pdf = ob.bilinear_to_square(2, 5, 7, 2, 0.4, 0.3)
    sx = sx * intg[1] + intg[0]; pdf *= 8.0 / 35.0 * s
    assert close(sample([sx, sy]), [0.7, 0.3, pdf]) and close(invert([0.7, 0.3]), [sx, sy, pdf]) and close(evalf([0.7, 0.3]), pdf)


def test_hierarchical2d_roundtrip_and_density():
    rng = np.random.default_rng(3)
    data = rng.uniform(0.05, 1.0, size=(13, 22)).astype(np.float32) ** 3
    u = rng.uniform(size=(20000, 2)).astype(np.float32)
    s = ob.hier2d(data, "sample", u)
    back = ob.hier2d(data, "invert", s[:, :2])
    assert np.allclose(back[:, :2], u, atol=2e-4) and np.allclose(back[:, 2], s[:, 2], rtol=1e-4)
    assert np.allclose(ob.hier2d(data, "eval", s[:, :2])[:, 2], s[:, 2], rtol=1e-4)
    # the warped points follow the normalised bilinear interpolant: compare cell masses
    hist, _, _ = np.histogram2d(s[:, 1], s[:, 0], bins=(4, 7), range=((0, 1), (0, 1)))
    g = (np.arange(400) + 0.5) / 400
    gx, gy = np.meshgrid(g, g)
    dens = ob.hier2d(data, "eval", np.stack([gx.ravel(), gy.ravel()], 1))[:, 2].reshape(400, 400)
    want = np.array([[dens[int(j * 100):int((j + 1) * 100), int(i * 400 / 7):int((i + 1) * 400 / 7)].sum() for i in range(7)] for j in range(4)]) / 400 ** 2
    assert abs(dens.mean() - 1) < 1e-3
    assert np.allclose(hist / len(s), want, atol=6e-3)


def test_envmap_sample_eval_pdf():
    """envmap.cpp:132-208: sampled direction <-> lat-long coordinates, radiance / pdf, pdf_direction"""
    rng = np.random.default_rng(1)
    img = rng.uniform(0.0, 2.0, size=(16, 32, 3)).astype(np.float32)
    img[4:6, 10:13] += 40.0                                      # a bright "sun"
    u = rng.uniform(size=(50000, 2)).astype(np.float32)
    for to_world in (None, [[0, 0, 1, 0], [0, 1, 0, 0], [-1, 0, 0, 0], [0, 0, 0, 1]]):
        r = ob.envmap_kat(img, u, scale=0.5, to_world=to_world)
        assert np.allclose(np.linalg.norm(r["d"], axis=1), 1, atol=1e-5)
        ok = r["pdf"] > 1e-4
        assert ok.mean() > 0.99
        assert np.allclose(r["pdf"][ok], r["pdf_again"][ok], rtol=5e-3)
        assert np.allclose(r["spec"][ok] * r["pdf"][ok, None], r["eval"][ok], rtol=5e-3, atol=1e-4)
        # Monte Carlo estimate of the total emitted power from the samples vs quadrature of the map
        th, ph = np.meshgrid((np.arange(256) + 0.5) / 256 * math.pi, (np.arange(512) + 0.5) / 512 * 2 * math.pi, indexing="ij")
        # uv -> direction as in sample_direction (local frame; the rotation does not change the integral)
        est = r["spec"][ok].sum(0) / len(u)
        vv, uu = th / math.pi, ph / (2 * math.pi)
        x = uu * (img.shape[1] - 1); y = vv * (img.shape[0] - 1)
        x0 = np.minimum(x.astype(int), img.shape[1] - 2); y0 = np.minimum(y.astype(int), img.shape[0] - 2)
        fx, fy = (x - x0)[..., None], (y - y0)[..., None]
        val = ((img[y0, x0] * (1 - fx) + img[y0, x0 + 1] * fx) * (1 - fy) + (img[y0 + 1, x0] * (1 - fx) + img[y0 + 1, x0 + 1] * fx) * fy) * 0.5
        quad = (val * np.sin(th)[..., None]).sum((0, 1)) * (math.pi / 256) * (2 * math.pi / 512)
        assert np.allclose(est, quad, rtol=3e-2)
