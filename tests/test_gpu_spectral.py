"""Spectral variant on the GPU (BASELINE config 3 semantics): k_bounce_spectral against the oracle's spectral path on
identical sampler seeds and the same coefficient table."""
import numpy as np
import pytest

from mitsuba2_amd import scenes

pytestmark = pytest.mark.gpu


def _relmse(a, b):
    return float(np.mean((a - b) ** 2 / (b ** 2 + 1e-2)))


@pytest.mark.parametrize("scene_name", ["cbox", "sphere"])
def test_spectral_render_matches_oracle(gpu, oracle, scene_name):
    path = gpu.srgb_coeff_path()
    if scene_name == "cbox":
        sd, p = scenes.cornell_box(), scenes.cornell_box_sensor(64, 48, 8, seed=2)
    else:
        sd, p = scenes.bumpy_sphere(48, 96), scenes.bumpy_sphere_sensor(64, 48, 4)
    scene = gpu.Scene(sd, variant="spectral")
    sensor = gpu.make_sensor(p)
    integ = gpu.PathIntegrator()
    assert integ.render(scene, sensor)
    film = sensor.film().bitmap(raw=True).cpu().numpy()
    S = oracle.OracleScene(sd, naive=(scene_name == "cbox"), spectral_path=path)
    ref, stats = S.render(oracle.make_desc(p), mode=1)
    assert np.allclose(film[..., 3:], ref[..., 3:], rtol=1e-5, atol=1e-5)
    assert _relmse(film[..., :3], ref[..., :3]) < 1e-3          # north_star tolerance
    assert _relmse(film[..., :3], ref[..., :3]) < 1e-4
    assert integ.stats["samples"] == stats[2]
    # per-sample tristimulus values
    first = 500
    count = min(20000, p["width"] * p["height"] * p["sample_count"] - first)
    xyz, mask, pos = integ.sample(scene, sensor, first, count)
    ref_rgba, ref_pos = S.sample_radiance(oracle.make_desc(p), first, count)
    assert (pos.cpu().numpy() == ref_pos).all()
    close = np.isclose(xyz.cpu().numpy(), ref_rgba[:, :3], rtol=5e-3, atol=2e-4).all(axis=1)
    assert close.mean() > 0.995, close.mean()


def test_spectral_vs_rgb_and_errors(gpu):
    sd = scenes.cornell_box()
    p = scenes.cornell_box_sensor(48, 48, 64)
    out = {}
    for variant in ("rgb", "spectral"):
        scene, sensor = gpu.Scene(sd, variant=variant), gpu.make_sensor(p)
        assert gpu.PathIntegrator().render(scene, sensor)
        out[variant] = sensor.film().bitmap().cpu().numpy()
    a, b = out["rgb"], out["spectral"]
    assert np.allclose(a[..., 3], b[..., 3], atol=1e-5)
    assert abs(a[..., :3].mean() - b[..., :3].mean()) / a[..., :3].mean() < 0.1
    bad = scenes.cornell_box()
    bad["bsdfs"][0]["reflectance"] = np.array([1.5, 0.2, 0.2], np.float32)
    with pytest.raises(RuntimeError, match="Invalid RGB reflectance"):
        gpu.Scene(bad, variant="spectral")                               # srgb.cpp:34-35
    with pytest.raises(RuntimeError):
        gpu.Scene(scenes.cornell_box(texture=np.full((4, 4, 3), 0.5, np.float32)), variant="spectral")
    with pytest.raises(RuntimeError):
        gpu.Scene(sd, variant="polarized")
