"""The fused kernel (k_bounce) and the split pipeline (k_trace<closest> + k_shade + k_trace<any>) are two schedules of the
same path.cpp iteration: same PCG32 streams, same order of floating-point additions into the radiance, so their films
must agree bit for bit -- on the LDS-resident Cornell box (where the split pipeline walks the BVH instead of the flat
primitive loop) and on a hierarchy scene, RGB and spectral."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(scene_dict, sensor_params, pipeline, variant="rgb", max_depth=-1, paths_per_wave=0):
    from mitsuba2_amd import render as R
    scene = R.Scene(scene_dict, variant=variant)
    sensor = R.make_sensor(sensor_params)
    integ = R.PathIntegrator(max_depth=max_depth, pipeline=pipeline, paths_per_wave=paths_per_wave)
    assert integ.render(scene, sensor)
    return sensor.film().bitmap(raw=True).cpu().numpy(), integ.stats


@pytest.mark.parametrize("variant", ["rgb", "spectral"])
def test_cbox_fused_equals_split(variant):
    from mitsuba2_amd import scenes
    cb, sp = scenes.cornell_box(), scenes.cornell_box_sensor(96, 96, spp=16, seed=3)
    a, sa = _render(cb, sp, 1, variant)
    b, sb = _render(cb, sp, 2, variant)
    q, sq = _render(cb, sp, 3, variant)          # closest hit fused, shadow rays queued
    assert np.array_equal(a, b) and np.array_equal(a, q) and sq["any_hit_rays"] == sa["any_hit_rays"] and sq["tri_tests"] == sa["tri_tests"]
    r, sr = _render(cb, sp, 4, variant)          # one kernel, shadow rays resolved 64 at a time from a per-wave LDS ring
    assert np.array_equal(a, r) and sr["any_hit_rays"] == sa["any_hit_rays"] and sr["tri_tests"] == sa["tri_tests"]
    r2, _ = _render(cb, sp, 4, variant, paths_per_wave=100)      # segments that are not a multiple of the wave size
    assert np.array_equal(a, r2)
    for k in ("closest_hit_rays", "any_hit_rays", "samples", "segments"):
        assert sa[k] == sb[k]
    assert a[..., 4].min() > 0 and np.isfinite(a).all() and a[..., :3].max() > 0


@pytest.mark.parametrize("variant", ["rgb", "spectral"])
def test_mesh_fused_equals_split(variant):
    from mitsuba2_amd import scenes
    sd, sp = scenes.bumpy_sphere(48, 96), scenes.bumpy_sphere_sensor(128, 96, 8)
    a, sa = _render(sd, sp, 1, variant)
    b, sb = _render(sd, sp, 2, variant)           # what pipeline = 0 selects for this scene
    c, _ = _render(sd, sp, 0, variant, paths_per_wave=64)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert sa["segments"] == sb["segments"] and sa["any_hit_rays"] == sb["any_hit_rays"]
    assert sb["tri_tests"] > 0


def test_split_max_depth_and_textures():
    """depth-limited paths and a bitmap reflectance through the split pipeline"""
    from mitsuba2_amd import scenes
    tex = np.random.default_rng(1).uniform(0.1, 0.9, size=(16, 16, 3)).astype(np.float32)
    cb, sp = scenes.cornell_box(texture=tex), scenes.cornell_box_sensor(64, 64, spp=8, seed=11)
    for depth in (1, 2, 5):
        a, _ = _render(cb, sp, 1, max_depth=depth)
        b, _ = _render(cb, sp, 2, max_depth=depth)
        assert np.array_equal(a, b)
        assert np.array_equal(a, _render(cb, sp, 4, max_depth=depth)[0])
