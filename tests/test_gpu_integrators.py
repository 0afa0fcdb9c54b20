"""`direct` and `depth` integrators on the GPU against the oracle, sample by sample (same PCG32 streams)."""
import numpy as np
import parity_util
import pytest
import torch

import oracle_binding as ob

pytestmark = pytest.mark.gpu


def _cbox():
    from mitsuba2_amd import scenes
    cb = scenes.cornell_box()
    cb["bsdfs"] = list(cb["bsdfs"]) + [{"type": "roughconductor", "alpha": 0.3, "distribution": "ggx", "eta": 0.0, "k": 1.0},
                                       {"type": "plastic", "diffuse_reflectance": [0.2, 0.3, 0.5]}]
    cb["meshes"][6] = dict(cb["meshes"][6], bsdf=4)
    cb["meshes"][7] = dict(cb["meshes"][7], bsdf=5)
    return cb


@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("cfg", [dict(), dict(emitter_samples=3, bsdf_samples=2), dict(emitter_samples=0, bsdf_samples=2),
                                 dict(shading_samples=2, hide_emitters=True)])
def test_direct_matches_oracle(general, cfg):
    from mitsuba2_amd import render as R, scenes
    cb = _cbox() if general else scenes.cornell_box()
    sp = scenes.cornell_box_sensor(64, 48, spp=4, seed=13)
    scene, sensor = R.Scene(cb), R.make_sensor(sp)
    integ = R.DirectIntegrator(**cfg)
    n = 64 * 48 * 4
    rgb, mask, pos = integ.sample(scene, sensor, 0, n)
    op = dict(sp, integrator="direct", emitter_samples=integ.emitter_samples, bsdf_samples=integ.bsdf_samples, hide_emitters=integ.hide_emitters)
    want, wpos = ob.OracleScene(cb).sample_radiance(ob.make_desc(op), 0, n)
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(mask.cpu().numpy(), want[:, 3] > 0.5)
    # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
    # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
    parity_util.check("per-sample radiance", rgb.cpu().numpy(), want[:, :3])


def test_direct_film_and_stats():
    from mitsuba2_amd import render as R, scenes
    cb, sp = scenes.cornell_box(), scenes.cornell_box_sensor(48, 48, spp=16, seed=1)
    scene, sensor = R.Scene(cb), R.make_sensor(sp)
    integ = R.DirectIntegrator()
    assert integ.render(scene, sensor)
    got = sensor.film().bitmap().cpu().numpy()[..., :3]
    film, _ = ob.OracleScene(cb).render(ob.make_desc(dict(sp, integrator="direct")), mode=1)
    want = ob.film_develop(film)[..., :3]
    assert np.mean((got - want) ** 2 / (want ** 2 + 1e-2)) < 1e-5
    assert integ.stats["samples"] == 48 * 48 * 16 and integ.stats["closest_hit_rays"] > integ.stats["samples"]
    with pytest.raises(RuntimeError, match="Cannot specify both"):
        R.DirectIntegrator(shading_samples=2, bsdf_samples=1)
    with pytest.raises(RuntimeError, match="at least 1 BSDF or emitter sample"):
        R.DirectIntegrator(emitter_samples=0, bsdf_samples=0)


def test_depth_matches_ray_intersect():
    from mitsuba2_amd import render as R, scenes
    sd, sp = scenes.bumpy_sphere(32, 64), scenes.bumpy_sphere_sensor(64, 48, 2)
    scene, sensor = R.Scene(sd), R.make_sensor(sp)
    n = 64 * 48 * 2
    rgb, mask, pos = R.DepthIntegrator().sample(scene, sensor, 0, n)
    want, wpos = ob.OracleScene(sd).sample_radiance(ob.make_desc(dict(sp, integrator="depth")), 0, n)
    assert np.array_equal(pos.cpu().numpy(), wpos)
    assert np.array_equal(rgb.cpu().numpy(), want[:, :3]) and np.array_equal(mask.cpu().numpy(), want[:, 3] > 0.5)


def test_moment_integrator_and_z_test():
    """src/integrators/moment.cpp:56-99 + the protocol of test_renders.py:60-134: the GPU render (wavefront seeding) must
    be statistically indistinguishable from the oracle's scalar_rgb block-mode samples, pixel by pixel."""
    from mitsuba2_amd import render as R, scenes
    import render_stats as testing
    cb = scenes.cornell_box()
    sp = dict(scenes.cornell_box_sensor(32, 32, spp=256, seed=3), rfilter="box", rfilter_param=0.5)
    scene, sensor = R.Scene(cb), R.make_sensor(sp)
    integ = R.MomentIntegrator(R.PathIntegrator(max_depth=6))
    assert integ.aov_channels() == ["X", "Y", "Z", "A", "W", "nested.X", "nested.Y", "nested.Z", "m2_nested.X", "m2_nested.Y", "m2_nested.Z"]
    assert integ.render(scene, sensor)
    raw = sensor.film().bitmap(raw=True).cpu().numpy()
    assert raw.shape == (32, 32, 11) and np.array_equal(raw[..., 5:8], raw[..., 0:3]) and (raw[..., 4] == 256).all()
    mean, var = (x.cpu().numpy() for x in integ.mean_and_variance(sensor.film()))
    # second moments against the per-sample API (box filter: every sample lands in its own pixel with weight 1)
    rgb, _, _ = R.PathIntegrator(max_depth=6).sample(scene, sensor, 0, 32 * 32 * 256)
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], np.float32)
    xyz = (rgb.cpu().numpy() @ m.T).reshape(32, 32, 256, 3)
    assert np.allclose(mean, xyz.mean(2), rtol=1e-4, atol=1e-5)
    assert np.allclose(raw[..., 8:11] / 256, (xyz.astype(np.float64) ** 2).mean(2), rtol=1e-4, atol=1e-6)
    assert np.allclose(var, xyz.astype(np.float64).var(2), rtol=2e-2, atol=1e-3)      # m2 - mean^2 cancels in fp32
    # reference mean / variance: a different set of random numbers (the oracle's block seeding, another seed), 1024 spp
    ref_sp = dict(sp, sample_count=1024, seed=77, max_depth=6)
    film, _ = ob.OracleScene(cb).render(ob.make_desc(ref_sp), mode=0)
    ref_mean = film[..., :3] / film[..., 4:5]
    vals, _ = ob.OracleScene(cb).sample_radiance(ob.make_desc(ref_sp), 0, 32 * 32 * 1024)
    ref_var = (vals[:, :3] @ m.T).reshape(32, 32, 1024, 3).var(2)
    ok, p_min, alpha = testing.accept(mean, 256, ref_mean, ref_var)
    assert ok, (p_min, alpha)
    bad, _, _ = testing.accept(mean * 1.05, 256, ref_mean, ref_var)      # a 5 % bias is detected
    assert not bad


def _open_scene(with_area):
    """the Cornell box without ceiling and back wall under a `constant` environment (optionally keeping the area light)"""
    from mitsuba2_amd import scenes
    cb = scenes.cornell_box()
    keep = [i for i, m in enumerate(cb["meshes"]) if i not in (1, 2) and (with_area or m.get("emitter", -1) < 0)]
    cb["meshes"] = [dict(cb["meshes"][i]) for i in keep]
    cb["bsdfs"] = list(cb["bsdfs"]) + [{"type": "roughconductor", "alpha": 0.2, "distribution": "ggx", "eta": 0.0, "k": 1.0}, {"type": "dielectric"}]
    cb["meshes"][-1]["bsdf"] = len(cb["bsdfs"]) - 2
    cb["meshes"][-2]["bsdf"] = len(cb["bsdfs"]) - 1
    if with_area:
        cb["emitters"] = [{"type": "constant", "radiance": [0.4, 0.6, 1.0]}] + list(cb["emitters"])     # environment first: index 0
        for m in cb["meshes"]:
            if m.get("emitter", -1) >= 0:
                m["emitter"] = 1
    else:
        cb["emitters"] = [{"type": "constant", "radiance": [0.4, 0.6, 1.0]}]
    return cb


@pytest.mark.parametrize("with_area", [False, True])
@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_constant_emitter_matches_oracle(with_area, integrator):
    """src/emitters/constant.cpp through path and direct: escaped rays see the environment, emitter sampling picks it with
    probability 1 / #emitters, MIS against BSDF sampling"""
    from mitsuba2_amd import render as R, scenes
    cb = _open_scene(with_area)
    sp = dict(scenes.cornell_box_sensor(64, 64, spp=4, seed=8), max_depth=5)
    scene, sensor = R.Scene(cb), R.make_sensor(sp)
    integ = R.PathIntegrator(max_depth=5) if integrator == "path" else R.DirectIntegrator(shading_samples=2)
    n = 64 * 64 * 4
    rgb, mask, pos = integ.sample(scene, sensor, 0, n)
    op = dict(sp, integrator=integrator, emitter_samples=2 if integrator == "direct" else 0, bsdf_samples=2 if integrator == "direct" else 0)
    want, wpos = ob.OracleScene(cb).sample_radiance(ob.make_desc(op), 0, n)
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(mask.cpu().numpy(), want[:, 3] > 0.5)
    assert (~mask.cpu().numpy()).any() and (rgb.cpu().numpy()[~mask.cpu().numpy()] == np.float32([0.4, 0.6, 1.0])).all()
    # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
    # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
    parity_util.check("per-sample radiance", rgb.cpu().numpy(), want[:, :3])
    if integrator == "path":
        a, _, _ = R.PathIntegrator(max_depth=5, pipeline=2).sample(scene, sensor, 0, n)
        assert torch.equal(a, rgb)


def test_constant_emitter_restrictions():
    from mitsuba2_amd import render as R
    cb = _open_scene(False)
    cb["emitters"] = cb["emitters"] * 2
    with pytest.raises(RuntimeError, match="Only one environment emitter"):
        R.Scene(cb)


def _envmap_image():
    rng = np.random.default_rng(5)
    img = rng.uniform(0.05, 1.0, size=(24, 48, 3)).astype(np.float32)
    img[5:8, 30:34] += 30.0                                   # a small bright region: importance sampling matters
    return img


@pytest.mark.parametrize("with_area", [False, True])
@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_envmap_emitter_matches_oracle(with_area, integrator):
    """src/emitters/envmap.cpp: hierarchical sample warping (distr_2d.h), lat-long lookup, MIS with BSDF sampling"""
    from mitsuba2_amd import render as R, scenes
    cb = _open_scene(with_area)
    rot = scenes.look_at([0, 0, 0], [1, 0.2, 0.3], [0, 1, 0])
    env = {"type": "envmap", "data": _envmap_image(), "scale": 0.7, "to_world": rot}
    cb["emitters"] = [env] + [e for e in cb["emitters"] if e.get("type", "area") == "area"]
    sp = dict(scenes.cornell_box_sensor(64, 64, spp=4, seed=17), max_depth=5)
    scene, sensor = R.Scene(cb), R.make_sensor(sp)
    integ = R.PathIntegrator(max_depth=5) if integrator == "path" else R.DirectIntegrator(shading_samples=2)
    n = 64 * 64 * 4
    rgb, mask, pos = integ.sample(scene, sensor, 0, n)
    op = dict(sp, integrator=integrator, emitter_samples=2 if integrator == "direct" else 0, bsdf_samples=2 if integrator == "direct" else 0)
    want, wpos = ob.OracleScene(cb).sample_radiance(ob.make_desc(op), 0, n)
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(mask.cpu().numpy(), want[:, 3] > 0.5)
    # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
    # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
    parity_util.check("per-sample radiance", rgb.cpu().numpy(), want[:, :3])
    assert abs(rgb.cpu().numpy().mean() - want[:, :3].mean()) < 0.02 * want[:, :3].mean()
    if integrator == "path":
        a, _, _ = R.PathIntegrator(max_depth=5, pipeline=2).sample(scene, sensor, 0, n)
        assert torch.equal(a, rgb)


def test_uniform_envmap_equals_constant_emitter():
    """an envmap of constant colour and a `constant` emitter of the same radiance light the scene identically (up to noise)"""
    from mitsuba2_amd import render as R, scenes
    sp = scenes.cornell_box_sensor(32, 32, spp=256, seed=3)
    imgs = []
    for em in ({"type": "constant", "radiance": [0.4, 0.6, 1.0]},
               {"type": "envmap", "data": np.tile(np.float32([0.4, 0.6, 1.0]), (8, 16, 1))}):
        cb = _open_scene(False)
        cb["emitters"] = [em]
        scene, sensor = R.Scene(cb), R.make_sensor(sp)
        assert R.PathIntegrator(max_depth=4).render(scene, sensor)
        imgs.append(sensor.film().bitmap().cpu().numpy()[..., :3])
    assert abs(imgs[0].mean() - imgs[1].mean()) < 0.01 * imgs[0].mean()
    assert np.mean((imgs[0] - imgs[1]) ** 2 / (imgs[0] ** 2 + 1e-2)) < 5e-3


def _delta_lights():
    from mitsuba2_amd import scenes
    spot_tw = scenes.look_at([278, 500, 200], [300, 0, 320], [0, 0, 1])
    return {
        "point": [{"type": "point", "position": [278, 400, 279], "intensity": [4e5, 3e5, 2e5]}],
        "spot": [{"type": "spot", "to_world": spot_tw, "intensity": [9e5, 9e5, 6e5], "cutoff_angle": 35.0, "beam_width": 20.0}],
        "directional": [{"type": "directional", "direction": [0.3, -1.0, 0.4], "irradiance": [3.0, 2.5, 2.0]}],
        "mixed": [{"type": "point", "position": [100, 300, 100], "intensity": [2e5, 2e5, 3e5]},
                  {"type": "spot", "to_world": spot_tw, "intensity": [5e5, 3e5, 3e5]},
                  {"type": "directional", "direction": [0.3, -1.0, 0.4], "irradiance": [1.0, 1.0, 1.0]}],
    }


@pytest.mark.parametrize("lights", ["point", "spot", "directional", "mixed", "mixed+area"])
@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_delta_emitters_match_oracle(lights, integrator):
    """src/emitters/point.cpp, spot.cpp, directional.cpp: pdf 1 / MIS weight 1 (path.cpp:170, direct.cpp:155-156), never hit by
    BSDF sampling; alone, mixed and next to the area light; RGB and spectral variant; fused == split"""
    from mitsuba2_amd import render as R, scenes
    cb = _open_scene(lights.endswith("+area"))
    area = [e for e in cb["emitters"] if e.get("type", "area") == "area"]
    cb["emitters"] = _delta_lights()[lights.split("+")[0]] + area
    for m in cb["meshes"]:
        if m.get("emitter", -1) >= 0:
            m["emitter"] = len(cb["emitters"]) - 1
    sp = dict(scenes.cornell_box_sensor(64, 64, spp=4, seed=23), max_depth=5)
    scene, sensor = R.Scene(cb), R.make_sensor(sp)
    integ = R.PathIntegrator(max_depth=5) if integrator == "path" else R.DirectIntegrator(emitter_samples=2, bsdf_samples=1)
    n = 64 * 64 * 4
    rgb, mask, pos = integ.sample(scene, sensor, 0, n)
    op = dict(sp, integrator=integrator, emitter_samples=2 if integrator == "direct" else 0, bsdf_samples=1 if integrator == "direct" else 0)
    want, wpos = ob.OracleScene(cb).sample_radiance(ob.make_desc(op), 0, n)
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(mask.cpu().numpy(), want[:, 3] > 0.5)
    got = rgb.cpu().numpy()
    assert want[:, :3].mean() > 1e-3
    # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
    # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
    parity_util.check("per-sample radiance", got, want[:, :3])
    assert abs(got.mean() - want[:, :3].mean()) < 0.02 * want[:, :3].mean()
    if integrator == "path":
        a, _, _ = R.PathIntegrator(max_depth=5, pipeline=2).sample(scene, sensor, 0, n)
        assert torch.equal(a, rgb)
        sscene = R.Scene(cb, variant="spectral")
        xyz, _, _ = R.PathIntegrator(max_depth=5, pipeline=1).sample(sscene, sensor, 0, n)
        xyz2, _, _ = R.PathIntegrator(max_depth=5, pipeline=2).sample(sscene, sensor, 0, n)
        assert torch.equal(xyz, xyz2)
        swant, _ = ob.OracleScene(cb, spectral_path=R.srgb_coeff_path()).sample_radiance(ob.make_desc(sp), 0, n)
        # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
        # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
        parity_util.check("per-sample radiance", xyz.cpu().numpy(), swant[:, :3])


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_cluster_culling_on_random_flat_scenes(seed):
    """LDS-resident scenes: the camera-path chunks take the cluster-culling loops (device_scene.h, traverse_flat_clustered*): boxes of the
    consecutive pairs of one shape, skipped by the whole wave.  The boxes must only ever cull: random triangle soups with odd triangle
    counts per shape (pairs straddle shapes), slivers, a camera inside the geometry and an axis-parallel view -- the depth integrator
    (hit distance per sample) is bit-exact against the oracle's brute force, path samples agree as for every other scene."""
    from mitsuba2_amd import render as R, scenes
    rng = np.random.default_rng(100 + seed)
    meshes, counts = [], [1, 3, 7, 2, 5, 9, 4][: 4 + seed % 3]
    for si, nt in enumerate(counts):
        centre = rng.uniform(-2.0, 2.0, 3)
        tris = centre + rng.normal(size=(nt, 3, 3)) * rng.uniform(0.05, 1.5)
        if si == 1:
            tris[0, 2] = tris[0, 0] + (tris[0, 1] - tris[0, 0]) * 0.5 + 1e-4       # a sliver
        meshes.append(dict(positions=tris.reshape(-1, 3).astype(np.float32), faces=np.arange(3 * nt, dtype=np.uint32).reshape(-1, 3),
                           bsdf=si % 2, emitter=0 if si == 0 else -1))
    sd = dict(meshes=meshes, bsdfs=[{"type": "diffuse", "reflectance": [0.6, 0.5, 0.4]}, {"type": "diffuse", "reflectance": [0.2, 0.7, 0.3]}],
              emitters=[dict(type="area", radiance=np.array([9.0, 8.0, 7.0], np.float32))])
    allp = np.concatenate([m["positions"] for m in meshes])
    mid = allp.mean(0)
    c0, c1 = meshes[2]["positions"].mean(0), meshes[3]["positions"].mean(0)
    # from outside; from inside one shape's cloud towards another; exactly along -x (zero direction components in the box tests)
    views = [(mid + [0.3, 0.2, -8.0], mid), (c0, c1), (mid + [7.0, 0.0, 0.0], mid)]
    origin, target = [[float(x) for x in v] for v in views[seed % 3]]
    sp = dict(scenes.cornell_box_sensor(48, 32, spp=64, seed=seed), to_world=scenes.look_at(origin, target, [0, 1, 0]), fov=50.0,
              near_clip=1e-2, far_clip=100.0, max_depth=4)
    scene, sensor = R.Scene(sd), R.make_sensor(sp)
    assert scene.info()["primitives"] == sum(counts) and scene.info()["primitives"] <= 64
    n = 48 * 32 * 64
    oracle = ob.OracleScene(sd, naive=True)
    rgb, mask, pos = R.DepthIntegrator().sample(scene, sensor, 0, n)
    want, wpos = oracle.sample_radiance(ob.make_desc(dict(sp, integrator="depth")), 0, n)
    assert np.array_equal(pos.cpu().numpy(), wpos)
    assert np.array_equal(mask.cpu().numpy(), want[:, 3] > 0.5) and np.array_equal(rgb.cpu().numpy(), want[:, :3])
    assert mask.float().mean().item() > 0.01                 # the view sees some geometry
    for pipeline in (0, 1):                                  # default schedule (culling + on-the-spot shadow rays) and the fused kernel
        rgb, mask, pos = R.PathIntegrator(max_depth=4, pipeline=pipeline).sample(scene, sensor, 0, n)
        want, wpos = oracle.sample_radiance(ob.make_desc(sp), 0, n)
        assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(mask.cpu().numpy(), want[:, 3] > 0.5)
        # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
        # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
        parity_util.check("per-sample radiance", rgb.cpu().numpy(), want[:, :3])
