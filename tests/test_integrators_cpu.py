"""`direct` and `depth` integrators of the oracle (src/integrators/direct.cpp:105-196, depth.cpp:19-33).  The reference has no
golden values for them that are reproducible here (scenes.py:261-286 needs absent data files), so they are pinned through
identities that follow from the reference's own code: with one emitter and one BSDF sample `direct` computes what `path`
computes at max_depth = 2 (same random numbers, same MIS weights up to the common factor 1/2), `hide_emitters` removes
exactly the directly visible emission, and `depth` returns the intersection distance of the camera ray."""
import numpy as np
import pytest

import oracle_binding as ob
from mitsuba2_amd import scenes


def _cbox():
    cb = scenes.cornell_box()
    cb["bsdfs"] = list(cb["bsdfs"]) + [{"type": "roughconductor", "alpha": 0.3, "distribution": "ggx", "eta": 0.0, "k": 1.0},
                                       {"type": "plastic", "diffuse_reflectance": [0.2, 0.3, 0.5]}]
    cb["meshes"][6] = dict(cb["meshes"][6], bsdf=4)
    cb["meshes"][7] = dict(cb["meshes"][7], bsdf=5)
    return cb


def test_direct_equals_path_depth_two():
    cb, sp = _cbox(), scenes.cornell_box_sensor(48, 48, spp=4, seed=9)
    oracle = ob.OracleScene(cb)
    n = 48 * 48 * 4
    direct, pos_d = oracle.sample_radiance(ob.make_desc(dict(sp, integrator="direct")), 0, n)
    path, pos_p = oracle.sample_radiance(ob.make_desc(dict(sp, max_depth=2)), 0, n)
    assert np.array_equal(pos_d, pos_p) and np.array_equal(direct[:, 3], path[:, 3])
    assert np.allclose(direct[:, :3], path[:, :3], rtol=1e-5, atol=1e-6)
    assert direct[:, :3].max() > 1


def test_direct_sample_counts_and_hide_emitters():
    cb, sp = _cbox(), scenes.cornell_box_sensor(32, 32, spp=64, seed=2)
    oracle = ob.OracleScene(cb)
    n = 32 * 32 * 64
    base, _ = oracle.sample_radiance(ob.make_desc(dict(sp, integrator="direct")), 0, n)
    hidden, _ = oracle.sample_radiance(ob.make_desc(dict(sp, integrator="direct", hide_emitters=True)), 0, n)
    path1, _ = oracle.sample_radiance(ob.make_desc(dict(sp, max_depth=1)), 0, n)       # visible emitters only
    assert np.allclose(base[:, :3] - hidden[:, :3], path1[:, :3], atol=2e-5)
    means = []
    for es, bs in ((4, 0), (0, 4), (3, 2)):                                           # every mix of techniques is unbiased
        v, _ = oracle.sample_radiance(ob.make_desc(dict(sp, integrator="direct", emitter_samples=es, bsdf_samples=bs, hide_emitters=True)), 0, n)
        means.append(v[:, :3].reshape(32 * 32, 64, 3).mean(1))
    ref = hidden[:, :3].reshape(32 * 32, 64, 3).mean(1)
    for m in means:
        assert abs(m.mean() - ref.mean()) < 0.03 * ref.mean()


def test_depth_is_hit_distance():
    cb, sp = _cbox(), scenes.cornell_box_sensor(32, 32, spp=2, seed=5)
    oracle = ob.OracleScene(cb)
    desc = ob.make_desc(dict(sp, integrator="depth"))
    n = 32 * 32 * 2
    v, pos = oracle.sample_radiance(desc, 0, n)
    o, d, mint, maxt = ob.camera_rays(desc, (pos[:, 0] - sp["crop"][0]) / sp["crop"][2], (pos[:, 1] - sp["crop"][1]) / sp["crop"][3])
    t, prim, _, _, _ = oracle.ray_intersect(o, d, mint, maxt)
    want = np.where(np.isfinite(t), t, 0).astype(np.float32)
    assert np.array_equal(v[:, 0], want) and np.array_equal(v[:, 1], want) and np.array_equal(v[:, 3] > 0.5, np.isfinite(t))


def _lit_floor(emitter):
    """a 2 x 2 diffuse floor in the plane y = 0 seen from above, lit by one delta emitter"""
    from mitsuba2_amd import scenes
    floor = dict(positions=np.float32([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]]), faces=np.uint32([[0, 2, 1], [0, 3, 2]]), bsdf=0, emitter=-1)
    sd = dict(meshes=[floor], bsdfs=[{"type": "diffuse", "reflectance": [0.5, 0.5, 0.5]}], emitters=[emitter])
    sp = dict(to_world=scenes.look_at([0, 4, 0], [0, 0, 0], [0, 0, 1]), fov=25.0, near_clip=0.1, far_clip=100.0, width=16, height=16,
              crop=(0, 0, 16, 16), rfilter="box", rfilter_param=0.5, sample_count=4, seed=1, max_depth=2, rr_depth=5)
    return sd, sp


def test_delta_emitters_analytic():
    """point.cpp:76-101, spot.cpp:95-151, directional.cpp:104-129 through the path integrator: direct illumination of a diffuse
    floor equals rho / pi * cos(theta) * I / d^2 (point), times the falloff curve (spot), or rho / pi * cos(theta) * E
    (directional); delta emitters are never hit by BSDF sampling and get MIS weight 1 (path.cpp:170)"""
    import oracle_binding as ob
    rho = 0.5
    # point light at height 2 above the origin
    sd, sp = _lit_floor({"type": "point", "position": [0, 2, 0], "intensity": [10.0, 20.0, 30.0]})
    S = ob.OracleScene(sd)
    n = 16 * 16 * 4
    rgb, pos = S.sample_radiance(ob.make_desc(sp), 0, n)
    hit = rgb[:, 3] > 0.5
    assert hit.all()
    # world position of each sample on the floor: the camera looks down -y from (0, 4, 0)
    ro, rd, _, _ = ob.camera_rays(ob.make_desc(sp), pos[:, 0] / 16, pos[:, 1] / 16)
    p = ro + rd * (-ro[:, 1:2] / rd[:, 1:2])
    d2 = (p[:, 0] ** 2 + p[:, 2] ** 2 + 4.0)
    cos = 2.0 / np.sqrt(d2)
    want = (rho / np.pi) * cos / d2
    assert np.allclose(rgb[:, 0], want * 10.0, rtol=2e-4) and np.allclose(rgb[:, 2], want * 30.0, rtol=2e-4)
    # the same light as a spot pointing straight down: full intensity inside the beam, linear falloff, zero outside
    down = np.eye(4, dtype=np.float32)
    down[:3, :3] = np.float32([[1, 0, 0], [0, 0, -1], [0, 1, 0]]).T @ np.eye(3, dtype=np.float32)      # local +z -> world -y
    down[:3, 2] = [0, -1, 0]; down[:3, 1] = [0, 0, 1]; down[:3, 0] = [1, 0, 0]
    down[:3, 3] = [0, 2, 0]
    sd2, _ = _lit_floor({"type": "spot", "to_world": down, "intensity": [10.0, 20.0, 30.0], "cutoff_angle": 20.0, "beam_width": 10.0})
    rgb2, _ = ob.OracleScene(sd2).sample_radiance(ob.make_desc(sp), 0, n)
    theta = np.arccos(cos)
    fall = np.where(theta <= np.radians(10.0), 1.0, np.clip((np.radians(20.0) - theta) / np.radians(10.0), 0.0, 1.0))
    assert np.allclose(rgb2[:, 0], want * 10.0 * fall, rtol=2e-3, atol=1e-5) and (fall == 0).any() and (fall == 1).any()
    # directional light at 60 degrees from the normal
    d = np.float32([np.sin(np.radians(60.0)), -np.cos(np.radians(60.0)), 0.0])
    sd3, _ = _lit_floor({"type": "directional", "direction": d.tolist(), "irradiance": [3.0, 2.0, 1.0]})
    rgb3, _ = ob.OracleScene(sd3).sample_radiance(ob.make_desc(sp), 0, n)
    assert np.allclose(rgb3[:, :3], (rho / np.pi) * 0.5 * np.float32([3.0, 2.0, 1.0]), rtol=2e-4)
    # two emitters: each is picked with probability 1/2 and weighted by 2 -> the expectation is the sum
    sd4, sp4 = _lit_floor({"type": "point", "position": [0, 2, 0], "intensity": [10.0, 20.0, 30.0]})
    sd4["emitters"].append({"type": "directional", "direction": d.tolist(), "irradiance": [3.0, 2.0, 1.0]})
    sp4 = dict(sp4, sample_count=256)
    rgb4, _ = ob.OracleScene(sd4).sample_radiance(ob.make_desc(sp4), 0, 16 * 16 * 256)
    mean4 = rgb4[:, 0].reshape(256, 256).mean(1) if False else rgb4[:, 0].reshape(-1, 256).mean(1)
    want4 = (rgb[:, 0].reshape(-1, 4).mean(1) + rgb3[:, 0].reshape(-1, 4).mean(1))
    assert np.allclose(mean4, want4, rtol=0.15)


def test_delta_emitter_parameters():
    from mitsuba2_amd import emitters as E
    assert E.normalize({"type": "spot"})["cutoff_angle"] == 20.0 and E.normalize({"type": "spot"})["beam_width"] == 15.0      # spot.cpp:81-82
    assert E.normalize({"type": "point"})["radiance"] == [1.0, 1.0, 1.0]
    with pytest.raises(RuntimeError, match="Only one of the parameters"):
        E.normalize({"type": "point", "position": [0, 0, 0], "to_world": np.eye(4)})
    with pytest.raises(RuntimeError, match="Only one of the parameters"):
        E.normalize({"type": "directional", "direction": [0, 0, 1], "to_world": np.eye(4)})
    n = E.normalize({"type": "directional", "direction": [0, 0, 2]})
    assert np.allclose(n["to_world"][:3, 2], [0, 0, 1]) and abs(np.linalg.det(n["to_world"][:3, :3]) - 1) < 1e-5
    with pytest.raises(RuntimeError, match="not supported by this backend"):
        E.normalize({"type": "projector"})
