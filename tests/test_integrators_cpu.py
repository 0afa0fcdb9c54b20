"""`direct` and `depth` integrators of the oracle (src/integrators/direct.cpp:105-196, depth.cpp:19-33).  The reference has no
golden values for them that are reproducible here (scenes.py:261-286 needs absent data files), so they are pinned through
identities that follow from the reference's own code: with one emitter and one BSDF sample `direct` computes what `path`
computes at max_depth = 2 (same random numbers, same MIS weights up to the common factor 1/2), `hide_emitters` removes
exactly the directly visible emission, and `depth` returns the intersection distance of the camera ray."""
import numpy as np

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
