"""BSDF models on the GPU against the oracle: per-sample radiance of the path integrator with
conductor / roughconductor / dielectric / plastic / twosided materials in the Cornell box (same PCG32 stream per sample on
both sides; every operation shared bit for bit since round 3: csrc/device_libm.h == oracle/mo_libm.h), through the fused and the split
pipeline, plus the film-level relMSE bar of north_star (< 1e-3)."""
import numpy as np
import parity_util
import pytest
import torch

import oracle_binding as ob

pytestmark = pytest.mark.gpu

MATERIALS = {
    "conductor": {"type": "conductor", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]},
    "mirror": {"type": "conductor"},
    "rough_ggx": {"type": "roughconductor", "alpha": 0.2, "distribution": "ggx", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]},
    "rough_beckmann_aniso": {"type": "roughconductor", "alpha_u": 0.3, "alpha_v": 0.1, "distribution": "beckmann", "eta": 0.0, "k": 1.0,
                             "specular_reflectance": [0.9, 0.8, 0.7]},
    "rough_ggx_all": {"type": "roughconductor", "alpha": 0.25, "distribution": "ggx", "sample_visible": False, "eta": 0.0, "k": 1.0},
    "rough_beckmann_all": {"type": "roughconductor", "alpha": 0.25, "distribution": "beckmann", "sample_visible": False, "eta": 0.0, "k": 1.0},
    "glass": {"type": "dielectric", "int_ior": "bk7", "ext_ior": "air", "specular_transmittance": [0.9, 0.95, 1.0]},
    "thin_glass": {"type": "thindielectric", "int_ior": "bk7", "ext_ior": "air", "specular_transmittance": [0.9, 0.95, 1.0], "specular_reflectance": 0.8},
    "plastic": {"type": "plastic", "diffuse_reflectance": [0.1, 0.27, 0.36], "int_ior": 1.9},
    "plastic_nl": {"type": "plastic", "diffuse_reflectance": [0.5, 0.2, 0.1], "nonlinear": True, "specular_reflectance": 0.8},
    "roughplastic": {"type": "roughplastic", "alpha": 0.15, "diffuse_reflectance": [0.1, 0.27, 0.36], "int_ior": 1.9},
    "roughplastic_ggx": {"type": "roughplastic", "alpha": 0.3, "distribution": "ggx", "diffuse_reflectance": 0.4, "specular_reflectance": 0.7,
                         "nonlinear": True},
    "frosted_glass": {"type": "roughdielectric", "alpha": 0.2, "specular_transmittance": [0.9, 0.95, 1.0]},
    "frosted_ggx_aniso": {"type": "roughdielectric", "alpha_u": 0.3, "alpha_v": 0.1, "distribution": "ggx", "int_ior": "diamond",
                          "specular_reflectance": [0.9, 0.8, 0.7]},
    "frosted_beckmann_all": {"type": "roughdielectric", "alpha": 0.3, "sample_visible": False, "int_ior": 1.0, "ext_ior": 1.5},
    "twosided_diffuse": {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": [0.6, 0.3, 0.2]}},
    "twosided_rough": {"type": "twosided", "bsdf": {"type": "roughconductor", "alpha": 0.15, "distribution": "ggx", "eta": 0.0, "k": 1.0}},
    # src/bsdfs/blendbsdf.cpp / mask.cpp over plain children; constant and textured (Texture::eval_1) weights
    "blend_rough_diffuse": {"type": "blendbsdf", "weight": 0.3,
                            "bsdf_0": {"type": "roughconductor", "alpha": 0.2, "distribution": "ggx", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]},
                            "bsdf_1": {"type": "diffuse", "reflectance": [0.2, 0.5, 0.7]}},
    "blend_plastic_glass": {"type": "blendbsdf", "weight": 0.6, "a": {"type": "plastic", "diffuse_reflectance": [0.1, 0.27, 0.36]},
                            "b": {"type": "dielectric", "int_ior": "bk7", "specular_transmittance": [0.9, 0.95, 1.0]}},
    "twosided_blend_checker": {"type": "twosided", "bsdf": {"type": "blendbsdf",
                               "weight": {"type": "checkerboard", "color0": 0.9, "color1": 0.15, "to_uv": [[4.0, 0, 0.1, 0], [0, 3.0, 0.2, 0], [0, 0, 1, 0], [0, 0, 0, 1]]},
                               "bsdf_0": {"type": "diffuse", "reflectance": [0.7, 0.2, 0.1]}, "bsdf_1": {"type": "conductor"}}},
    "blend_bitmap_weight": {"type": "blendbsdf", "weight": {"type": "bitmap", "data": np.random.default_rng(5).uniform(0.0, 1.3, size=(6, 5, 3)).astype(np.float32)},
                            "bsdf_0": {"type": "diffuse", "reflectance": [0.7, 0.2, 0.1]},
                            "bsdf_1": {"type": "roughplastic", "alpha": 0.2, "diffuse_reflectance": [0.1, 0.3, 0.6]}},
    "mask_diffuse": {"type": "mask", "opacity": 0.4, "nested": {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": [0.6, 0.3, 0.2]}}},
    "mask_checker_glass": {"type": "mask", "opacity": {"type": "checkerboard", "color0": 1.0, "color1": 0.0, "to_uv": [[5.0, 0, 0, 0], [0, 5.0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]},
                           "nested": {"type": "roughdielectric", "alpha": 0.2}},
    "mask_default_opacity": {"type": "mask", "nested": {"type": "roughconductor", "alpha": 0.3, "eta": 0.0, "k": 1.0}},
    # textured children (RGB variant): a leaf-style cutout -- checkerboard opacity over a bitmap-textured two-sided diffuse child -- and a
    # blend of a checkerboard diffuse with a bitmap-textured roughplastic (Texture::mean() feeds the child's lobe weights)
    "mask_textured_child": {"type": "mask", "opacity": {"type": "checkerboard", "color0": 1.0, "color1": 0.0, "to_uv": [[6.0, 0, 0, 0], [0, 6.0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]},
                            "nested": {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": {"type": "bitmap",
                                       "data": np.random.default_rng(6).uniform(0.1, 0.9, size=(7, 5, 3)).astype(np.float32)}}}},
    "blend_textured_children": {"type": "blendbsdf", "weight": 0.45,
                                "bsdf_0": {"type": "diffuse", "reflectance": {"type": "checkerboard", "color0": [0.8, 0.2, 0.1], "color1": [0.1, 0.3, 0.7]}},
                                "bsdf_1": {"type": "roughplastic", "alpha": 0.2, "diffuse_reflectance": {"type": "bitmap",
                                           "data": np.random.default_rng(7).uniform(0.1, 0.9, size=(4, 6, 3)).astype(np.float32)}}},
}


def _scene(material):
    from mitsuba2_amd import scenes
    cb = scenes.cornell_box()
    cb["bsdfs"] = list(cb["bsdfs"]) + [MATERIALS[material], {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": [0.7, 0.7, 0.7]}}]
    cb["meshes"][6] = dict(cb["meshes"][6], bsdf=len(cb["bsdfs"]) - 2)       # tall box: the material under test
    cb["meshes"][0] = dict(cb["meshes"][0], bsdf=len(cb["bsdfs"]) - 1)       # floor: twosided diffuse
    return cb


@pytest.mark.parametrize("material", sorted(MATERIALS))
@pytest.mark.parametrize("pipeline", [0, 1, 2, 4])       # 0: automatic (a pass this small: one launch of persistent lanes); 4: the in-kernel shadow ring, the default of LDS-resident scenes
def test_sample_radiance_matches_oracle(material, pipeline):
    from mitsuba2_amd import render as R, scenes
    cb, sp = _scene(material), scenes.cornell_box_sensor(64, 64, spp=8, seed=21)
    sp["max_depth"] = 6
    scene, sensor = R.Scene(cb), R.make_sensor(sp)
    integ = R.PathIntegrator(max_depth=6, pipeline=pipeline)
    n = 64 * 64 * 8
    rgb, mask, pos = integ.sample(scene, sensor, 0, n)
    rgb, mask, pos = rgb.cpu().numpy(), mask.cpu().numpy(), pos.cpu().numpy()
    oracle = ob.OracleScene(cb)
    want, wpos = oracle.sample_radiance(ob.make_desc(sp), 0, n)
    assert np.array_equal(pos, wpos) and np.array_equal(mask, want[:, 3] > 0.5)
    # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
    # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
    parity_util.check("per-sample radiance", rgb, want[:, :3])
    assert abs(rgb.mean() - want[:, :3].mean()) < 2e-2 * max(want[:, :3].mean(), 1e-3)


@pytest.mark.parametrize("material", ["rough_ggx", "glass", "plastic", "frosted_glass"])
def test_film_relmse(material):
    from mitsuba2_amd import render as R, scenes
    cb, sp = _scene(material), scenes.cornell_box_sensor(48, 48, spp=32, seed=4)
    scene, sensor = R.Scene(cb), R.make_sensor(sp)
    assert R.PathIntegrator(max_depth=8).render(scene, sensor)
    got = sensor.film().bitmap().cpu().numpy()[..., :3]
    sp["max_depth"] = 8
    film, _ = ob.OracleScene(cb).render(ob.make_desc(sp), mode=1)
    want = ob.film_develop(film)[..., :3]
    relmse = np.mean((got - want) ** 2 / (want ** 2 + 1e-2))
    assert relmse < 1e-3, relmse


def test_unsupported_combinations():
    from mitsuba2_amd import render as R, scenes
    cb = _scene("conductor")                      # RGB eta / k cannot be upsampled (values > 1): uniform spectra only
    with pytest.raises(RuntimeError, match="uniform"):
        R.Scene(cb, variant="spectral")
    with pytest.raises(RuntimeError, match="not supported by this backend"):
        R.Scene(dict(cb, bsdfs=[{"type": "measured"}] * len(cb["bsdfs"])))
    # constructor errors of the nesting plugins (blendbsdf.cpp:57-72, mask.cpp:71-82, twosided.cpp:78-80)
    d = {"type": "diffuse"}
    for bad, msg in (({"type": "blendbsdf", "a": d, "b": d}, "weight"), ({"type": "blendbsdf", "weight": 0.5, "a": d}, "Two child BSDFs"),
                     ({"type": "blendbsdf", "weight": 0.5, "a": d, "b": d, "c": d}, "more than two"), ({"type": "mask"}, "Child BSDF not specified"),
                     ({"type": "mask", "a": d, "b": d}, "more than one"), ({"type": "twosided", "bsdf": {"type": "mask", "a": d}}, "transmission"),
                     ({"type": "mask", "a": {"type": "mask", "a": d}}, "one level")):
        with pytest.raises(RuntimeError, match=msg):
            R.Scene(dict(cb, bsdfs=[bad] * len(cb["bsdfs"])))
    with pytest.raises(RuntimeError, match="RGB variant only"):      # textured children of a nest: RGB variant
        R.Scene(dict(cb, bsdfs=[{"type": "mask", "a": {"type": "diffuse", "reflectance": {"type": "checkerboard"}}}] * len(cb["bsdfs"])), variant="spectral")
    with pytest.raises(RuntimeError, match="eval_1"):      # a textured weight is converted into spectra in the spectral variant
        R.Scene(dict(cb, bsdfs=[{"type": "blendbsdf", "weight": {"type": "checkerboard"}, "a": d, "b": d}] * len(cb["bsdfs"])), variant="spectral")
    with pytest.raises(RuntimeError, match="positive and differ"):
        R.Scene(dict(cb, bsdfs=[{"type": "roughdielectric", "int_ior": 1.2, "ext_ior": 1.2}] * len(cb["bsdfs"])))


def test_roughplastic_tables_match_oracle():
    """the transmittance table / internal reflectance integrated on the device against the oracle's (roughplastic.cpp:380-399)"""
    import ctypes as C
    from mitsuba2_amd import render as R, scenes, _lib as L
    for mat in ({"type": "roughplastic", "alpha": 0.15, "int_ior": 1.9}, {"type": "roughplastic", "alpha": 0.4, "distribution": "ggx"}):
        cb = scenes.cornell_box()
        cb["bsdfs"] = [mat] + list(cb["bsdfs"][1:])
        scene = R.Scene(cb)
        got = np.zeros(65, np.float32)
        L.check(L.lib().mtsamd_scene_roughplastic_tables(scene._handle, 0, got.ctypes.data_as(L.f32p)))
        want, r_int = ob.roughplastic_tables(mat)
        assert np.allclose(got[:64], want, rtol=1e-4, atol=1e-5) and abs(got[64] - r_int) < 1e-4


@pytest.mark.parametrize("kind", ["checkerboard", "bitmap_to_uv"])
def test_procedural_and_transformed_textures(kind):
    """src/textures/checkerboard.cpp:46-63 and the `to_uv` transform of textures (bitmap.cpp:62,254): per-sample parity"""
    from mitsuba2_amd import render as R, scenes, xml as mxml
    tex = np.random.default_rng(2).uniform(0.1, 0.9, size=(8, 8, 3)).astype(np.float32)
    cb = scenes.cornell_box(texture=tex)
    to_uv = mxml.scale([3.0, 5.0, 1.0])
    to_uv[0, 2], to_uv[1, 2] = 0.25, 0.1                     # the third column of the 4x4 is the translation after extract()
    for b in cb["bsdfs"]:
        if isinstance(b.get("reflectance"), dict):
            if kind == "checkerboard":
                b["reflectance"] = {"type": "checkerboard", "color0": [0.8, 0.2, 0.1], "color1": 0.3, "to_uv": to_uv}
            else:
                b["reflectance"] = dict(b["reflectance"], to_uv=to_uv)
    sp = dict(scenes.cornell_box_sensor(64, 64, spp=4, seed=9), max_depth=4)
    scene, sensor = R.Scene(cb), R.make_sensor(sp)
    n = 64 * 64 * 4
    rgb, mask, pos = R.PathIntegrator(max_depth=4).sample(scene, sensor, 0, n)
    want, wpos = ob.OracleScene(cb).sample_radiance(ob.make_desc(sp), 0, n)
    assert np.array_equal(pos.cpu().numpy(), wpos)
    # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
    # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
    parity_util.check("per-sample radiance", rgb.cpu().numpy(), want[:, :3])
    plain, _, _ = R.PathIntegrator(max_depth=4).sample(R.Scene(scenes.cornell_box(texture=tex)), sensor, 0, n)
    assert not torch.equal(plain, rgb)                       # the transform / pattern really changes the picture
