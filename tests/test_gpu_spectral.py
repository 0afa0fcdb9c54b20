"""Spectral variant on the GPU (BASELINE config 3 semantics): k_bounce_spectral against the oracle's spectral path on
identical sampler seeds and the same coefficient table."""
import numpy as np
import parity_util
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
    # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
    # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
    parity_util.check("per-sample radiance", xyz.cpu().numpy(), ref_rgba[:, :3])


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
        gpu.Scene(sd, variant="polarized")


SPECTRAL_MATERIALS = {
    "uniform_diffuse": {"type": "diffuse", "reflectance": 0.4},
    "twosided_diffuse": {"type": "twosided", "bsdf": {"type": "diffuse", "reflectance": [0.6, 0.3, 0.2]}},
    "conductor": {"type": "conductor", "eta": 0.2, "k": 3.9, "specular_reflectance": [0.9, 0.7, 0.3]},
    "rough_ggx": {"type": "roughconductor", "alpha": 0.2, "distribution": "ggx", "eta": 0.2, "k": 3.9},
    "rough_beckmann": {"type": "roughconductor", "alpha_u": 0.3, "alpha_v": 0.1, "distribution": "beckmann", "eta": 0.0, "k": 1.0,
                       "specular_reflectance": 0.8},
    "glass": {"type": "dielectric", "int_ior": "bk7", "specular_transmittance": [0.9, 0.95, 1.0]},
    "thin_glass": {"type": "thindielectric", "specular_transmittance": [0.9, 0.95, 1.0], "specular_reflectance": 0.8},
    "frosted_glass": {"type": "roughdielectric", "alpha": 0.2, "specular_transmittance": [0.9, 0.95, 1.0], "specular_reflectance": 0.9},
    "plastic": {"type": "plastic", "diffuse_reflectance": [0.1, 0.27, 0.36], "int_ior": 1.9},
    "plastic_uniform": {"type": "plastic", "diffuse_reflectance": 0.3, "specular_reflectance": 0.7, "nonlinear": True},
    "blend": {"type": "blendbsdf", "weight": 0.35, "bsdf_0": {"type": "diffuse", "reflectance": [0.6, 0.3, 0.2]},
              "bsdf_1": {"type": "roughconductor", "alpha": 0.2, "eta": 0.2, "k": 3.9, "specular_reflectance": [0.9, 0.7, 0.3]}},
    "mask": {"type": "mask", "opacity": 0.6, "nested": {"type": "plastic", "diffuse_reflectance": [0.1, 0.27, 0.36]}},
}


@pytest.mark.parametrize("material", sorted(SPECTRAL_MATERIALS))
def test_spectral_materials_match_oracle(gpu, oracle, material):
    """the BSDF models of row f-2 in the spectral variant: `srgb` parameters upsampled per wavelength, `uniform` ones constant
    (src/spectra/srgb.cpp:45-52, uniform.cpp), conductors with uniform eta / k; fused and split pipeline"""
    path = gpu.srgb_coeff_path()
    cb = scenes.cornell_box()
    cb["bsdfs"] = list(cb["bsdfs"]) + [SPECTRAL_MATERIALS[material]]
    cb["meshes"][6] = dict(cb["meshes"][6], bsdf=len(cb["bsdfs"]) - 1)
    p = dict(scenes.cornell_box_sensor(48, 48, 8, seed=12), max_depth=6)
    scene, sensor = gpu.Scene(cb, variant="spectral"), gpu.make_sensor(p)
    n = 48 * 48 * 8
    xyz, mask, pos = gpu.PathIntegrator(max_depth=6, pipeline=1).sample(scene, sensor, 0, n)
    xyz2, _, _ = gpu.PathIntegrator(max_depth=6, pipeline=2).sample(scene, sensor, 0, n)
    assert (xyz == xyz2).all()
    xyz0, _, _ = gpu.PathIntegrator(max_depth=6).sample(scene, sensor, 0, n)          # default schedule: in-kernel shadow ring
    assert (xyz == xyz0).all()
    ref, ref_pos = oracle.OracleScene(cb, spectral_path=path).sample_radiance(oracle.make_desc(p), 0, n)
    assert (pos.cpu().numpy() == ref_pos).all() and ((ref[:, 3] > 0.5) == mask.cpu().numpy()).all()
    # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
    # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
    parity_util.check("per-sample radiance", xyz.cpu().numpy(), ref[:, :3])


def test_spectral_conductor_needs_uniform_ior(gpu):
    cb = scenes.cornell_box()
    cb["bsdfs"] = list(cb["bsdfs"]) + [{"type": "conductor", "eta": [0.2, 0.9, 1.1], "k": [3.9, 2.4, 2.1]}]
    cb["meshes"][6] = dict(cb["meshes"][6], bsdf=len(cb["bsdfs"]) - 1)
    with pytest.raises(RuntimeError, match="uniform"):
        gpu.Scene(cb, variant="spectral")


@pytest.mark.parametrize("with_area", [False, True])
@pytest.mark.parametrize("kind", ["constant", "envmap"])
def test_spectral_environment_emitters(gpu, oracle, kind, with_area):
    """`constant` (radiance upsampled like any emitter colour, srgb_d65.cpp) and `envmap` (per-texel model coefficients + scale,
    D65 whitepoint: envmap.cpp:96-109, :283-306) in the spectral variant, alone and next to an area light; fused == split"""
    from test_gpu_integrators import _open_scene, _envmap_image
    path = gpu.srgb_coeff_path()
    cb = _open_scene(with_area)
    if kind == "envmap":
        rot = scenes.look_at([0, 0, 0], [1, 0.2, 0.3], [0, 1, 0])
        env = {"type": "envmap", "data": _envmap_image(), "scale": 0.7, "to_world": rot}
        cb["emitters"] = [env] + [e for e in cb["emitters"] if e.get("type", "area") == "area"]
    p = dict(scenes.cornell_box_sensor(64, 64, 4, seed=21), max_depth=5)
    scene, sensor = gpu.Scene(cb, variant="spectral"), gpu.make_sensor(p)
    n = 64 * 64 * 4
    xyz, mask, pos = gpu.PathIntegrator(max_depth=5, pipeline=1).sample(scene, sensor, 0, n)
    xyz2, _, _ = gpu.PathIntegrator(max_depth=5, pipeline=2).sample(scene, sensor, 0, n)
    assert (xyz == xyz2).all()
    ref, ref_pos = oracle.OracleScene(cb, spectral_path=path).sample_radiance(oracle.make_desc(p), 0, n)
    m = mask.cpu().numpy()
    assert (pos.cpu().numpy() == ref_pos).all() and ((ref[:, 3] > 0.5) == m).all() and (~m).any()
    got = xyz.cpu().numpy()
    assert np.isfinite(got).all() and (got[~m].sum(1) > 0).all()          # escaped camera rays see the environment
    # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
    # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
    parity_util.check("per-sample radiance", got, ref[:, :3])
    assert abs(got.mean() - ref[:, :3].mean()) < 0.02 * ref[:, :3].mean()
    # the spectral rendering agrees with the RGB one up to the upsampling model (film: XYZ -> RGB on both sides)
    out = {}
    for variant in ("rgb", "spectral"):
        sc, se = gpu.Scene(cb, variant=variant), gpu.make_sensor(dict(p, sample_count=32))
        assert gpu.PathIntegrator(max_depth=5).render(sc, se)
        out[variant] = se.film().bitmap().cpu().numpy()[..., :3]
    assert abs(out["rgb"].mean() - out["spectral"].mean()) < 0.1 * out["rgb"].mean()


@pytest.mark.parametrize("kind", ["bitmap", "checkerboard", "plastic_bitmap", "roughplastic_checkerboard"])
def test_spectral_textures_match_oracle(gpu, oracle, kind):
    """textures in the spectral variant: bitmap texels become model coefficients that are evaluated at the four corners and
    interpolated (bitmap.cpp:116-123, :274-286), checkerboard colours are `srgb` spectra; Texture::mean() of either feeds the
    plastic lobe weights (plastic.cpp:170-175)"""
    from mitsuba2_amd import xml as mxml
    path = gpu.srgb_coeff_path()
    tex = np.random.default_rng(6).uniform(0.0, 1.0, size=(8, 8, 3)).astype(np.float32)
    tex[2, 3] = 0.0; tex[5, 1] = 1.0                        # black / white texels: infinite c2 (srgb.cpp:31-36)
    cb = scenes.cornell_box(texture=tex)
    to_uv = mxml.scale([3.0, 2.0, 1.0])
    for i, b in enumerate(cb["bsdfs"]):
        if isinstance(b.get("reflectance"), dict):
            spec = dict(b["reflectance"], to_uv=to_uv)
            if kind.endswith("checkerboard"):
                spec = {"type": "checkerboard", "color0": [0.8, 0.2, 0.1], "color1": [0.3, 0.3, 0.9], "to_uv": to_uv}
            if kind.startswith("plastic"):
                cb["bsdfs"][i] = {"type": "plastic", "diffuse_reflectance": spec, "specular_reflectance": [0.9, 0.8, 0.7]}
            elif kind.startswith("roughplastic"):
                cb["bsdfs"][i] = {"type": "roughplastic", "diffuse_reflectance": spec, "alpha": 0.15, "distribution": "ggx"}
            else:
                cb["bsdfs"][i] = dict(b, reflectance=spec)
    p = dict(scenes.cornell_box_sensor(64, 64, 4, seed=14), max_depth=5)
    scene, sensor = gpu.Scene(cb, variant="spectral"), gpu.make_sensor(p)
    n = 64 * 64 * 4
    xyz, mask, pos = gpu.PathIntegrator(max_depth=5, pipeline=1).sample(scene, sensor, 0, n)
    xyz2, _, _ = gpu.PathIntegrator(max_depth=5, pipeline=2).sample(scene, sensor, 0, n)
    assert (xyz == xyz2).all()
    ref, ref_pos = oracle.OracleScene(cb, spectral_path=path).sample_radiance(oracle.make_desc(p), 0, n)
    assert (pos.cpu().numpy() == ref_pos).all()
    got = xyz.cpu().numpy()
    assert np.isfinite(got).all()
    # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
    # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
    parity_util.check("per-sample radiance", got, ref[:, :3])
    # and the RGB variant of the same scene (textured plastic weights from the texture's mean luminance)
    rgb, _, _ = gpu.PathIntegrator(max_depth=5).sample(gpu.Scene(cb), sensor, 0, n)
    want, _ = oracle.OracleScene(cb).sample_radiance(oracle.make_desc(p), 0, n)
    # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
    # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
    parity_util.check("per-sample radiance", rgb.cpu().numpy(), want[:, :3])


def test_spectral_parameter_updates_equal_a_fresh_scene(gpu):
    """parameters_changed() in the spectral variant: colours set on a live scene go through the same srgb -> coefficient conversion as at
    scene creation (round 3; until then the setters wrote the RGB fields the spectral kernels never read).  The film after the updates
    is the film of a scene built with the new values, bit for bit."""
    import copy
    sd = scenes.cornell_box()
    sd["bsdfs"].append({"type": "plastic", "diffuse_reflectance": np.array([0.1, 0.27, 0.36], np.float32), "specular_reflectance": np.array([0.9, 0.8, 0.7], np.float32)})
    plastic = len(sd["bsdfs"]) - 1
    sd["meshes"][6]["bsdf"] = plastic                       # one of the boxes
    p = scenes.cornell_box_sensor(48, 40, 8, seed=4)
    scene, sensor = gpu.Scene(sd, variant="spectral"), gpu.make_sensor(p)
    integ = gpu.PathIntegrator()
    assert integ.render(scene, sensor)
    before = sensor.film().bitmap(raw=True).clone()
    new = copy.deepcopy(sd)
    new["bsdfs"][1]["reflectance"] = np.array([0.2, 0.5, 0.3], np.float32)
    new["bsdfs"][plastic]["diffuse_reflectance"] = np.array([0.3, 0.1, 0.05], np.float32)
    new["bsdfs"][plastic]["specular_reflectance"] = np.array([0.5, 0.6, 0.9], np.float32)
    new["emitters"][0]["radiance"] = np.array([10.0, 12.0, 15.0], np.float32)
    scene.set_bsdf_reflectance(1, new["bsdfs"][1]["reflectance"])
    scene.set_bsdf_param(plastic, 0, new["bsdfs"][plastic]["diffuse_reflectance"])
    scene.set_bsdf_param(plastic, 1, new["bsdfs"][plastic]["specular_reflectance"])
    scene.set_emitter_radiance(0, new["emitters"][0]["radiance"])
    assert integ.render(scene, sensor)
    after = sensor.film().bitmap(raw=True).clone()
    assert not np.allclose(before.cpu().numpy(), after.cpu().numpy(), rtol=1e-2)
    fresh, sensor2 = gpu.Scene(new, variant="spectral"), gpu.make_sensor(p)
    assert gpu.PathIntegrator().render(fresh, sensor2)
    assert (sensor2.film().bitmap(raw=True) == after).all()
    with pytest.raises(RuntimeError, match="Invalid RGB reflectance"):
        scene.set_bsdf_reflectance(1, [1.2, 0.1, 0.1])
    uni = scenes.cornell_box()
    uni["bsdfs"][0] = {"type": "diffuse", "reflectance": 0.4}             # a `uniform` spectrum: not an srgb colour
    with pytest.raises(RuntimeError, match="uniform spectrum"):
        gpu.Scene(uni, variant="spectral").set_bsdf_reflectance(0, [0.3, 0.3, 0.3])
