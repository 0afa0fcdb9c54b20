"""Scene files and film output on the GPU: a scene loaded from XML + OBJ files renders
exactly what the same scene passed as buffers renders, and HDRFilm.develop() writes files that read back to the film's
contents (src/films/tests/test_hdrfilm.py:74-160)."""
import os

import numpy as np
import parity_util
import pytest
import torch

pytestmark = pytest.mark.gpu

XML = """<scene version="2.0.0">
    <integrator type="path"><integer name="max_depth" value="$depth"/></integrator>
    <sensor type="perspective">
        <float name="near_clip" value="{near}"/> <float name="far_clip" value="{far}"/>
        <float name="fov" value="{fov}"/>
        <transform name="to_world"><matrix value="{mat}"/></transform>
        <sampler type="independent"><integer name="sample_count" value="$spp"/><integer name="seed" value="{seed}"/></sampler>
        <film type="hdrfilm"><integer name="width" value="{w}"/><integer name="height" value="{h}"/>
            <string name="file_format" value="pfm"/><rfilter type="gaussian"/></film>
    </sensor>
{bsdfs}
{shapes}
</scene>"""


def _write_scene(tmp, cb, sp):
    bsdfs = "\n".join('<bsdf type="diffuse" id="b%d"><rgb name="reflectance" value="%.9g, %.9g, %.9g"/></bsdf>' % ((i,) + tuple(b["reflectance"]))
                      for i, b in enumerate(cb["bsdfs"]))
    shapes = []
    for i, m in enumerate(cb["meshes"]):
        pos, faces = np.asarray(m["positions"], np.float32).reshape(-1, 3), np.asarray(m["faces"]).reshape(-1, 3)
        with open(os.path.join(tmp, "m%d.obj" % i), "w") as fh:
            for p in pos:
                fh.write("v %.9g %.9g %.9g\n" % tuple(p))
            for f in faces:
                fh.write("f %d %d %d\n" % tuple(f + 1))
        em = ""
        if m.get("emitter", -1) >= 0:
            em = '<emitter type="area"><rgb name="radiance" value="%.9g, %.9g, %.9g"/></emitter>' % tuple(cb["emitters"][m["emitter"]]["radiance"])
        shapes.append('<shape type="obj"><string name="filename" value="m%d.obj"/><boolean name="face_normals" value="true"/>'
                      '<ref id="b%d"/>%s</shape>' % (i, m["bsdf"], em))
    path = os.path.join(tmp, "scene.xml")
    with open(path, "w") as fh:
        fh.write(XML.format(near=sp["near_clip"], far=sp["far_clip"], fov="%.9g" % sp["fov"], seed=sp["seed"], w=sp["width"], h=sp["height"],
                            mat=" ".join("%.9g" % x for x in np.asarray(sp["to_world"], np.float32).reshape(-1)),
                            bsdfs=bsdfs, shapes="\n".join(shapes)))
    return path


def test_xml_scene_renders_like_buffers(tmp_path):
    from mitsuba2_amd import bitmap, render as R, scenes, xml as mxml
    cb = scenes.cornell_box()
    sp = scenes.cornell_box_sensor(48, 48, spp=8, seed=5)
    path = _write_scene(str(tmp_path), cb, sp)
    scene = mxml.load_file(path, depth=5, spp=8)
    assert scene.integrator().max_depth == 5 and scene.shape_count() == len(cb["meshes"])
    assert scene.integrator().render(scene)
    got = scene.sensors()[0].film().bitmap(raw=True).cpu().numpy()

    ref_scene = R.Scene(cb)
    sensor = R.make_sensor(sp)
    assert R.PathIntegrator(max_depth=5).render(ref_scene, sensor)
    want = sensor.film().bitmap(raw=True).cpu().numpy()
    assert np.array_equal(got, want)

    film = scene.sensors()[0].film()
    with pytest.raises(RuntimeError, match="Destination file not specified"):
        film.develop()
    film.set_destination_file(str(tmp_path / "out.exr"))
    out = film.develop()
    assert out.endswith("out.pfm")
    assert np.array_equal(bitmap.read_pfm(out), film.bitmap().cpu().numpy()[..., :3])


@pytest.mark.parametrize("file_format", ["exr", "rgbe", "pfm"])
def test_hdrfilm_develop(tmp_path, file_format):
    """test_hdrfilm.py:74-160 (test03_develop)"""
    from mitsuba2_amd import bitmap, render as R
    rng = np.random.default_rng(12345 + ord(file_format[0]))
    film = R.HDRFilm(41, 37, rfilter=R.BoxFilter(), file_format=file_format, pixel_format="xyza" if file_format == "exr" else "rgba",
                     component_format="float32")
    contents = rng.uniform(size=(37, 41, 5)).astype(np.float32)
    if file_format == "rgbe":
        contents = (1 + 0.1 * contents).astype(np.float32)
    contents[:, :, 4] = 1.0
    block = R.ImageBlock(film.size(), 5, film.reconstruction_filter())
    block.clear()
    yy, xx = np.mgrid[0:37, 0:41]
    pos = torch.as_tensor(np.stack([xx.reshape(-1) + 0.5, yy.reshape(-1) + 0.5], 1).astype(np.float32), device="cuda")
    block.put(pos, torch.as_tensor(contents.reshape(-1, 5), device="cuda"))
    film.prepare(["X", "Y", "Z", "A", "W"])
    film.put(block)
    film.set_destination_file(str(tmp_path / ("test_image." + file_format)))
    out = film.develop()
    if file_format == "exr":
        ch, _ = bitmap.read_exr(out)
        img = np.stack([ch[k] for k in "XYZA"], 2)
        assert np.allclose(img, contents[..., :4], atol=1e-5)
    else:
        rgb = film.bitmap().cpu().numpy()[..., :3]
        img = bitmap.read_pfm(out) if file_format == "pfm" else bitmap.read_rgbe(out)
        assert np.allclose(img, rgb, atol=1e-2 if file_format == "rgbe" else 1e-6)


def test_film_parameter_validation():
    """test_hdrfilm.py:22-32"""
    from mitsuba2_amd import render as R
    with pytest.raises(RuntimeError):
        R.HDRFilm(component_format="uint8")
    with pytest.raises(RuntimeError):
        R.HDRFilm(pixel_format="brga")


MATPREVIEW_XML = """<scene version="2.0.0">
    <default name="spp" value="8"/>
    <integrator type="path"><integer name="max_depth" value="6"/></integrator>
    <sensor type="perspective">
        <float name="fov" value="35"/>
        <transform name="to_world"><lookat origin="0, 1.5, 4.5" target="0, 0.2, 0" up="0, 1, 0"/></transform>
        <sampler type="independent"><integer name="sample_count" value="$spp"/><integer name="seed" value="3"/></sampler>
        <film type="hdrfilm"><integer name="width" value="64"/><integer name="height" value="48"/><rfilter type="gaussian"/></film>
    </sensor>
    <bsdf type="roughplastic" id="object">
        <float name="alpha" value="0.2"/><string name="distribution" value="ggx"/>
        <rgb name="diffuse_reflectance" value="0.3, 0.1, 0.05"/><float name="int_ior" value="1.6"/>
    </bsdf>
    <shape type="serialized"><string name="filename" value="ball.serialized"/><ref id="object"/></shape>
    <shape type="rectangle">
        <transform name="to_world"><rotate x="1" angle="-90"/><scale value="4"/><translate y="-1.05"/></transform>
        <bsdf type="twosided"><bsdf type="diffuse">
            <texture type="checkerboard" name="reflectance"><rgb name="color0" value="0.6, 0.6, 0.6"/><spectrum name="color1" value="0.15"/>
                <transform name="to_uv"><scale x="6" y="6"/></transform></texture>
        </bsdf></bsdf>
    </shape>
    <emitter type="envmap"><string name="filename" value="sky.pfm"/><float name="scale" value="1.5"/>
        <transform name="to_world"><rotate y="1" angle="40"/></transform></emitter>
</scene>"""


def test_matpreview_style_scene_from_files(tmp_path):
    """a material-preview style scene assembled from files -- serialized mesh, roughplastic, checkerboard ground, envmap from a
    PFM, XML with defaults -- renders like the oracle fed with the same parsed description"""
    import oracle_binding as ob
    from mitsuba2_amd import bitmap, loaders, scenes, xml as mxml
    ball = scenes.bumpy_sphere(24, 48)["meshes"][0]
    loaders.write_serialized(str(tmp_path / "ball.serialized"), [dict(positions=ball["positions"], faces=ball["faces"], normals=ball["normals"])])
    yy, xx = np.mgrid[0:32, 0:64]
    sky = np.stack([0.3 + 0.5 * (yy < 14), 0.4 + 0.4 * (yy < 14), 0.9 - 0.01 * yy], 2).astype(np.float32)
    sky[6:9, 40:44] += 25.0
    bitmap.write_pfm(str(tmp_path / "sky.pfm"), sky)
    with open(str(tmp_path / "scene.xml"), "w") as fh:
        fh.write(MATPREVIEW_XML)
    desc = mxml.parse_file(str(tmp_path / "scene.xml"), spp=16)
    scene = mxml.instantiate(desc)
    sensor = scene.sensors()[0]
    assert sensor.sampler().sample_count() == 16 and scene.integrator().max_depth == 6
    n = 64 * 48 * 16
    rgb, mask, pos = scene.integrator().sample(scene, sensor, 0, n)
    sd = desc.sensors[0]
    op = dict(to_world=sd["to_world"], fov=sensor.x_fov(), near_clip=sd["near_clip"], far_clip=sd["far_clip"], width=64, height=48,
              crop=(0, 0, 64, 48), rfilter="gaussian", rfilter_param=0.5, sample_count=16, seed=3, max_depth=6, rr_depth=5)
    want, wpos = ob.OracleScene(desc.scene_dict).sample_radiance(ob.make_desc(op), 0, n)
    assert np.array_equal(pos.cpu().numpy(), wpos) and np.array_equal(mask.cpu().numpy(), want[:, 3] > 0.5)
    # every operation of the path is shared bit for bit with the oracle (round 3: explicit elementary functions on both sides):
    # all samples identical -- rounds 1-2 accepted 99-99.9 % "close" here
    parity_util.check("per-sample radiance", rgb.cpu().numpy(), want[:, :3])
    assert scene.integrator().render(scene, sensor)
    img = sensor.film().bitmap().cpu().numpy()
    assert np.isfinite(img).all() and img[..., :3].mean() > 0.05 and img[..., 3].min() >= 0


def test_xml_thinlens_sensor():
    """`thinlens` sensor from XML: aperture_radius is mandatory (thinlens.cpp:112), focus_distance defaults to far_clip"""
    from mitsuba2_amd import xml as mxml
    x = """<scene version="2.0.0"><sensor type="thinlens"><float name="aperture_radius" value="0.2"/>%s
           <float name="far_clip" value="50"/><film type="hdrfilm"><integer name="width" value="16"/><integer name="height" value="8"/>
           <rfilter type="mitchell"><float name="B" value="0.2"/></rfilter></film></sensor>
           <shape type="rectangle"><emitter type="area"><rgb name="radiance" value="1,1,1"/></emitter>
           <transform name="to_world"><rotate y="1" angle="180"/><translate z="4"/></transform></shape></scene>"""
    scene = mxml.load_string(x % '<float name="focus_distance" value="4"/>')
    cam = scene.sensors()[0]
    assert cam.needs_aperture_sample() and abs(cam.aperture_radius() - 0.2) < 1e-7 and cam.focus_distance() == 4.0
    assert cam.film().reconstruction_filter().kind == 4 and abs(cam.film().reconstruction_filter().param - 0.2) < 1e-7
    assert mxml.load_string(x % "").sensors()[0].focus_distance() == 50.0
    with pytest.raises(Exception, match="aperture_radius"):
        mxml.load_string(x.replace('<float name="aperture_radius" value="0.2"/>', "") % "")
    assert scene.integrator() is None
    from mitsuba2_amd import render as R
    assert R.PathIntegrator(max_depth=2).render(scene, cam)
    img = cam.film().bitmap().cpu().numpy()
    assert np.isfinite(img).all() and img[..., :3].max() > 0
