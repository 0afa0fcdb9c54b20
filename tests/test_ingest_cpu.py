"""Scene ingestion and image IO, CPU only: mesh loaders and normal generation against
the reference's own fixtures and expected values (src/librender/tests/test_mesh.py:35-107, data/triangle*.ply copied
to tests/golden/ as data), the XML subset's error behaviour (src/libcore/tests/test_xml.py) and file-format round trips.
The EXR writer has no reference-produced file to compare with (the reference ships none): its byte layout is
"parity unpinned" and covered by a write -> read round trip only."""
import math
import os

import numpy as np
import pytest

from mitsuba2_amd import bitmap, loaders, scenes
from mitsuba2_amd import xml as mxml

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ------------------------------------------------------------------------------------------------ meshes
def test_ply_triangle():
    """test_mesh.py:35-57 (test02_ply_triangle)"""
    d = mxml.parse_string("""<scene version="2.0.0"><shape type="ply">
            <string name="filename" value="triangle.ply"/>
            <boolean name="face_normals" value="true"/>
        </shape></scene>""", base_dir=GOLDEN)
    m = d.scene_dict["meshes"][0]
    assert m["normals"] is None
    assert m["positions"].size == 9
    assert np.allclose(m["positions"], [[0, 0, 0], [0, 0, 1], [0, 1, 0]])
    assert m["faces"].tolist() == [[0, 1, 2]]


def test_ply_computed_normals():
    """test_mesh.py:60-77 (test03_ply_computed_normals)"""
    m = loaders.load_ply(os.path.join(GOLDEN, "triangle.ply"))
    assert m["normals"] is not None
    assert np.allclose(m["normals"], [[-1, 0, 0]] * 3)


def test_ply_binary_with_face_attributes():
    """binary_little_endian file with normals and per-face float attributes (data/triangle_face_colors.ply)"""
    m = loaders.load_ply(os.path.join(GOLDEN, "triangle_face_colors.ply"))
    assert np.allclose(m["positions"], [[0, 0, 0], [0, 0, 1], [0, 1, 0]])
    assert np.allclose(m["normals"], [[-1, 0, 0]] * 3)
    assert m["faces"].tolist() == [[0, 1, 2]]


def test_normal_weighting_scheme():
    """test_mesh.py:80-107 (test04_normal_weighting_scheme)"""
    a, b = 1.0, 0.5
    vertices = np.array([0, 0, 0, -a, 1, 0, a, 1, 0, -b, 0, 1, b, 0, 1], dtype=np.float32).reshape(5, 3)
    n0, n1 = np.array([0.0, 0.0, -1.0]), np.array([0.0, 1.0, 0.0])
    n2 = n0 * (math.pi / 2.0) + n1 * math.acos(3.0 / 5.0)
    n2 /= np.linalg.norm(n2)
    expected = np.vstack([n2, n0, n0, n1, n1])
    got = loaders.compute_vertex_normals(vertices, [[0, 1, 2], [0, 3, 4]])
    assert np.allclose(got, expected, atol=5e-4)


def test_normals_unreferenced_vertex_is_bogus():
    """mesh.cpp:243-249: a vertex without a valid normal gets (1, 0, 0)"""
    got = loaders.compute_vertex_normals([[0, 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5]], [[0, 1, 2]])
    assert np.allclose(got[3], [1, 0, 0]) and np.allclose(got[:3], [[0, 0, 1]] * 3)


def _write_box_obj(path, with_normals, with_uv):
    """the Cornell box's small box written the way the reference's cbox_smallbox.obj is: 24 vertices, 6 quads"""
    cb = scenes.cornell_box()
    m = cb["meshes"][7]
    pos, faces = np.asarray(m["positions"]).reshape(-1, 3), np.asarray(m["faces"]).reshape(-1, 3)
    with open(path, "w") as fh:
        fh.write("# test\n")
        for p in pos:
            fh.write("v %g %g %g\n" % tuple(p))
        if with_uv:
            for k in range(len(pos)):
                fh.write("vt %g %g\n" % (0.25 * (k % 4), 0.125 * (k // 4)))
        if with_normals:
            for f in faces:
                n = np.cross(pos[f[1]] - pos[f[0]], pos[f[2]] - pos[f[0]])
                fh.write("vn %g %g %g\n" % tuple(n / np.linalg.norm(n)))
        for i, f in enumerate(faces):
            def key(v):
                return "%d%s%s" % (v + 1, ("/%d" % (v + 1)) if with_uv else ("/" if with_normals else ""), ("/%d" % (i + 1)) if with_normals else "")
            fh.write("f %s %s %s\n" % tuple(key(v) for v in f))
    return pos, faces


@pytest.mark.parametrize("features", ["none", "normals", "uv", "normals_uv"])
@pytest.mark.parametrize("face_normals", [True, False])
def test_obj_features(tmp_path, features, face_normals):
    """test_mesh.py:137-170 (test06_load_various_features): the loader honours vn / vt / face_normals combinations"""
    path = str(tmp_path / "box.obj")
    pos, faces = _write_box_obj(path, "normals" in features, "uv" in features)
    m = loaders.load_obj(path, face_normals=face_normals)
    assert m["faces"].shape == faces.shape
    tri = m["positions"][m["faces"]]
    assert np.allclose(tri, pos[faces])                              # same triangles, whatever the vertex split
    assert (m["normals"] is None) == face_normals
    assert (m["texcoords"] is not None) == ("uv" in features)
    if not face_normals:
        fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        fn /= np.linalg.norm(fn, axis=1, keepdims=True)
        if "normals" in features:                                     # per-face vn: every corner carries the face normal
            assert np.allclose(m["normals"][m["faces"]], np.repeat(fn[:, None], 3, 1), atol=1e-5)
        else:
            assert np.allclose(np.linalg.norm(m["normals"], axis=1), 1, atol=1e-5)
    if "uv" in features:                                              # flip_tex_coords default: v -> 1 - v (obj.cpp:188-195)
        uv = m["texcoords"][m["faces"]]
        assert np.allclose(uv[..., 0], 0.25 * (faces % 4)) and np.allclose(uv[..., 1], 1 - 0.125 * (faces // 4))


def test_obj_quads_and_to_world(tmp_path):
    """obj.cpp:255-264 fan triangulation; obj.cpp:176-178 to_world baked into positions"""
    path = str(tmp_path / "quad.obj")
    with open(path, "w") as fh:
        fh.write("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv -0.5 0.5 0\nf 1 2 3 4 5\n")
    m = loaders.load_obj(path, to_world=mxml.translate([1, 2, 3]) @ mxml.scale([2, 2, 2]))
    assert m["faces"].tolist() == [[0, 1, 2], [0, 2, 3], [0, 3, 4]]
    assert np.allclose(m["positions"][2], [3, 4, 3])
    assert np.allclose(m["normals"], [[0, 0, 1]] * 5)
    with open(path, "w") as fh:
        fh.write("v 0 0 0\nv 1 0 0\nf 1 2 7\n")
    with pytest.raises(RuntimeError, match="invalid vertex"):
        loaders.load_obj(path)


# ------------------------------------------------------------------------------------------------ XML
def test_xml_invalid_roots():
    """test_xml.py:7-27"""
    with pytest.raises(Exception):
        mxml.parse_string('<?xml version="1.0"?>')
    with pytest.raises(Exception):
        mxml.parse_string('<?xml version="1.0"?><invalid></invalid>')
    with pytest.raises(Exception, match='root element "integer" must be an object'):
        mxml.parse_string('<?xml version="1.0"?><integer name="a" value="10"></integer>')
    assert mxml.parse_string('<?xml version="1.0"?>\n<scene version="2.0.0"></scene>').scene_dict["meshes"] == []


@pytest.mark.parametrize("body,message", [
    ('<shape type="ply" id="my_id"/><shape type="ply" id="my_id"/>', '"shape" has duplicate id "my_id"'),
    ('<shape type="ply" id="_test"/>', 'invalid id "_test" in element "shape": leading underscores are reserved for internal identifiers.'),
    ('<shape type="ply"><integer name="_test" value="1"/></shape>',
     'invalid parameter name "_test" in element "integer": leading underscores are reserved for internal identifiers.'),
    ('<shape type="ply"><integer name="value" value="1"><shape type="ply"/></integer></shape>', 'node "shape" cannot occur as child of a property'),
    ('<shape type="ply"><integer name="value" value="1"><float name="value" value="1"/></integer></shape>',
     'node "float" cannot occur as child of a property'),
    ('<shape type="ply"><translate name="value" x="0" y="1" z="2"/></shape>', 'transform operations can only occur in a transform node'),
    ('<shape type="ply"><transform name="toWorld"><integer name="value" value="10"/></transform></shape>',
     'transform nodes can only contain transform operations'),
    ('<ref id="unknown"/>', 'reference to unknown object "unknown"'),
    ('<shape type="ply" param2="abc"></shape>', 'unexpected attribute "param2" in element "shape".'),
    ('<integer name="a"/>', 'missing attribute "value" in element "integer".'),
    ('<integer name="a" value="1"/><integer name="a" value="1"/>', 'Property "a" was specified multiple times'),
    ('<shape type="ply"/>', 'Property "filename" has not been specified'),
    ('<shape type="ply"><float name="filename" value="1.0"/></shape>', r'The property "filename" has the wrong type \(expected <string>\).'),
    ('<integer name="n" value="a"/>', 'could not parse integer value "a".'),
    ('<integer name="n" value="1.5"/>', 'could not parse integer value "1.5".'),
    ('<float name="n" value="a"/>', 'could not parse floating point value "a".'),
    ('<shape type="sphere"/>', 'Shape plugin "sphere" is not supported'),
    ('<shape type="rectangle"><bsdf type="hair"/></shape>', "BSDF plugin 'hair' is not supported"),
    ('<shape type="rectangle"><bsdf type="blendbsdf"><bsdf type="diffuse"/><bsdf type="diffuse"/></bsdf></shape>', 'Property "weight" has not been specified'),
    ('<shape type="rectangle"><bsdf type="roughconductor"/></shape>', 'measured IOR tables'),
    ('<shape type="rectangle"><bsdf type="twosided"><bsdf type="dielectric"/></bsdf></shape>', 'Only materials without a transmission component can be nested'),
    ('<shape type="rectangle"><bsdf type="dielectric"><float name="int_ior" value="-0.5"/></bsdf></shape>', 'indices of refraction must be positive'),
    ('<shape type="rectangle"><float name="bogus" value="1"/></shape>', 'unreferenced property "bogus"'),
])
def test_xml_errors(body, message):
    """test_xml.py:40-215 (test05 .. test17): the same messages, minus the line/column prefix"""
    with pytest.raises(Exception, match=message):
        mxml.parse_string('<scene version="2.0.0">%s</scene>' % body)


def test_xml_transform_order_and_lookat():
    """xml.cpp:845-892: each operation is applied on the left of what came before it"""
    d = mxml.parse_string("""<scene version="2.0.0"><shape type="rectangle"><transform name="to_world">
        <scale value="2"/><rotate z="1" angle="90"/><translate x="1" y="2" z="3"/></transform></shape></scene>""")
    p = d.scene_dict["meshes"][0]["positions"]
    assert np.allclose(p[1], [1 + 2, 2 + 2, 3], atol=1e-5)          # (1,-1,0) -> scale (2,-2,0) -> rot z 90 (2,2,0) -> +t
    m = mxml.look_at([0, 0, 5], [0, 0, 0], [0, 1, 0])
    assert np.allclose(m[:3, 2], [0, 0, -1]) and np.allclose(m[:3, 3], [0, 0, 5]) and np.allclose(m[:3, 0], [-1, 0, 0])
    with pytest.raises(Exception, match="invalid lookat transformation"):
        mxml.parse_string('<scene version="2.0.0"><sensor type="perspective"><transform name="to_world">'
                          '<lookat origin="0,0,0" target="0,1,0" up="0,1,0"/></transform></sensor></scene>')


CBOX_XML = """<scene version="2.0.0">
    <default name="spp" value="4"/>
    <default name="res" value="32"/>
    <integrator type="path"><integer name="max_depth" value="$depth"/></integrator>
    <sensor type="perspective">
        <string name="fov_axis" value="smaller"/>
        <float name="near_clip" value="10"/> <float name="far_clip" value="2800"/>
        <float name="fov" value="39.3077"/>
        <transform name="to_world"><lookat origin="278, 273, -800" target="278, 273, -799" up="0, 1, 0"/></transform>
        <sampler type="independent"><integer name="sample_count" value="$spp"/><integer name="seed" value="7"/></sampler>
        <film type="hdrfilm"><integer name="width" value="$res"/><integer name="height" value="$res"/>
            <rfilter type="gaussian"/></film>
    </sensor>
    <bsdf type="diffuse" id="white"><rgb name="reflectance" value="0.885809, 0.698859, 0.666422"/></bsdf>
    <bsdf type="diffuse" id="light"><spectrum name="reflectance" value="0"/></bsdf>
    <alias id="white" as="box"/>
    <shape type="obj" id="floor"><string name="filename" value="floor.obj"/><ref id="box"/></shape>
    <shape type="rectangle" id="lamp">
        <transform name="to_world"><rotate x="1" angle="90"/><scale value="60"/><translate x="278" y="548" z="280"/></transform>
        <ref id="light"/>
        <emitter type="area"><rgb name="radiance" value="18.387, 13.9873, 6.75357"/></emitter>
    </shape>
</scene>"""


def test_xml_scene_subset(tmp_path):
    with open(str(tmp_path / "floor.obj"), "w") as fh:
        fh.write("v 552.8 0 0\nv 0 0 0\nv 0 0 559.2\nv 549.6 0 559.2\nf 1 2 3 4\n")
    with pytest.raises(Exception, match="undefined parameter"):
        mxml.parse_string(CBOX_XML, base_dir=str(tmp_path))
    d = mxml.parse_string(CBOX_XML, base_dir=str(tmp_path), depth=6, spp=16)
    sd = d.scene_dict
    assert d.integrator == dict(type="path", max_depth=6, rr_depth=5)
    assert [b["id"] for b in sd["bsdfs"]] == ["white", "light"]
    assert np.allclose(sd["bsdfs"][0]["reflectance"], [0.885809, 0.698859, 0.666422]) and sd["bsdfs"][1]["reflectance"] == 0.0
    floor, lamp = sd["meshes"]
    assert floor["bsdf"] == 0 and floor["emitter"] == -1 and floor["faces"].tolist() == [[0, 1, 2], [0, 2, 3]]
    assert lamp["bsdf"] == 1 and lamp["emitter"] == 0 and np.allclose(sd["emitters"][0]["radiance"], [18.387, 13.9873, 6.75357])
    assert np.allclose(lamp["positions"][:, 1], 548, atol=1e-3)                      # rotated into the xz plane, then lifted
    n = np.cross(lamp["positions"][1] - lamp["positions"][0], lamp["positions"][2] - lamp["positions"][0])
    assert n[1] < 0                                                                  # rotate x 90: +z -> -y (faces down)
    s = d.sensors[0]
    assert s["sampler"] == dict(sample_count=16, seed=7) and s["film"]["width"] == 32 and s["film"]["rfilter"] == ("gaussian", 0.5)
    assert s["fov_axis"] == "smaller" and abs(s["fov"] - 39.3077) < 1e-6
    assert np.allclose(s["to_world"][:3, 3], [278, 273, -800]) and np.allclose(s["to_world"][:3, 2], [0, 0, 1])


def test_load_dict_subset():
    """xml_v.cpp:100-260: nested dictionaries, rgb entries, references by key"""
    d = mxml.parse_dict({
        "type": "scene",
        "red": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.8, 0.1, 0.1]}},
        "wall": {"type": "rectangle", "to_world": mxml.translate([0, 0, -1]), "bsdf": {"type": "ref", "id": "red"}},
        "lamp": {"type": "rectangle", "flip_normals": True, "emitter": {"type": "area", "radiance": {"type": "spectrum", "value": 3.0}}},
        "integrator": {"type": "path", "max_depth": 3},
    })
    sd = d.scene_dict
    assert len(sd["meshes"]) == 2 and sd["meshes"][0]["bsdf"] == 0 and np.allclose(sd["bsdfs"][0]["reflectance"], [0.8, 0.1, 0.1])
    assert sd["meshes"][1]["emitter"] == 0 and sd["emitters"][0]["radiance"] == [3.0] * 3 and d.integrator["max_depth"] == 3
    with pytest.raises(Exception, match='Referenced id "nope" not found'):
        mxml.parse_dict({"type": "scene", "s": {"type": "rectangle", "b": {"type": "ref", "id": "nope"}}})
    with pytest.raises(Exception, match="Missing key 'type'"):
        mxml.parse_dict({"type": "scene", "s": {"to_world": 1}})


# ------------------------------------------------------------------------------------------------ image files
def test_image_roundtrips(tmp_path):
    """test_hdrfilm.py:74-160 (test03_develop) checks write -> read for exr / rgbe / pfm with these tolerances"""
    rng = np.random.default_rng(12345)
    img = rng.uniform(size=(37, 41, 3)).astype(np.float32)
    p = str(tmp_path / "a.pfm")
    bitmap.write_pfm(p, img)
    assert np.array_equal(bitmap.read_pfm(p), img)
    p = str(tmp_path / "a.rgbe")
    bitmap.write_rgbe(p, 1 + 0.1 * img)
    assert np.allclose(bitmap.read_rgbe(p), 1 + 0.1 * img, atol=1e-2)
    for comp in ("none", "zips", "zip"):
        p = str(tmp_path / ("a_%s.exr" % comp))
        ch = {"R": img[..., 0], "G": img[..., 1].astype(np.float16), "B": (img[..., 2] * 1000).astype(np.uint32)}
        bitmap.write_exr(p, ch, compression=comp)
        back, order = bitmap.read_exr(p)
        assert order == ["B", "G", "R"]
        for k in ch:
            assert back[k].dtype == ch[k].dtype and np.array_equal(back[k], ch[k])
    u8 = (img * 255).astype(np.uint8)
    p = str(tmp_path / "a.png")
    bitmap.write_png(p, u8)
    assert np.array_equal(bitmap.read_png(p), u8)
    tex = bitmap.read_rgb(p)
    assert tex.shape == (37, 41, 3) and np.allclose(tex, bitmap.srgb_to_linear(u8 / np.float32(255)), atol=1e-6)


def test_png_against_pillow(tmp_path):
    """our PNG codec against an independent implementation (all five scanline filters appear in Pillow's output)"""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:64, 0:48]
    img = np.stack([(xx * 5) % 256, (yy * 3 + xx) % 256, rng.integers(0, 255, (64, 48))], axis=2).astype(np.uint8)
    p = str(tmp_path / "pil.png")
    Image.fromarray(img).save(p, optimize=True)
    assert np.array_equal(bitmap.read_png(p), img)
    p2 = str(tmp_path / "ours.png")
    bitmap.write_png(p2, img)
    assert np.array_equal(np.asarray(Image.open(p2)), img)


def test_tabulated_spectra_become_rgb():
    """xml.cpp:1084-1140 + spectrum.cpp:41-86: wavelength:value spectra are integrated against the CIE observer in the RGB
    variants; cie1931_xyz spot values from src/librender/tests/test_spectra.py:7-15"""
    from mitsuba2_amd import spectrum as S
    assert np.allclose(S.cie1931_xyz(600.0), [1.0622, 0.631, 0.0008], atol=1e-6)
    assert np.allclose(S.cie1931_xyz([350.0, 840.0]), 0)
    d = mxml.parse_string("""<scene version="2.0.0">
        <bsdf type="diffuse" id="grey"><spectrum name="reflectance" value="360:0.5 600:0.5 830:0.5"/></bsdf>
        <bsdf type="diffuse" id="redish"><spectrum name="reflectance" value="400:0.04, 500:0.05, 600:0.55, 700:0.63"/></bsdf>
        <shape type="rectangle"><ref id="grey"/></shape>
        <shape type="rectangle"><ref id="redish"/>
            <emitter type="area"><spectrum name="radiance" value="400:0, 500:8, 600:15.6, 700:18.4"/></emitter></shape>
    </scene>""")
    grey, red = d.scene_dict["bsdfs"][0]["reflectance"], d.scene_dict["bsdfs"][1]["reflectance"]
    e_white = S.spectrum_to_rgb([360, 830], [float(S.MTS_CIE_Y_NORMALIZATION)] * 2, False)      # equal-energy white in sRGB primaries
    assert np.allclose(grey, np.minimum(0.5 * np.array(e_white), 1), atol=2e-3)
    assert red[0] > 0.4 and red[0] > 2 * red[1] and red[1] > red[2] and max(red) <= 1.0 and min(red) >= 0.0
    rad = d.scene_dict["emitters"][0]["radiance"]
    assert rad[0] > rad[1] > rad[2] > 0 and rad[0] > 10                                           # the warm Cornell-box light
    assert d.uses_tabulated_spectra
    with pytest.raises(Exception, match="increasing order"):
        mxml.parse_string('<scene version="2.0.0"><bsdf type="diffuse"><spectrum name="reflectance" value="500:1 400:1"/></bsdf></scene>')
    with pytest.raises(Exception, match="expected wavelength:value pairs"):
        mxml.parse_string('<scene version="2.0.0"><bsdf type="diffuse"><spectrum name="reflectance" value="500 400"/></bsdf></scene>')


@pytest.mark.parametrize("version", [3, 4])
def test_serialized_meshes(tmp_path, version):
    """src/shapes/serialized.cpp:190-336: multi-mesh files with the end-of-file dictionary, optional normals / texcoords,
    computed normals when the file has none (test_mesh.py:137-170 covers the same feature matrix for 'serialized')"""
    cb = scenes.cornell_box()
    a = dict(positions=np.asarray(cb["meshes"][7]["positions"], np.float32).reshape(-1, 3), faces=np.asarray(cb["meshes"][7]["faces"]).reshape(-1, 3))
    sphere = scenes.bumpy_sphere(8, 16)["meshes"][0]
    b = dict(positions=np.asarray(sphere["positions"], np.float32).reshape(-1, 3), faces=np.asarray(sphere["faces"]).reshape(-1, 3),
             normals=np.asarray(sphere["normals"], np.float32).reshape(-1, 3), texcoords=np.asarray(sphere["positions"], np.float32).reshape(-1, 3)[:, :2] * np.float32(0.1))
    path = str(tmp_path / "two.serialized")
    loaders.write_serialized(path, [a, b], version=version)
    m0 = loaders.load_serialized(path, 0)
    assert np.array_equal(m0["positions"], a["positions"]) and np.array_equal(m0["faces"], a["faces"]) and m0["texcoords"] is None
    assert np.allclose(np.linalg.norm(m0["normals"], axis=1), 1, atol=1e-5)              # computed (the file has none)
    assert loaders.load_serialized(path, 0, face_normals=True)["normals"] is None
    m1 = loaders.load_serialized(path, 1, to_world=mxml.translate([1, 2, 3]))
    assert np.allclose(m1["positions"], b["positions"] + [1, 2, 3]) and np.allclose(m1["normals"], b["normals"], atol=1e-6)
    assert np.array_equal(m1["texcoords"], b["texcoords"]) and np.array_equal(m1["faces"], b["faces"])
    with pytest.raises(RuntimeError, match="out of range"):
        loaders.load_serialized(path, 2)
    d = mxml.parse_string('<scene version="2.0.0"><shape type="serialized"><string name="filename" value="two.serialized"/>'
                          '<integer name="shape_index" value="1"/></shape></scene>', base_dir=str(tmp_path))
    assert d.scene_dict["meshes"][0]["faces"].shape == b["faces"].shape
    open(str(tmp_path / "bad.serialized"), "wb").write(b"\x00\x00\x04\x00abcdef")
    with pytest.raises(RuntimeError, match="invalid file format"):
        loaders.load_serialized(str(tmp_path / "bad.serialized"))


def test_xml_delta_emitters():
    """<emitter type="point|spot|directional"> at scene level (point.cpp:52-65, spot.cpp:68-91, directional.cpp:43-63)"""
    from mitsuba2_amd import xml as mxml, emitters as E
    x = """<scene version="2.0.0">
        <emitter type="point"><point name="position" x="1" y="2" z="3"/><rgb name="intensity" value="5, 6, 7"/></emitter>
        <emitter type="spot"><transform name="to_world"><lookat origin="0, 4, 0" target="0, 0, 0" up="0, 0, 1"/></transform>
            <spectrum name="intensity" value="2"/><float name="cutoff_angle" value="30"/></emitter>
        <emitter type="directional"><vector name="direction" x="0" y="-2" z="0"/></emitter>
        <shape type="rectangle"/></scene>"""
    desc = mxml.parse_string(x)
    em = desc.scene_dict["emitters"]
    assert [e["type"] for e in em] == ["point", "spot", "directional"]
    n = [E.normalize(e) for e in em]
    assert np.allclose(n[0]["to_world"][:3, 3], [1, 2, 3]) and n[0]["radiance"] == [5.0, 6.0, 7.0]
    assert n[1]["cutoff_angle"] == 30.0 and n[1]["beam_width"] == 22.5 and np.allclose(n[1]["to_world"][:3, 2], [0, -1, 0], atol=1e-6)
    assert n[1]["radiance"] == [2.0, 2.0, 2.0] and np.allclose(n[1]["to_world"][:3, 3], [0, 4, 0])
    assert np.allclose(n[2]["to_world"][:3, 2], [0, -1, 0]) and n[2]["radiance"] == [1.0, 1.0, 1.0]
    with pytest.raises(mxml.XMLError, match="Only one of the parameters"):
        mxml.parse_string(x.replace('<point name="position" x="1" y="2" z="3"/>', '<point name="position" x="1" y="2" z="3"/><transform name="to_world"><translate x="1"/></transform>'))
    with pytest.raises(mxml.XMLError, match="unreferenced"):
        mxml.parse_string(x.replace('<float name="cutoff_angle" value="30"/>', '<float name="cutoff" value="30"/>'))


def test_xml_nesting_bsdfs():
    """blendbsdf / mask in a scene file (the layouts of blendbsdf.cpp:36-49 and mask.cpp:44-60): children in document order, the weight as
    <spectrum> / <float> or a <texture>; the host module flattens the children behind the top-level records"""
    from mitsuba2_amd import bsdfs as B
    xml = """<scene version="2.0.0">
    <shape type="rectangle">
        <bsdf type="blendbsdf" id="mix">
            <texture name="weight" type="checkerboard"><float name="color0" value="0.9"/><float name="color1" value="0.1"/></texture>
            <bsdf type="conductor"/>
            <bsdf type="roughplastic"><spectrum name="diffuse_reflectance" value="0.1"/></bsdf>
        </bsdf>
    </shape>
    <shape type="rectangle">
        <bsdf type="mask" id="cutout">
            <float name="opacity" value="0.25"/>
            <bsdf type="twosided"><bsdf type="diffuse"><rgb name="reflectance" value="0.2, 0.6, 0.3"/></bsdf></bsdf>
        </bsdf>
    </shape>
    <shape type="rectangle"><bsdf type="diffuse"/></shape>
</scene>"""
    sd = mxml.parse_string(xml).scene_dict
    assert [b["type"] for b in sd["bsdfs"]] == ["blendbsdf", "mask", "diffuse"]
    recs = [B.normalize(b) for b in sd["bsdfs"]]
    mix, cut, plain = recs
    assert mix["type"] == B.BLEND and [c["type"] for c in mix["children"]] == [B.CONDUCTOR, B.ROUGHPLASTIC] and mix["reflectance"]["type"] == "checkerboard"
    assert np.allclose(mix["children"][1]["reflectance"], 0.1) and B.is_smooth(mix) and not B.is_transmissive(mix)
    assert cut["type"] == B.MASK and cut["reflectance"] == [0.25] * 3 and cut["children"][0]["twosided"] and cut["children"][0]["type"] == B.DIFFUSE
    assert B.is_transmissive(cut) and B.is_smooth(cut) and cut["id"] == "cutout"
    flat = B.flatten(recs)
    assert len(flat) == 6 and mix["nested"] == [3, 4] and cut["nested"] == [5, -1] and "nested" not in plain
    assert flat[3] is mix["children"][0] and flat[5] is cut["children"][0]
    with pytest.raises(RuntimeError, match="scalar"):
        B.normalize({"type": "mask", "opacity": [0.1, 0.2, 0.3], "a": {"type": "diffuse"}})
