"""CPU-side checks of the product boundary: the C-ABI library loads and exports every symbol that
include/mtsamd.h declares, argument validation works without a GPU, and the host-side mirror of the
reference's Python surface behaves like the reference (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mtsamd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:mtsamd|plugin)_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from mitsuba2_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 20
    # the Python binding covers exactly the declared entry points
    assert sorted(_lib.SYMBOLS) == declared
    handle = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), "libmtsamd.so does not export %s" % name
    assert _lib.lib().mtsamd_abi_version() == 6
    # ... and nothing else: the library is built with -fvisibility=hidden, internals (scheduler, kernel launchers) stay private
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    # exported *functions* (hipcc keeps the handle objects of __global__ kernels visible for its own registration: data, not code)
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] == "T")
    assert [s for s in exported if not s.startswith("__")] == declared, sorted(set(exported) ^ set(declared))
    # the two symbols PluginManager reads from a plugin .so (class.h:205-211, plugin.cpp:19-31)
    assert _lib.lib().plugin_name() == b"path_amd" and len(_lib.lib().plugin_descr()) > 0


def test_integrator_properties_follow_the_reference():
    # SamplingIntegrator (integrator.cpp:27-39): samples_per_pass default (size_t) -1, timeout default -1
    from mitsuba2_amd import render
    integ = render.PathIntegrator()
    assert integ.samples_per_pass == -1 and integ.timeout == -1.0
    sensor = render.make_sensor(dict(width=8, height=8, crop=(0, 0, 8, 8), rfilter="gaussian", rfilter_param=None, sample_count=12, seed=0,
                                     to_world=np.eye(4, dtype=np.float32), fov=40.0, near_clip=0.1, far_clip=100.0))
    d = render.PathIntegrator(samples_per_pass=4, timeout=2.5)._desc(sensor)
    assert d.samples_per_pass == 4 and d.timeout == 2.5 and d.profile == 0 and d.sample_count == 12


def test_every_entry_point_cites_the_reference():
    # each declaration in the header names the reference interface it replaces (file:line)
    text = open(os.path.join(ROOT, "include", "mtsamd.h")).read()
    for ref in ("scene.h:36", "scene.h:62", "integrator.cpp:52-176", "imageblock.cpp:80-172", "hdrfilm.cpp:249-320",
                "optix/common.h:15-35", "kdtree.h:2079-2174", "perspective.cpp", "rfilter.cpp:9-20"):
        assert ref in text, ref


def test_argument_validation_without_gpu():
    from mitsuba2_amd import _lib as L
    lib = L.lib()
    # null descriptors are rejected with a message, as the reference would Throw
    assert lib.mtsamd_scene_create(None, 0, None) < 0
    assert b"null" in lib.mtsamd_last_error()
    table = (C.c_float * 32)()
    radius, border = C.c_float(), C.c_int32()
    assert lib.mtsamd_rfilter_info(0, 0.5, 0.0, table, C.byref(radius), C.byref(border)) == 0
    assert radius.value == 2.0 and border.value == 2 and table[31] == 0.0      # gaussian.cpp:33-47, rfilter.cpp:9-20
    assert lib.mtsamd_rfilter_info(1, 0.5, 0.0, table, C.byref(radius), C.byref(border)) == 0
    assert border.value == 0 and abs(radius.value - 0.5) < 1e-3                # box.cpp:30-36
    assert lib.mtsamd_rfilter_info(7, 0.5, 0.0, table, C.byref(radius), C.byref(border)) == -5
    with pytest.raises(RuntimeError, match="unsupported reconstruction filter"):
        L.check(lib.mtsamd_rfilter_info(7, 0.5, 0.0, table, C.byref(radius), C.byref(border)))


def test_rfilter_tables_match_oracle(oracle):
    from mitsuba2_amd import render
    for cls, kind, params in ((render.GaussianFilter, 0, (0.5,)), (render.GaussianFilter, 0, (1.5,)), (render.BoxFilter, 1, (0.5,)),
                              (render.BoxFilter, 1, (0.4,)), (render.TentFilter, 2, ()), (render.CatmullRomFilter, 3, ()),
                              (render.MitchellFilter, 4, (1 / 3, 1 / 3)), (render.MitchellFilter, 4, (0.2, 0.6)),
                              (render.LanczosFilter, 5, (3,)), (render.LanczosFilter, 5, (2,))):
        f = cls(*params)
        tbl, radius, border = oracle.rfilter_table(kind, *params)
        assert (f._table == tbl).all() and f.radius() == radius and f.border_size() == border
        for x in np.linspace(-radius * 1.2, radius * 1.2, 41):
            idx = min(int(abs(np.float32(x) * np.float32(31.0 / radius))), 31)
            assert f.eval_discretized(x) == tbl[idx]


def test_parse_fov():
    # src/librender/sensor.cpp:119-169
    from mitsuba2_amd.render import parse_fov
    assert parse_fov(fov=45.0) == 45.0
    assert abs(parse_fov(fov=45.0, fov_axis="y", aspect=2.0) - np.degrees(2 * np.arctan(np.tan(np.radians(22.5)) * 2.0))) < 1e-4
    assert parse_fov(fov=30.0, fov_axis="smaller", aspect=2.0) == parse_fov(fov=30.0, fov_axis="y", aspect=2.0)
    assert parse_fov(fov=30.0, fov_axis="larger", aspect=2.0) == 30.0
    # default 50mm focal length -> diagonal fov of a 36x24 mm frame
    diag = 2 * np.degrees(np.arctan(np.sqrt(36 ** 2 + 24 ** 2) / 100.0))
    width = 2 * np.tan(np.radians(diag) / 2) / np.sqrt(1 + 1 / 1.5 ** 2)
    assert abs(parse_fov(aspect=1.5) - np.degrees(2 * np.arctan(width / 2))) < 1e-3
    with pytest.raises(RuntimeError):
        parse_fov(fov=30.0, focal_length="50mm")
    with pytest.raises(RuntimeError):
        parse_fov(fov=30.0, fov_axis="sideways")
    with pytest.raises(RuntimeError):
        parse_fov(fov=180.0)


def test_film_and_integrator_parameter_checks():
    from mitsuba2_amd import render
    film = render.HDRFilm(32, 24)
    assert film.size() == (32, 24) and film.crop_size() == (32, 24) and film.crop_offset() == (0, 0)
    assert isinstance(film.reconstruction_filter(), render.GaussianFilter)        # film.cpp:46-50
    film.set_crop_window((4, 2), (10, 12))
    assert film.crop_offset() == (4, 2) and film.crop_size() == (10, 12)
    for bad in (((-1, 0), (4, 4)), ((0, 0), (0, 4)), ((30, 0), (4, 4))):
        with pytest.raises(RuntimeError):
            film.set_crop_window(*bad)
    assert render.HDRFilm().size() == (768, 576)                                    # film.cpp:11-14
    assert render.IndependentSampler().sample_count() == 4                          # sampler.cpp:7-8
    with pytest.raises(RuntimeError):
        render.PathIntegrator(max_depth=-2)                                         # test_integrator.py:90-106
    with pytest.raises(RuntimeError):
        render.PathIntegrator(rr_depth=0)
    assert render.PathIntegrator().rr_depth == 5
    with pytest.raises(RuntimeError):
        render.PerspectiveCamera(near_clip=0.0)
    with pytest.raises(RuntimeError):
        render.PerspectiveCamera(near_clip=10.0, far_clip=1.0)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from mitsuba2_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_synthetic_scenes():
    from mitsuba2_amd import scenes
    sd = scenes.cornell_box()
    assert sum(m["faces"].shape[0] for m in sd["meshes"]) == 36
    assert sum(1 for m in sd["meshes"] if m["emitter"] >= 0) == 1
    # every room surface faces the room centre, every block face faces outwards (diffuse is one-sided)
    c = np.array([278.0, 274.4, 279.6])
    for m in sd["meshes"][:6]:
        p, f = m["positions"].astype(np.float64), m["faces"]
        for tri in f:
            n = np.cross(p[tri[1]] - p[tri[0]], p[tri[2]] - p[tri[0]])
            assert np.dot(n, c - p[tri].mean(0)) > 0
    st = scenes.stairs(20)
    assert st["meshes"][0]["positions"].shape == (80, 3) and st["meshes"][0]["faces"].shape == (78, 3)
    m = scenes.look_at([0, 0, 0], [0, 0, 1], [0, 1, 0])
    assert np.allclose(m, np.eye(4))
