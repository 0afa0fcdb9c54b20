"""ctypes binding of the CPU oracle (oracle/libmts_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.environ.get("MTS_ORACLE_LIB", os.path.join(ORACLE_DIR, "libmts_oracle.so"))   # override: sanitizer builds of the oracle

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)


class RenderDesc(C.Structure):
    _fields_ = [("to_world", C.c_float * 16), ("fov_x_deg", C.c_float), ("near_clip", C.c_float), ("far_clip", C.c_float),
                ("film_w", C.c_int32), ("film_h", C.c_int32), ("crop_x", C.c_int32), ("crop_y", C.c_int32), ("crop_w", C.c_int32),
                ("crop_h", C.c_int32), ("rfilter", C.c_int32), ("rfilter_param", C.c_float), ("rfilter_param2", C.c_float), ("spp", C.c_int32),
                ("base_seed", C.c_uint64), ("max_depth", C.c_int32), ("rr_depth", C.c_int32), ("filter_analytic", C.c_int32),
                ("film_rgb", C.c_int32), ("integrator", C.c_int32), ("emitter_samples", C.c_int32), ("bsdf_samples", C.c_int32),
                ("hide_emitters", C.c_int32), ("aperture_radius", C.c_float), ("focus_distance", C.c_float)]


class BsdfDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("twosided", C.c_int32), ("reflectance", C.c_float * 3), ("specular_reflectance", C.c_float * 3),
                ("specular_transmittance", C.c_float * 3), ("eta", C.c_float * 3), ("k", C.c_float * 3), ("int_ior", C.c_float),
                ("ext_ior", C.c_float), ("alpha_u", C.c_float), ("alpha_v", C.c_float), ("distribution", C.c_int32),
                ("sample_visible", C.c_int32), ("nonlinear", C.c_int32), ("uniform_mask", C.c_int32)]


def bsdf_desc(plugin_dict, normalized=None):
    """plugin dictionary -> mo_bsdf_desc (parameter defaults via the host module mitsuba2_amd.bsdfs, shared with the product)"""
    from mitsuba2_amd import bsdfs
    n = normalized if normalized is not None else bsdfs.normalize(plugin_dict)
    d = BsdfDesc()
    d.type, d.twosided = n["type"], int(n["twosided"])
    refl = [0.5, 0.5, 0.5] if isinstance(n["reflectance"], dict) else n["reflectance"]
    for name, v in (("reflectance", refl), ("specular_reflectance", n["specular_reflectance"]),
                    ("specular_transmittance", n["specular_transmittance"]), ("eta", n["eta"]), ("k", n["k"])):
        setattr(d, name, (C.c_float * 3)(*v))
    d.int_ior, d.ext_ior, d.alpha_u, d.alpha_v = n["int_ior"], n["ext_ior"], n["alpha_u"], n["alpha_v"]
    d.distribution, d.sample_visible, d.nonlinear = n["distribution"], int(n["sample_visible"]), int(n["nonlinear"])
    d.uniform_mask = n["uniform_mask"]
    return d, n


_lib = None


def build():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "libmts_oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.mo_scene_new.restype = C.c_void_p
        L.mo_scene_free.argtypes = [C.c_void_p]
        L.mo_scene_add_mesh.argtypes = [C.c_void_p, C.c_uint32, f32p, f32p, f32p, C.c_uint32, u32p, C.c_int, f32p, f32p]
        L.mo_scene_finalize.argtypes = [C.c_void_p]
        L.mo_scene_add_texture.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p]
        L.mo_scene_set_texture.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.mo_scene_set_texture_transform.argtypes = [C.c_void_p, C.c_uint32, f32p]
        L.mo_scene_add_checkerboard.argtypes = [C.c_void_p, f32p, f32p, f32p]
        L.mo_scene_update_texture.argtypes = [C.c_void_p, C.c_uint32, f32p]
        L.mo_scene_set_reflectance.argtypes = [C.c_void_p, C.c_uint32, f32p]
        L.mo_scene_set_naive.argtypes = [C.c_void_p, C.c_int]
        L.mo_scene_prim_count.argtypes = [C.c_void_p]
        L.mo_scene_prim_count.restype = C.c_uint32
        L.mo_scene_emitter_area.argtypes = [C.c_void_p, C.c_uint32]
        L.mo_scene_emitter_area.restype = C.c_float
        vp = C.c_void_p
        L.mo_ray_intersect.argtypes = [vp, C.c_uint64] + [vp] * 8 + [C.c_int] + [vp] * 5
        L.mo_ray_test.argtypes = [vp, C.c_uint64] + [vp] * 8 + [C.c_int, vp]
        L.mo_fill_si.argtypes = [vp, C.c_uint64] + [vp] * 7
        L.mo_packet_ray_intersect.argtypes = [vp, C.c_uint64] + [vp] * 13
        L.mo_render.argtypes = [vp, C.POINTER(RenderDesc), C.c_int, C.c_int, C.c_int, vp, vp]
        L.mo_sample_radiance.argtypes = [vp, C.POINTER(RenderDesc), C.c_uint64, C.c_uint64, vp, vp]
        L.mo_render_rows.argtypes = [vp, C.POINTER(RenderDesc), C.c_int, C.c_int, vp]
        L.mo_render_window.argtypes = [vp, C.POINTER(RenderDesc), C.c_int, C.c_int, C.c_int, C.c_int, vp]
        L.mo_film_develop.argtypes = [vp, C.c_uint64, vp]
        L.mo_libm_eval.argtypes = [C.c_int, C.c_uint64, vp, vp, vp]
        L.mo_render_adjoint_param.argtypes = [vp, C.POINTER(RenderDesc), vp, vp, vp, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_double)]
        L.mo_libm_eval.restype = None
        L.mo_render_adjoint.argtypes = [vp, C.POINTER(RenderDesc), vp, vp, vp, vp, vp]
        L.mo_scene_set_emitter_radiance.argtypes = [vp, C.c_uint32, vp]
        L.mo_render_adjoint_envmap.argtypes = [vp, C.POINTER(RenderDesc), vp, vp, vp]
        L.mo_scene_update_envmap.argtypes = [vp, vp, C.c_int]
        L.mo_camera_rays.argtypes = [C.POINTER(RenderDesc), C.c_uint64] + [vp] * 7
        L.mo_imageblock_put.argtypes = [C.c_int] * 6 + [C.c_float, C.c_float, C.c_int, C.c_int, C.c_uint64, vp, vp, vp]
        L.mo_rfilter_table.argtypes = [C.c_int, C.c_float, C.c_float, vp, f32p, C.POINTER(C.c_int)]
        L.mo_kat_tea32.restype = C.c_uint32
        L.mo_kat_tea32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.mo_kat_tea64_u32.restype = C.c_uint64
        L.mo_kat_tea64_u32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.mo_kat_tea64_u64.restype = C.c_uint64
        L.mo_kat_tea64_u64.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
        L.mo_kat_tea_float32.restype = C.c_float
        L.mo_kat_tea_float32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.mo_kat_tea_float64.restype = C.c_double
        L.mo_kat_tea_float64.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
        L.mo_kat_pcg32.argtypes = [C.c_uint64, C.c_uint64, C.c_int, vp, vp]
        L.mo_kat_warp.argtypes = [C.c_int, C.c_uint64, vp, vp, vp]
        L.mo_kat_coordinate_system.argtypes = [vp, vp, vp]
        L.mo_kat_spiral.argtypes = [C.c_int] * 7 + [vp]
        L.mo_kat_morton.argtypes = [C.c_uint32, vp]
        L.mo_kat_distr.restype = C.c_float
        L.mo_kat_distr.argtypes = [C.c_uint32, vp, vp, C.c_uint32, vp, vp, vp]
        L.mo_kat_diffuse.argtypes = [vp] * 9
        L.mo_kat_sample_emitter.argtypes = [vp] * 4
        L.mo_scene_set_spectral.argtypes = [vp, C.c_char_p]
        L.mo_scene_add_delta_emitter.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_float, C.c_float]
        L.mo_scene_set_bsdf.argtypes = [vp, C.c_uint32, C.POINTER(BsdfDesc)]
        L.mo_scene_add_constant_emitter.argtypes = [vp, f32p]
        L.mo_scene_set_emitter_order.argtypes = [vp, C.c_uint32, vp]
        L.mo_scene_add_envmap_emitter.argtypes = [vp, C.c_int, C.c_int, vp, C.c_float, vp]
        L.mo_kat_hier2d.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint64, vp, vp]
        L.mo_kat_bilinear_to_square.argtypes = [C.c_float] * 6 + [vp]
        L.mo_kat_envmap.argtypes = [C.c_int, C.c_int, vp, C.c_float, vp, C.c_uint64, vp, vp]
        L.mo_kat_fresnel.argtypes = [C.c_float, C.c_float, vp]
        L.mo_kat_fresnel_conductor.argtypes = [C.c_float] * 3
        L.mo_kat_fresnel_conductor.restype = C.c_float
        L.mo_kat_fresnel_diffuse.argtypes = [C.c_float]
        L.mo_kat_fresnel_diffuse.restype = C.c_float
        L.mo_kat_microfacet.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_uint64, vp, vp, vp]
        L.mo_kat_microfacet_sample.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_uint64, vp, vp, vp, vp]
        L.mo_kat_bsdf.argtypes = [C.POINTER(BsdfDesc), C.c_uint64, vp, vp, vp, vp]
        L.mo_scene_set_nested_bsdf.argtypes = [vp, C.c_uint32, C.c_int, C.c_float, C.c_int, C.POINTER(BsdfDesc), C.POINTER(BsdfDesc), C.c_int, C.c_int]
        L.mo_kat_nested_bsdf.argtypes = [C.c_int, C.c_float, C.c_int, C.POINTER(BsdfDesc), C.POINTER(BsdfDesc), C.c_uint64, vp, vp, vp, vp]
        L.mo_kat_gauss_legendre.argtypes = [C.c_int, vp, vp]
        L.mo_kat_roughplastic_tables.argtypes = [C.POINTER(BsdfDesc), vp]
        L.mo_kat_srgb_model_fetch.argtypes = [C.c_char_p, vp, vp]
        L.mo_kat_spectral.argtypes = [C.c_float, vp, C.c_float, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class OracleScene:
    def __init__(self, scene_dict, naive=False, spectral_path=None):
        L = lib()
        self.h = C.c_void_p(L.mo_scene_new())
        self.sd = scene_dict
        self.tex_of_bsdf = {}
        self.tex_of_child = {}                                # (bsdf, child) -> texture of a blendbsdf / mask child
        from mitsuba2_amd import bsdfs as B

        def add_texture(spec):
            uvm = None
            if spec.get("to_uv") is not None:
                m = np.asarray(spec["to_uv"], np.float32).reshape(4, 4)
                uvm = _f([m[0, 0], m[0, 1], m[0, 2], m[1, 0], m[1, 1], m[1, 2]])
            if spec.get("type") == "checkerboard":
                c0, c1 = _f(B._rgb(spec.get("color0"), 0.4)), _f(B._rgb(spec.get("color1"), 0.2))
                t = L.mo_scene_add_checkerboard(self.h, c0.ctypes.data_as(f32p), c1.ctypes.data_as(f32p),
                                                uvm.ctypes.data_as(f32p) if uvm is not None else None)
            else:
                data = _f(spec["data"])
                t = L.mo_scene_add_texture(self.h, data.shape[1], data.shape[0], data.ctypes.data_as(f32p))
                if uvm is not None:
                    assert L.mo_scene_set_texture_transform(self.h, t, uvm.ctypes.data_as(f32p)) == 0
            assert t >= 0
            return t

        # texture table in the order of the product's flattened BSDF table (top-level records, then the children of nests)
        norm = [B.normalize(b) for b in scene_dict["bsdfs"]]   # unwraps `twosided`, whatever the nested key is called
        for bi, b in enumerate(norm):
            if isinstance(b.get("reflectance"), dict):
                self.tex_of_bsdf[bi] = add_texture(b["reflectance"])
        for bi, b in enumerate(norm):
            for k, c in enumerate(b.get("children", [])):
                if isinstance(c.get("reflectance"), dict):
                    self.tex_of_child[(bi, k)] = add_texture(c["reflectance"])
        self.shapes_of_bsdf = {}
        for si, m in enumerate(scene_dict["meshes"]):
            self.shapes_of_bsdf.setdefault(m["bsdf"], []).append(si)
        for m in scene_dict["meshes"]:
            pos = _f(m["positions"]).reshape(-1, 3)
            faces = np.ascontiguousarray(m["faces"], dtype=np.uint32).reshape(-1, 3)
            nrm = _f(m["normals"]) if m.get("normals") is not None else None
            uv = _f(m["texcoords"]) if m.get("texcoords") is not None else None
            bd, bn = bsdf_desc(scene_dict["bsdfs"][m["bsdf"]])
            rb = bn["reflectance"]
            refl = _f([0.5, 0.5, 0.5]) if isinstance(rb, dict) else _f(rb)
            em = _f(scene_dict["emitters"][m["emitter"]]["radiance"]) if m.get("emitter", -1) >= 0 else None
            rc = L.mo_scene_add_mesh(self.h, pos.shape[0], pos.ctypes.data_as(f32p), nrm.ctypes.data_as(f32p) if nrm is not None else None,
                                     uv.ctypes.data_as(f32p) if uv is not None else None, faces.shape[0], faces.ctypes.data_as(u32p), 0,
                                     refl.ctypes.data_as(f32p), em.ctypes.data_as(f32p) if em is not None else None)
            assert rc >= 0, rc
            if bn["type"] in (8, 9):                          # blendbsdf / mask over plain children (mo_scene_set_nested_bsdf)
                kids = [bsdf_desc(None, c)[0] for c in bn["children"]]
                w = 0.5 if isinstance(bn["reflectance"], dict) else bn["reflectance"][0]
                assert L.mo_scene_set_nested_bsdf(self.h, rc, 1 if bn["type"] == 8 else 2, C.c_float(w), int(bn["twosided"]), C.byref(kids[0]),
                                                  C.byref(kids[1]) if len(kids) > 1 else None, self.tex_of_child.get((m["bsdf"], 0), -1),
                                                  self.tex_of_child.get((m["bsdf"], 1), -1)) == 0
            elif bn["type"] != 0 or bn["twosided"] or bn["uniform_mask"]:
                assert L.mo_scene_set_bsdf(self.h, rc, C.byref(bd)) == 0
            if m["bsdf"] in self.tex_of_bsdf:
                assert L.mo_scene_set_texture(self.h, rc, self.tex_of_bsdf[m["bsdf"]]) == 0
        # emitter order = the scene dictionary's (Scene::m_emitters follows the scene description, scene.cpp:31-56)
        emitters = scene_dict.get("emitters", [])
        created = [m["emitter"] for m in scene_dict["meshes"] if m.get("emitter", -1) >= 0]      # dict index of oracle emitter k
        for ei, e in enumerate(emitters):
            if e.get("type", "area") == "constant":
                rad = _f(e["radiance"])
                assert L.mo_scene_add_constant_emitter(self.h, rad.ctypes.data_as(f32p)) == len(created)
                created.append(ei)
            elif e.get("type", "area") == "envmap":
                img = _f(e["data"])
                tw = _f(np.asarray(e["to_world"], np.float32).reshape(4, 4)[:3, :3]) if e.get("to_world") is not None else None
                assert L.mo_scene_add_envmap_emitter(self.h, img.shape[1], img.shape[0], _p(img), float(e.get("scale", 1.0)), _p(tw)) == len(created)
                created.append(ei)
            elif e.get("type", "area") in ("point", "spot", "directional"):
                from mitsuba2_amd import emitters as E
                n = E.normalize(e)                                    # parameter defaults shared with the product
                tw = n["to_world"]
                pos, direction, rot = _f(tw[:3, 3]), _f(tw[:3, 2]), _f(tw[:3, :3])
                rc = L.mo_scene_add_delta_emitter(self.h, n["type"], _p(_f(n["radiance"])), _p(pos), _p(direction), _p(rot),
                                                  n["cutoff_angle"], n["beam_width"])
                assert rc == len(created), rc
                created.append(ei)
        if created != list(range(len(created))):
            order = np.array([created.index(i) for i in range(len(created))], dtype=np.uint32)
            assert L.mo_scene_set_emitter_order(self.h, len(created), _p(order)) == 0
        assert L.mo_scene_finalize(self.h) == 0
        L.mo_scene_set_naive(self.h, 1 if naive else 0)
        if spectral_path is not None:
            rc = L.mo_scene_set_spectral(self.h, spectral_path.encode())
            if rc != 0:
                raise RuntimeError("oracle: spectral setup failed (%d)" % rc)

    def __del__(self):
        if getattr(self, "h", None):
            lib().mo_scene_free(self.h)
            self.h = None

    def update_texture(self, bsdf, data):
        data = _f(data)
        assert lib().mo_scene_update_texture(self.h, self.tex_of_bsdf[bsdf], data.ctypes.data_as(f32p)) == 0

    def set_bsdf_reflectance(self, bsdf, rgb):
        rgb = _f(rgb)
        for si in self.shapes_of_bsdf.get(bsdf, []):
            assert lib().mo_scene_set_reflectance(self.h, si, rgb.ctypes.data_as(f32p)) == 0

    def set_naive(self, naive):
        lib().mo_scene_set_naive(self.h, 1 if naive else 0)

    def ray_intersect(self, o, d, mint, maxt, naive=True):
        o, d, mint, maxt = _f(o), _f(d), _f(mint), _f(maxt)
        n = o.shape[0]
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        t = np.empty(n, np.float32); prim = np.empty(n, np.uint32); shape = np.empty(n, np.uint32)
        u = np.empty(n, np.float32); v = np.empty(n, np.float32)
        lib().mo_ray_intersect(self.h, n, *[_p(c) for c in cols], _p(mint), _p(maxt), 1 if naive else 0, _p(t), _p(prim), _p(shape), _p(u), _p(v))
        return t, prim, shape, u, v

    def ray_test(self, o, d, mint, maxt, naive=True):
        o, d, mint, maxt = _f(o), _f(d), _f(mint), _f(maxt)
        n = o.shape[0]
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        hit = np.empty(n, np.uint8)
        lib().mo_ray_test(self.h, n, *[_p(c) for c in cols], _p(mint), _p(maxt), 1 if naive else 0, _p(hit))
        return hit.astype(bool)

    def packet_intersect(self, o, d, mint, maxt):
        """closest hit + any hit through the 8-wide packet traversal (oracle/mo_packet.c): t, prim, u, v, hit"""
        o, d, mint, maxt = _f(o), _f(d), _f(mint), _f(maxt)
        n = o.shape[0]
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        t = np.empty(n, np.float32); prim = np.empty(n, np.uint32); u = np.empty(n, np.float32); v = np.empty(n, np.float32)
        hit = np.empty(n, np.uint8)
        lib().mo_packet_ray_intersect(self.h, n, *[_p(c) for c in cols], _p(mint), _p(maxt), _p(t), _p(prim), _p(u), _p(v), _p(hit))
        return t, prim, u, v, hit.astype(bool)

    def fill_si(self, d, prim, u, v):
        d = _f(d); n = d.shape[0]
        cols = [np.ascontiguousarray(d[:, k]) for k in range(3)]
        prim = np.ascontiguousarray(prim, dtype=np.uint32); u = _f(u); v = _f(v)
        out = np.empty((n, 26), np.float32)
        lib().mo_fill_si(self.h, n, *[_p(c) for c in cols], _p(prim), _p(u), _p(v), _p(out))
        return out

    def render(self, desc, mode=1, n_threads=0, block_size=0):
        film = np.empty((desc.crop_h, desc.crop_w, 5), np.float32)
        stats = np.zeros(3, np.uint64)
        rc = lib().mo_render(self.h, C.byref(desc), mode, n_threads, block_size, _p(film), _p(stats))
        if rc != 0:
            raise RuntimeError("oracle render failed (%d)" % rc)
        return film, stats

    def render_rows(self, desc, row0, row1):
        film = np.empty((desc.crop_h, desc.crop_w, 5), np.float32)
        assert lib().mo_render_rows(self.h, C.byref(desc), row0, row1, _p(film)) == 0
        return film

    def render_window(self, desc, row0, row1, col0, col1):
        """wavefront-mode film of the samples of the pixels [row0, row1) x [col0, col1) only (full-size film array)"""
        film = np.empty((desc.crop_h, desc.crop_w, 5), np.float32)
        assert lib().mo_render_window(self.h, C.byref(desc), row0, row1, col0, col1, _p(film)) == 0
        return film

    def render_image(self, desc):
        """mitsuba.python.autodiff._render_helper: RGB / (W + 1e-8), plus the film (desc.film_rgb must be 1)."""
        assert desc.film_rgb == 1
        film, _ = self.render(desc, mode=1)
        return film[..., :3] / (film[..., 4:5] + np.float32(1e-8)), film

    def render_adjoint(self, desc, dimage, film, n_shapes, tex_floats, n_emitters=None):
        """gradients w.r.t. constant reflectances (per shape), texels, and -- with n_emitters -- area-light radiances"""
        dimage = _f(dimage); film = _f(film)
        gs = np.zeros((n_shapes, 3), np.float32); gt = np.zeros(max(tex_floats, 1), np.float32)
        ge = np.zeros((max(n_emitters or 0, 1), 3), np.float32)
        rc = lib().mo_render_adjoint(self.h, C.byref(desc), _p(dimage), _p(film), _p(gs), _p(gt), _p(ge) if n_emitters is not None else None)
        if rc != 0:
            raise RuntimeError("oracle adjoint failed (%d)" % rc)
        if n_emitters is not None:
            return gs, gt[:tex_floats], ge[:n_emitters]
        return gs, gt[:tex_floats]

    def render_adjoint_param(self, desc, dimage, film, shapes, kind, comp, h):
        """d(loss)/d(one scalar BSDF parameter of the given shapes): forward-mode replay with detached sampling (checker of
        mtsamd_render_adjoint_param)"""
        dimage = _f(dimage); film = _f(film)
        mask = np.zeros(len(self.sd["meshes"]), np.uint8)
        mask[list(shapes)] = 1
        g = C.c_double(0.0)
        rc = lib().mo_render_adjoint_param(self.h, C.byref(desc), _p(dimage), _p(film), mask.ctypes.data_as(C.c_void_p), int(kind), int(comp), float(h), C.byref(g))
        if rc != 0:
            raise RuntimeError("oracle parameter adjoint failed (%d)" % rc)
        return g.value

    def render_adjoint_envmap(self, desc, dimage, film, shape):
        """gradient w.r.t. the texels of the envmap emitter (`shape` = (h, w, 3))"""
        dimage = _f(dimage); film = _f(film)
        g = np.zeros(shape, np.float32)
        rc = lib().mo_render_adjoint_envmap(self.h, C.byref(desc), _p(dimage), _p(film), _p(g))
        if rc != 0:
            raise RuntimeError("oracle envmap adjoint failed (%d)" % rc)
        return g

    def update_envmap(self, data, rebuild_warp=True):
        data = _f(data)
        assert lib().mo_scene_update_envmap(self.h, _p(data), 1 if rebuild_warp else 0) == 0

    def set_emitter_radiance(self, emitter, rgb):
        rgb = _f(rgb)
        assert lib().mo_scene_set_emitter_radiance(self.h, emitter, rgb.ctypes.data_as(f32p)) == 0

    def sample_radiance(self, desc, first, count):
        rgba = np.empty((count, 4), np.float32); pos = np.empty((count, 2), np.float32)
        assert lib().mo_sample_radiance(self.h, C.byref(desc), first, count, _p(rgba), _p(pos)) == 0
        return rgba, pos

    def sample_emitter(self, ref_p, sample2):
        out = np.empty(15, np.float32)
        lib().mo_kat_sample_emitter(self.h, _p(_f(ref_p)), _p(_f(sample2)), _p(out))
        return out


def make_desc(params, analytic=False, film_rgb=False):
    """scenes.*_sensor() dict -> oracle RenderDesc (fov must already be the horizontal fov)."""
    d = RenderDesc()
    d.to_world = (C.c_float * 16)(*np.asarray(params["to_world"], dtype=np.float32).reshape(-1).tolist())
    d.fov_x_deg = params["fov"]
    d.near_clip, d.far_clip = params["near_clip"], params["far_clip"]
    d.film_w, d.film_h = params["width"], params["height"]
    d.crop_x, d.crop_y, d.crop_w, d.crop_h = params["crop"]
    d.rfilter = RFILTERS[params["rfilter"]]
    rp = params.get("rfilter_param")
    rp = [] if rp is None else (list(rp) if isinstance(rp, (list, tuple)) else [rp])
    rp = rp + RFILTER_DEFAULTS[params["rfilter"]][len(rp):]
    d.rfilter_param, d.rfilter_param2 = float(rp[0]), float(rp[1])
    d.spp = params["sample_count"]
    d.base_seed = params["seed"]
    d.max_depth, d.rr_depth = params["max_depth"], params["rr_depth"]
    d.filter_analytic = 1 if analytic else 0
    d.film_rgb = 1 if film_rgb else 0
    d.integrator = {"path": 0, "direct": 1, "depth": 2}[params.get("integrator", "path")]
    d.emitter_samples, d.bsdf_samples = params.get("emitter_samples", 0), params.get("bsdf_samples", 0)
    d.hide_emitters = 1 if params.get("hide_emitters", False) else 0
    if params.get("aperture_radius") is not None:          # thinlens.cpp:112-117, sensor.cpp:104
        d.aperture_radius = params["aperture_radius"] if params["aperture_radius"] != 0 else float(np.finfo(np.float32).eps) / 2
        d.focus_distance = params["focus_distance"] if params.get("focus_distance") is not None else params["far_clip"]
    return d


def film_develop(xyzaw):
    xyzaw = _f(xyzaw)
    n = xyzaw.size // 5
    out = np.empty(xyzaw.shape[:-1] + (4,), np.float32)
    lib().mo_film_develop(_p(xyzaw), n, _p(out))
    return out


LIBM_FUNCTIONS = ("sin", "cos", "tan", "exp", "log", "erf", "acos", "atan2", "atanh", "cosh")


def libm_eval(name, x, y=None):
    """oracle/mo_libm.h on float32 arrays (atan2: (y, x))"""
    x = _f(x)
    y = _f(y) if y is not None else x
    out = np.empty_like(x)
    lib().mo_libm_eval(LIBM_FUNCTIONS.index(name), x.size, _p(x), _p(y), _p(out))
    return out


def camera_rays(desc, sx, sy, aperture=None):
    sx, sy = _f(sx), _f(sy)
    ap = _f(aperture).reshape(-1, 2) if aperture is not None else None
    n = sx.shape[0]
    o = np.empty((n, 3), np.float32); d = np.empty((n, 3), np.float32)
    mint = np.empty(n, np.float32); maxt = np.empty(n, np.float32)
    lib().mo_camera_rays(C.byref(desc), n, _p(sx), _p(sy), _p(ap) if ap is not None else None, _p(o), _p(d), _p(mint), _p(maxt))
    return o, d, mint, maxt


# mo_render modes (oracle/mo_api.h): scalar blocks, wavefront, packet blocks (8-wide ray queries), scalar emulation of the packet schedule
SCALAR_MODE, WAVEFRONT_MODE, PACKET_MODE, PACKET_CHECK_MODE = 0, 1, 2, 3

RFILTERS = {"gaussian": 0, "box": 1, "tent": 2, "catmullrom": 3, "mitchell": 4, "lanczos": 5}
RFILTER_DEFAULTS = {"gaussian": [0.5, 0.0], "box": [0.5, 0.0], "tent": [0.0, 0.0], "catmullrom": [0.0, 0.0], "mitchell": [1.0 / 3.0, 1.0 / 3.0],
                    "lanczos": [3.0, 0.0]}


def imageblock_put(w, h, ox, oy, ch, rfilter, param, border, pos, values, analytic=False, param2=0.0):
    pos, values = _f(pos).reshape(-1, 2), _f(values).reshape(-1, ch)
    tbl = np.empty(32, np.float32); radius = C.c_float(); b = C.c_int()
    lib().mo_rfilter_table(rfilter, param, param2, _p(tbl), C.byref(radius), C.byref(b))
    bs = b.value if border else 0
    data = np.zeros((h + 2 * bs, w + 2 * bs, ch), np.float32)
    lib().mo_imageblock_put(w, h, ox, oy, ch, rfilter, param, param2, 1 if border else 0, 1 if analytic else 0, pos.shape[0], _p(pos), _p(values), _p(data))
    return data


def rfilter_table(rfilter, param=0.0, param2=0.0):
    tbl = np.empty(32, np.float32); radius = C.c_float(); b = C.c_int()
    lib().mo_rfilter_table(rfilter, param, param2, _p(tbl), C.byref(radius), C.byref(b))
    return tbl, radius.value, b.value


def srgb_model_fetch(path, rgb):
    out = np.empty(3, np.float32)
    lib().mo_kat_srgb_model_fetch(path.encode(), _p(_f(rgb)), _p(out))
    return out


def spectral_kat(sample, coeff, d65_scale):
    out = np.empty(19, np.float32)
    lib().mo_kat_spectral(C.c_float(sample), _p(_f(coeff)), C.c_float(d65_scale), _p(out))
    return dict(wav=out[0:4], weight=out[4:8], refl=out[8:12], d65=out[12:16], xyz=out[16:19])


def fresnel(cos_theta_i, eta):
    out = np.zeros(4, np.float32)
    lib().mo_kat_fresnel(float(cos_theta_i), float(eta), _p(out))
    return out


def fresnel_conductor(cos_theta_i, eta_r, eta_i):
    return lib().mo_kat_fresnel_conductor(float(cos_theta_i), float(eta_r), float(eta_i))


def microfacet(ggx, alpha_u, alpha_v, visible, which, v, wi):
    """which: 'eval' (m = v), 'pdf' (wi, m = v), 'smith_g1' (v, m = wi)"""
    v, wi = _f(v).reshape(-1, 3), _f(wi).reshape(-1, 3)
    out = np.zeros(v.shape[0], np.float32)
    lib().mo_kat_microfacet(int(ggx), alpha_u, alpha_v, int(visible), {"eval": 0, "pdf": 1, "smith_g1": 2}[which], v.shape[0], _p(v), _p(wi), _p(out))
    return out


def microfacet_sample(ggx, alpha_u, alpha_v, visible, wi, sample2):
    wi, s = _f(wi).reshape(-1, 3), _f(sample2).reshape(-1, 2)
    m, pdf = np.zeros((wi.shape[0], 3), np.float32), np.zeros(wi.shape[0], np.float32)
    lib().mo_kat_microfacet_sample(int(ggx), alpha_u, alpha_v, int(visible), wi.shape[0], _p(wi), _p(s), _p(m), _p(pdf))
    return m, pdf


def bsdf_kat(plugin_dict, wi, wo, sample3):
    """-> dict(eval (N,3), pdf, s_wo (N,3), s_pdf, s_eta, s_delta, s_weight (N,3), s_valid) for the oracle's BSDF models"""
    d, n = bsdf_desc(plugin_dict)
    wi, wo, s = _f(wi).reshape(-1, 3), _f(wo).reshape(-1, 3), _f(sample3).reshape(-1, 3)
    out = np.zeros((wi.shape[0], 14), np.float32)
    if n["type"] in (8, 9):                                   # blendbsdf / mask with a constant weight
        kids = [bsdf_desc(None, c)[0] for c in n["children"]]
        lib().mo_kat_nested_bsdf(1 if n["type"] == 8 else 2, C.c_float(n["reflectance"][0]), int(n["twosided"]), C.byref(kids[0]),
                                 C.byref(kids[1]) if len(kids) > 1 else None, wi.shape[0], _p(wi), _p(wo), _p(s), _p(out))
    else:
        lib().mo_kat_bsdf(C.byref(d), wi.shape[0], _p(wi), _p(wo), _p(s), _p(out))
    return dict(eval=out[:, 0:3], pdf=out[:, 3], s_wo=out[:, 4:7], s_pdf=out[:, 7], s_eta=out[:, 8], s_delta=out[:, 9] > 0.5,
                s_weight=out[:, 10:13], s_valid=out[:, 13] > 0.5)


def gauss_legendre(n):
    nodes, weights = np.zeros(n, np.float32), np.zeros(n, np.float32)
    lib().mo_kat_gauss_legendre(n, _p(nodes), _p(weights))
    return nodes, weights


def roughplastic_tables(plugin_dict):
    d, _ = bsdf_desc(plugin_dict)
    out = np.zeros(65, np.float32)
    lib().mo_kat_roughplastic_tables(C.byref(d), _p(out))
    return out[:64], float(out[64])


def hier2d(data, which, points, normalize=True):
    """Hierarchical2D0: which in ('sample', 'invert', 'eval') -> (N, 3) = (x, y, pdf)"""
    data, pts = _f(data), _f(points).reshape(-1, 2)
    out = np.zeros((pts.shape[0], 3), np.float32)
    lib().mo_kat_hier2d(_p(data), data.shape[1], data.shape[0], int(normalize), {"sample": 0, "invert": 1, "eval": 2}[which], pts.shape[0], _p(pts), _p(out))
    return out


def bilinear_to_square(v00, v10, v01, v11, x, y):
    out = np.zeros(3, np.float32)
    lib().mo_kat_bilinear_to_square(v00, v10, v01, v11, x, y, _p(out))
    return out


def envmap_kat(rgb, sample2, scale=1.0, to_world=None):
    rgb, s = _f(rgb), _f(sample2).reshape(-1, 2)
    tw = _f(np.asarray(to_world, np.float32).reshape(4, 4)[:3, :3]) if to_world is not None else None
    out = np.zeros((s.shape[0], 11), np.float32)
    lib().mo_kat_envmap(rgb.shape[1], rgb.shape[0], _p(rgb), scale, _p(tw), s.shape[0], _p(s), _p(out))
    return dict(d=out[:, 0:3], pdf=out[:, 3], spec=out[:, 4:7], eval=out[:, 7:10], pdf_again=out[:, 10])
