"""ctypes binding of ``libmtsamd.so`` (the C ABI declared in ``include/mtsamd.h``).

The library is the product: there is no Python or CPU fallback.  If the shared object is
missing or a call fails, an exception is raised (``RuntimeError``, as the reference's pybind
layer maps ``std::runtime_error``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MTSAMD_LIB", os.path.join(_HERE, "libmtsamd.so"))   # override: kernel experiments only

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)
vp = C.c_void_p


class MeshDesc(C.Structure):
    _fields_ = [("vertex_count", C.c_uint32), ("face_count", C.c_uint32), ("positions", f32p), ("normals", f32p),
                ("texcoords", f32p), ("faces", u32p), ("bsdf", C.c_int32), ("emitter", C.c_int32)]


class BsdfDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("reflectance", C.c_float * 3), ("texture", C.c_int32), ("twosided", C.c_int32),
                ("specular_reflectance", C.c_float * 3), ("specular_transmittance", C.c_float * 3), ("eta", C.c_float * 3),
                ("k", C.c_float * 3), ("int_ior", C.c_float), ("ext_ior", C.c_float), ("alpha_u", C.c_float), ("alpha_v", C.c_float),
                ("distribution", C.c_int32), ("sample_visible", C.c_int32), ("nonlinear", C.c_int32), ("uniform_mask", C.c_int32),
                ("nested", C.c_int32 * 2)]


class EmitterDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("radiance", C.c_float * 3), ("envmap_data", f32p), ("envmap_width", C.c_int32),
                ("envmap_height", C.c_int32), ("envmap_scale", C.c_float), ("to_world", C.c_float * 16),
                ("cutoff_angle", C.c_float), ("beam_width", C.c_float)]


class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("data", f32p), ("kind", C.c_int32), ("color0", C.c_float * 3),
                ("color1", C.c_float * 3), ("to_uv", C.c_float * 6)]


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(MeshDesc)), ("mesh_count", C.c_uint32), ("bsdfs", C.POINTER(BsdfDesc)),
                ("bsdf_count", C.c_uint32), ("emitters", C.POINTER(EmitterDesc)), ("emitter_count", C.c_uint32),
                ("textures", C.POINTER(TextureDesc)), ("texture_count", C.c_uint32),
                ("spectral", C.c_int32), ("rgb2spec_path", C.c_char_p)]


class Rays(C.Structure):
    _fields_ = [(n, vp) for n in ("ox", "oy", "oz", "dx", "dy", "dz", "mint", "maxt", "active")]


class RenderDesc(C.Structure):
    _fields_ = [("to_world", C.c_float * 16), ("fov_x_deg", C.c_float), ("near_clip", C.c_float), ("far_clip", C.c_float),
                ("film_width", C.c_int32), ("film_height", C.c_int32), ("crop_x", C.c_int32), ("crop_y", C.c_int32),
                ("crop_width", C.c_int32), ("crop_height", C.c_int32), ("rfilter", C.c_int32), ("rfilter_param", C.c_float), ("rfilter_param2", C.c_float),
                ("rfilter_analytic", C.c_int32), ("sample_count", C.c_int32), ("seed", C.c_uint64), ("max_depth", C.c_int32),
                ("rr_depth", C.c_int32), ("row_begin", C.c_int32), ("row_end", C.c_int32), ("part_index", C.c_int32),
                ("part_count", C.c_int32), ("part_tile_rows", C.c_int32), ("paths_per_wave", C.c_int32),
                ("pipeline", C.c_int32), ("film_rgb", C.c_int32), ("integrator", C.c_int32), ("emitter_samples", C.c_int32),
                ("bsdf_samples", C.c_int32), ("hide_emitters", C.c_int32), ("moment", C.c_int32),
                ("aperture_radius", C.c_float), ("focus_distance", C.c_float),
                ("timeout", C.c_float), ("samples_per_pass", C.c_int32), ("profile", C.c_int32),
                ("max_pass_log2", C.c_int32), ("finish_kernel", C.c_int32)]


# every symbol include/mtsamd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mtsamd_abi_version": (C.c_int, []),
    "mtsamd_last_error": (C.c_char_p, []),
    "mtsamd_device_count": (C.c_int, []),
    "plugin_name": (C.c_char_p, []),
    "plugin_descr": (C.c_char_p, []),
    "mtsamd_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(vp)]),
    "mtsamd_scene_destroy": (None, [vp]),
    "mtsamd_scene_bbox": (C.c_int, [vp, f32p]),
    "mtsamd_scene_info": (C.c_int, [vp, u32p]),
    "mtsamd_scene_set_bsdf_reflectance": (C.c_int, [vp, C.c_uint32, f32p]),
    "mtsamd_scene_set_emitter_radiance": (C.c_int, [vp, C.c_uint32, f32p]),
    "mtsamd_scene_update_texture": (C.c_int, [vp, C.c_uint32, vp, vp]),
    "mtsamd_ray_intersect": (C.c_int, [vp, C.c_uint64, C.POINTER(Rays), vp, vp, vp, vp, vp, vp]),
    "mtsamd_ray_intersect_naive": (C.c_int, [vp, C.c_uint64, C.POINTER(Rays), vp, vp, vp, vp, vp, vp]),
    "mtsamd_ray_test": (C.c_int, [vp, C.c_uint64, C.POINTER(Rays), vp, vp]),
    "mtsamd_ray_intersect_si": (C.c_int, [vp, C.c_uint64, C.POINTER(Rays), vp, vp, vp, vp, vp]),
    "mtsamd_render": (C.c_int, [vp, C.POINTER(RenderDesc), vp, u64p, vp]),
    "mtsamd_cancel": (C.c_int, [vp]),
    "mtsamd_rgb2spec_build": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32]),
    "mtsamd_srgb_model_fetch": (C.c_int, [C.c_char_p, f32p, f32p]),
    "mtsamd_render_adjoint": (C.c_int, [vp, C.POINTER(RenderDesc), vp, vp, vp, vp, vp, vp]),
    "mtsamd_render_adjoint_envmap": (C.c_int, [vp, C.POINTER(RenderDesc), vp, vp, vp, vp]),
    "mtsamd_scene_update_envmap": (C.c_int, [vp, f32p, C.c_int32]),
    "mtsamd_scene_roughplastic_tables": (C.c_int, [vp, C.c_uint32, f32p]),
    "mtsamd_scene_texture_info": (C.c_int, [vp, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), u64p]),
    "mtsamd_sample_radiance": (C.c_int, [vp, C.POINTER(RenderDesc), C.c_uint64, C.c_uint64, vp, vp, vp]),
    "mtsamd_camera_sample_rays": (C.c_int, [C.POINTER(RenderDesc), C.c_uint64] + [vp] * 13),
    "mtsamd_imageblock_put": (C.c_int, [C.c_int32] * 6 + [C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_uint64, vp, vp, vp, vp]),
    "mtsamd_imageblock_put_block": (C.c_int, [vp] + [C.c_int32] * 5 + [vp] + [C.c_int32] * 6 + [vp]),
    "mtsamd_rfilter_info": (C.c_int, [C.c_int32, C.c_float, C.c_float, f32p, f32p, C.POINTER(C.c_int32)]),
    "mtsamd_film_develop": (C.c_int, [vp, C.c_uint64, vp, vp]),
    "mtsamd_libm_eval": (C.c_int, [C.c_int32, C.c_uint64, vp, vp, vp, vp]),
    "mtsamd_scene_set_bsdf_param": (C.c_int, [vp, C.c_uint32, C.c_int32, f32p]),
    "mtsamd_render_adjoint_param": (C.c_int, [vp, C.POINTER(RenderDesc), vp, vp, C.c_uint32, C.c_int32, C.c_int32, C.c_float, vp, vp]),
}

_lib = None


def lib():
    """Load libmtsamd.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmtsamd.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C mitsuba2_amd/csrc`).  There is no CPU fallback." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)      # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        if handle.mtsamd_abi_version() != 6:
            raise RuntimeError("libmtsamd.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc):
    """Turn a negative status into the exception the reference's bindings would raise."""
    if rc < 0:
        msg = lib().mtsamd_last_error()
        raise RuntimeError((msg or b"mtsamd error").decode("utf-8", "replace") + " (status %d)" % rc)
    return rc
