"""Host-side mirror of the reference's Python surface for the path-tracing hot path.

Class and method names follow the pybind11 bindings of Mitsuba 2 (``src/librender/python/scene_v.cpp:37-86``,
``integrator_v.cpp:61-170``, ``imageblock_v.cpp:5-40``, ``src/films/hdrfilm.cpp``) for the supported subset;
every compute call goes through the C ABI of ``libmtsamd.so`` (``include/mtsamd.h``).  PyTorch is used only
to own device memory and streams.
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import _lib as L

RayEpsilon = float(np.float32(np.finfo(np.float32).eps / 2 * 1500))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# --------------------------------------------------------------------------------------------
# records (include/mitsuba/core/ray.h:21-62, include/mitsuba/render/interaction.h:33-126)
@dataclass
class Ray3f:
    o: torch.Tensor                     # (N,3)
    d: torch.Tensor                     # (N,3)
    mint: Optional[torch.Tensor] = None  # (N,), default RayEpsilon (ray.h:33)
    maxt: Optional[torch.Tensor] = None  # (N,), default +inf (ray.h:34)
    time: float = 0.0

    def __post_init__(self):
        n = self.o.shape[0]
        dev = self.o.device
        if self.mint is None:
            self.mint = torch.full((n,), RayEpsilon, dtype=torch.float32, device=dev)
        if self.maxt is None:
            self.maxt = torch.full((n,), float("inf"), dtype=torch.float32, device=dev)


@dataclass
class SurfaceInteraction3f:
    t: torch.Tensor
    prim_index: torch.Tensor
    shape_index: torch.Tensor
    p: Optional[torch.Tensor] = None
    n: Optional[torch.Tensor] = None
    uv: Optional[torch.Tensor] = None
    sh_frame_s: Optional[torch.Tensor] = None
    sh_frame_t: Optional[torch.Tensor] = None
    sh_frame_n: Optional[torch.Tensor] = None
    dp_du: Optional[torch.Tensor] = None
    dp_dv: Optional[torch.Tensor] = None
    wi: Optional[torch.Tensor] = None
    prim_uv: Optional[torch.Tensor] = None   # barycentric (u,v): the kd-tree "cache" (kdtree.h:2432-2452)

    def is_valid(self):
        """interaction.h:53-55"""
        return self.t != float("inf")


# --------------------------------------------------------------------------------------------
class ReconstructionFilter:
    """include/mitsuba/core/rfilter.h; discretisation from src/libcore/rfilter.cpp:9-20 via the C ABI."""
    kind = -1

    def __init__(self, param=0.0, param2=0.0):
        self.param, self.param2 = float(param), float(param2)
        table = (C.c_float * 32)()
        radius = C.c_float()
        border = C.c_int32()
        L.check(L.lib().mtsamd_rfilter_info(self.kind, self.param, self.param2, table, C.byref(radius), C.byref(border)))
        self._table = np.array(table, dtype=np.float32)
        self._radius = radius.value
        self._border = border.value

    def radius(self):
        return self._radius

    def border_size(self):
        return self._border

    def eval_discretized(self, x):
        idx = min(int(abs(np.float32(x) * np.float32(31.0 / self._radius))), 31)
        return float(self._table[idx])


class GaussianFilter(ReconstructionFilter):
    """src/rfilters/gaussian.cpp"""
    kind = 0

    def __init__(self, stddev=0.5):
        super().__init__(stddev)


class BoxFilter(ReconstructionFilter):
    """src/rfilters/box.cpp"""
    kind = 1

    def __init__(self, radius=0.5):
        super().__init__(radius)


class TentFilter(ReconstructionFilter):
    """src/rfilters/tent.cpp (radius 1: ImageBlock::put treats it like a one-pixel footprint, imageblock.cpp:117)"""
    kind = 2


class CatmullRomFilter(ReconstructionFilter):
    """src/rfilters/catmullrom.cpp"""
    kind = 3


class MitchellFilter(ReconstructionFilter):
    """src/rfilters/mitchell.cpp"""
    kind = 4

    def __init__(self, B=1.0 / 3.0, C=1.0 / 3.0):
        super().__init__(B, C)


class LanczosFilter(ReconstructionFilter):
    """src/rfilters/lanczos.cpp"""
    kind = 5

    def __init__(self, lobes=3):
        super().__init__(int(lobes))


def make_filter(name, *params):
    """reconstruction filter plugin by name: gaussian(stddev) | box(radius) | tent | catmullrom | mitchell(B, C) | lanczos(lobes)"""
    classes = {"gaussian": GaussianFilter, "box": BoxFilter, "tent": TentFilter, "catmullrom": CatmullRomFilter,
               "mitchell": MitchellFilter, "lanczos": LanczosFilter}
    if name not in classes:
        raise RuntimeError('Reconstruction filter "%s" is not supported by this backend (%s)' % (name, ", ".join(classes)))
    return classes[name](*params)


class ImageBlock:
    """src/librender/imageblock.cpp.  ``data()`` is a (H+2b, W+2b, C) float32 CUDA tensor."""

    def __init__(self, size, channel_count, filter=None, warn_negative=True, warn_invalid=True, border=True,
                 normalize=False, device="cuda"):
        if normalize:
            raise RuntimeError("ImageBlock: normalize=True is not supported by this backend")
        self._size = (int(size[0]), int(size[1]))
        self._offset = (0, 0)
        self._channels = int(channel_count)
        self._filter = filter
        self._border = filter.border_size() if (filter is not None and border) else 0
        self._device = torch.device(device)
        self._data = torch.zeros((self._size[1] + 2 * self._border, self._size[0] + 2 * self._border, self._channels),
                                 dtype=torch.float32, device=self._device)

    def size(self): return self._size
    def width(self): return self._size[0]
    def height(self): return self._size[1]
    def offset(self): return self._offset
    def set_offset(self, o): self._offset = (int(o[0]), int(o[1]))
    def channel_count(self): return self._channels
    def border_size(self): return self._border
    def data(self): return self._data
    def clear(self): self._data.zero_()

    def put(self, pos, values=None, active=None):
        """put(block) or put(pos, values): imageblock.cpp:49-77 / :80-172."""
        lib = L.lib()
        if isinstance(pos, ImageBlock):
            src = pos
            if src.channel_count() != self.channel_count():
                raise RuntimeError("ImageBlock::put(): mismatched channel counts!")
            L.check(lib.mtsamd_imageblock_put_block(_ptr(src._data), src._size[0], src._size[1], src._offset[0], src._offset[1],
                                                    src._border, _ptr(self._data), self._size[0], self._size[1], self._offset[0],
                                                    self._offset[1], self._border, self._channels, _stream()))
            return
        if self._filter is None:
            raise RuntimeError("ImageBlock::put(): a reconstruction filter is required")
        pos = torch.as_tensor(pos, dtype=torch.float32, device=self._device).reshape(-1, 2).contiguous()
        values = torch.as_tensor(values, dtype=torch.float32, device=self._device).reshape(-1, self._channels).contiguous()
        if active is not None:
            keep = torch.as_tensor(active, device=self._device).bool().reshape(-1)
            pos, values = pos[keep].contiguous(), values[keep].contiguous()
        f = self._filter
        L.check(lib.mtsamd_imageblock_put(self._size[0], self._size[1], self._offset[0], self._offset[1], self._channels, f.kind,
                                          f.param, f.param2, 0, self._border, pos.shape[0], _ptr(pos), _ptr(values), _ptr(self._data),
                                          _stream()))


# --------------------------------------------------------------------------------------------
class IndependentSampler:
    """src/samplers/independent.cpp (sample_count default 4, seed 0: src/librender/sampler.cpp:7-8)"""

    def __init__(self, sample_count=4, seed=0):
        self._sample_count = int(sample_count)
        self._seed = int(seed)

    def sample_count(self): return self._sample_count
    def seed_value(self): return self._seed


class HDRFilm:
    """src/films/hdrfilm.cpp + src/librender/film.cpp (defaults 768x576, gaussian filter)."""

    def __init__(self, width=768, height=576, crop_offset=None, crop_size=None, rfilter=None, file_format="openexr",
                 pixel_format="rgba", component_format="float16", high_quality_edges=False):
        self._size = (int(width), int(height))
        co = (0, 0) if crop_offset is None else (int(crop_offset[0]), int(crop_offset[1]))
        cs = self._size if crop_size is None else (int(crop_size[0]), int(crop_size[1]))
        self.set_crop_window(co, cs)
        self._filter = rfilter if rfilter is not None else GaussianFilter()
        self._storage = None
        self._dest_file = None
        self._hq_edges = bool(high_quality_edges)
        # hdrfilm.cpp:42-123: parameter validation and the per-format overrides
        ff, pf, cf = file_format.lower(), pixel_format.lower(), component_format.lower()
        if ff in ("openexr", "exr"):
            ff = "exr"
        elif ff not in ("rgbe", "pfm"):
            raise RuntimeError('The "file_format" parameter must either be equal to "openexr", "pfm", or "rgbe", found %s instead.' % ff)
        if pf not in ("luminance", "luminance_alpha", "rgb", "rgba", "xyz", "xyza"):
            raise RuntimeError('The "pixel_format" parameter must either be equal to "luminance", "luminance_alpha", "rgb", "rgba", '
                               '"xyz", "xyza". Found %s.' % pf)
        if cf not in ("float16", "float32", "uint32"):
            raise RuntimeError('The "component_format" parameter must either be equal to "float16", "float32", or "uint32". Found %s instead.' % cf)
        if ff == "rgbe":
            pf, cf = "rgb", "float32"
        elif ff == "pfm":
            pf, cf = (pf if pf in ("rgb", "luminance") else "rgb"), "float32"
        self._file_format, self._pixel_format, self._component_format = ff, pf, cf

    def size(self): return self._size
    def crop_size(self): return self._crop_size
    def crop_offset(self): return self._crop_offset
    def reconstruction_filter(self): return self._filter
    def has_high_quality_edges(self): return self._hq_edges

    def set_destination_file(self, filename):
        """hdrfilm.cpp:205-209"""
        self._dest_file = str(filename)

    def develop(self):
        """hdrfilm.cpp:322-342: convert the storage to pixel_format / component_format and write it to the destination
        file (the extension is replaced by the file format's)."""
        import os
        from . import bitmap as B
        if not self._dest_file:
            raise RuntimeError("Destination file not specified, cannot develop.")
        ext = {"exr": ".exr", "rgbe": ".rgbe", "pfm": ".pfm"}[self._file_format]
        root, cur = os.path.splitext(self._dest_file)
        filename = self._dest_file if cur.lower() == ext else root + ext
        raw = self.bitmap(raw=True)
        if raw.shape[2] != 5:
            raise RuntimeError("HDRFilm::develop(): only the X, Y, Z, A, W storage layout can be written")
        if self._pixel_format in ("rgb", "rgba"):
            px = self.bitmap().cpu().numpy()
            px = px if self._pixel_format == "rgba" else px[..., :3]
            names = "RGBA"[:px.shape[2]]
        else:                           # Bitmap::convert from XYZAW: divide by the weight, keep XYZ / take Y
            r = raw.cpu().numpy()
            w = r[..., 4:5]
            inv = np.where(w != 0, 1.0 / np.where(w != 0, w, 1), 0).astype(np.float32)
            xyz, a = r[..., :3] * inv, r[..., 3:4] * inv
            px = {"xyz": xyz, "xyza": np.concatenate([xyz, a], 2), "luminance": xyz[..., 1:2],
                  "luminance_alpha": np.concatenate([xyz[..., 1:2], a], 2)}[self._pixel_format]
            names = {"xyz": "XYZ", "xyza": "XYZA", "luminance": "Y", "luminance_alpha": "YA"}[self._pixel_format]
        if self._file_format == "pfm":
            B.write_pfm(filename, px)
        elif self._file_format == "rgbe":
            B.write_rgbe(filename, px)
        else:
            dt = {"float16": np.float16, "float32": np.float32, "uint32": np.uint32}[self._component_format]
            B.write_exr(filename, {n: np.ascontiguousarray(px[..., i]).astype(dt) for i, n in enumerate(names)})
        return filename

    def set_crop_window(self, crop_offset, crop_size):
        """film.cpp:55-64"""
        if (crop_offset[0] < 0 or crop_offset[1] < 0 or crop_size[0] <= 0 or crop_size[1] <= 0 or
                crop_offset[0] + crop_size[0] > self._size[0] or crop_offset[1] + crop_size[1] > self._size[1]):
            raise RuntimeError("Invalid crop window specification!")
        self._crop_offset = (int(crop_offset[0]), int(crop_offset[1]))
        self._crop_size = (int(crop_size[0]), int(crop_size[1]))

    def prepare(self, channels=("X", "Y", "Z", "A", "W"), device="cuda"):
        """hdrfilm.cpp:188-203: storage ImageBlock(crop_size, n_channels), no filter / border."""
        if len(set(channels)) != len(channels):
            raise RuntimeError("Film::prepare(): duplicate channel name")
        self._storage = ImageBlock(self._crop_size, len(channels), device=device)
        self._storage.set_offset(self._crop_offset)

    def put(self, block):
        self._storage.put(block)

    def bitmap(self, raw=False):
        """hdrfilm.cpp:249-320: raw=True -> XYZAW storage, else RGBA float32 (H, W, 4)."""
        if self._storage is None:
            raise RuntimeError("HDRFilm::bitmap(): no storage (render first)")
        data = self._storage.data()
        if raw:
            return data
        out = torch.empty((data.shape[0], data.shape[1], 4), dtype=torch.float32, device=data.device)
        L.check(L.lib().mtsamd_film_develop(_ptr(data), data.shape[0] * data.shape[1], _ptr(out), _stream()))
        return out


LIBM_FUNCTIONS = ("sin", "cos", "tan", "exp", "log", "erf", "acos", "atan2", "atanh", "cosh")


def libm_eval(name, x, y=None):
    """The kernels' own elementary functions (csrc/device_libm.h) on a CUDA tensor of float32 arguments -- the role enoki::sin / exp /
    erf ... play in the reference (include/mitsuba/core/warp.h:54-90, render/microfacet.h:187-493).  `atan2` takes (y, x)."""
    fn = LIBM_FUNCTIONS.index(name)
    x = x.contiguous().float()
    if fn == 7:
        if y is None:
            raise RuntimeError("atan2 takes two arguments")
        y = y.contiguous().float()
    out = torch.empty_like(x)
    L.check(L.lib().mtsamd_libm_eval(fn, x.numel(), _ptr(x), _ptr(y) if fn == 7 else None, _ptr(out), _stream()))
    return out


def parse_fov(fov=None, focal_length=None, fov_axis="x", aspect=1.0):
    """src/librender/sensor.cpp:119-169"""
    if fov is not None and focal_length is not None:
        raise RuntimeError("Please specify either a focal length ('focal_length') or a field of view ('fov')!")
    f32 = np.float32
    if fov is not None:
        fov = f32(fov)
        fov_axis = fov_axis.lower()
        if fov_axis == "smaller":
            fov_axis = "y" if aspect > 1 else "x"
        elif fov_axis == "larger":
            fov_axis = "x" if aspect > 1 else "y"
    else:
        f = "50mm" if focal_length is None else str(focal_length)
        if f.endswith("mm"):
            f = f[:-2]
        try:
            value = f32(float(f))
        except ValueError:
            raise RuntimeError("Could not parse the focal length (must be of the form <x>mm, where <x> is a positive integer)!")
        fov = f32(2.0) * f32(np.degrees(np.arctan(f32(np.sqrt(f32(36 * 36 + 24 * 24))) / (f32(2.0) * value))))
        fov_axis = "diagonal"
    if fov_axis == "x":
        result = fov
    elif fov_axis == "y":
        result = f32(np.degrees(f32(2.0) * np.arctan(np.tan(f32(0.5) * f32(np.radians(fov))) * f32(aspect))))
    elif fov_axis == "diagonal":
        diagonal = f32(2.0) * np.tan(f32(0.5) * f32(np.radians(fov)))
        width = diagonal / f32(np.sqrt(f32(1.0) + f32(1.0) / f32(aspect * aspect)))
        result = f32(np.degrees(f32(2.0) * np.arctan(width * f32(0.5))))
    else:
        raise RuntimeError("The 'fov_axis' parameter must be set to one of 'smaller', 'larger', 'diagonal', 'x', or 'y'!")
    if result <= 0.0 or result >= 180.0:
        raise RuntimeError("The horizontal field of view must be in the range [0, 180]!")
    return float(result)


class PerspectiveCamera:
    """src/sensors/perspective.cpp"""

    def __init__(self, to_world=None, fov=None, focal_length=None, fov_axis="x", near_clip=1e-2, far_clip=1e4, film=None,
                 sampler=None):
        self._film = film if film is not None else HDRFilm()
        self._sampler = sampler if sampler is not None else IndependentSampler()
        self._to_world = np.eye(4, dtype=np.float32) if to_world is None else _f32(to_world).reshape(4, 4)
        if near_clip <= 0:
            raise RuntimeError("The 'near_clip' parameter must be greater than zero!")
        if near_clip >= far_clip:
            raise RuntimeError("The 'near_clip' parameter must be smaller than 'far_clip'.")
        self._near, self._far = float(near_clip), float(far_clip)
        w, h = self._film.size()
        self._x_fov = parse_fov(fov, focal_length, fov_axis, w / h)

    def film(self): return self._film
    def sampler(self): return self._sampler
    def x_fov(self): return self._x_fov
    def near_clip(self): return self._near
    def far_clip(self): return self._far
    def world_transform(self): return self._to_world

    def _fill_desc(self, d):
        d.to_world = (C.c_float * 16)(*self._to_world.reshape(-1).tolist())
        d.fov_x_deg = self._x_fov
        d.near_clip, d.far_clip = self._near, self._far
        f = self._film
        d.film_width, d.film_height = f.size()
        d.crop_x, d.crop_y = f.crop_offset()
        d.crop_width, d.crop_height = f.crop_size()
        d.rfilter = f.reconstruction_filter().kind
        d.rfilter_param = f.reconstruction_filter().param
        d.rfilter_param2 = f.reconstruction_filter().param2
        d.rfilter_analytic = 0
        d.sample_count = self._sampler.sample_count()
        d.seed = self._sampler.seed_value()
        d.aperture_radius, d.focus_distance = 0.0, 0.0

    def needs_aperture_sample(self):
        return False

    def sample_ray(self, position_sample, aperture_sample=None):
        """perspective.cpp:153-188 / thinlens.cpp:175-214 for (N,2) film-plane samples in [0,1)^2 (and (N,2) aperture samples)
        -> Ray3f."""
        ps = torch.as_tensor(position_sample, dtype=torch.float32, device="cuda").reshape(-1, 2)
        n = ps.shape[0]
        sx, sy = ps[:, 0].contiguous(), ps[:, 1].contiguous()
        apx = apy = None
        if aperture_sample is not None:
            ap = torch.as_tensor(aperture_sample, dtype=torch.float32, device="cuda").reshape(-1, 2)
            if ap.shape[0] != n:
                raise RuntimeError("aperture_sample must have one entry per position sample")
            apx, apy = ap[:, 0].contiguous(), ap[:, 1].contiguous()
        out = torch.empty((8, n), dtype=torch.float32, device="cuda")
        d = L.RenderDesc()
        self._fill_desc(d)
        d.max_depth, d.rr_depth = -1, 5
        L.check(L.lib().mtsamd_camera_sample_rays(C.byref(d), n, _ptr(sx), _ptr(sy), _ptr(apx) if apx is not None else None,
                                                  _ptr(apy) if apy is not None else None, *[_ptr(out[k]) for k in range(8)], _stream()))
        return Ray3f(o=out[0:3].t().contiguous(), d=out[3:6].t().contiguous(), mint=out[6].clone(), maxt=out[7].clone())


class ThinLensCamera(PerspectiveCamera):
    """src/sensors/thinlens.cpp: perspective camera with a circular aperture focused at `focus_distance`"""

    def __init__(self, aperture_radius=None, focus_distance=None, **kwargs):
        super().__init__(**kwargs)
        if aperture_radius is None:
            raise RuntimeError('Property "aperture_radius" has not been specified!')       # props.float_("aperture_radius"), thinlens.cpp:112
        self._aperture_radius = float(aperture_radius)
        if self._aperture_radius == 0.0:              # thinlens.cpp:114-117
            self._aperture_radius = float(np.finfo(np.float32).eps) / 2
        if self._aperture_radius < 0.0:
            raise RuntimeError("The 'aperture_radius' parameter must not be negative")
        self._focus_distance = float(focus_distance) if focus_distance is not None else self._far      # sensor.cpp:104

    def aperture_radius(self): return self._aperture_radius
    def focus_distance(self): return self._focus_distance

    def needs_aperture_sample(self):
        return True

    def _fill_desc(self, d):
        super()._fill_desc(d)
        d.aperture_radius, d.focus_distance = self._aperture_radius, self._focus_distance


def srgb_coeff_path(build=True):
    """The RGB -> spectrum coefficient table ('data/srgb.coeff' of the reference, src/librender/srgb.cpp:24-27): generated
    on first use with mtsamd_rgb2spec_build at the reference's resolution 64 (build artefact, not tracked)."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "srgb.coeff")
    if not os.path.exists(path) and build:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        L.check(L.lib().mtsamd_rgb2spec_build(path.encode(), 64, min(os.cpu_count() or 1, 16)))
    return path


# --------------------------------------------------------------------------------------------
class Scene:
    """src/librender/scene.cpp: shapes + BSDFs + emitters uploaded to one GPU, BVH built by the library."""

    def __init__(self, scene_dict, device=0, sensor=None, integrator=None, variant="rgb"):
        lib = L.lib()
        if variant not in ("rgb", "spectral"):
            raise RuntimeError("unsupported variant '%s' (rgb or spectral)" % variant)
        self._variant = variant
        if not torch.cuda.is_available():
            raise RuntimeError("mitsuba2_amd requires a HIP device (torch.cuda.is_available() is False)")
        self._device_index = int(device)
        self._dict = scene_dict
        self._sensors = [sensor] if sensor is not None else []
        self._integrator = integrator
        meshes, bsdfs, emitters = scene_dict["meshes"], scene_dict["bsdfs"], scene_dict.get("emitters", [])
        keep = []
        md = (L.MeshDesc * len(meshes))()
        for i, m in enumerate(meshes):
            pos, faces = _f32(m["positions"]).reshape(-1, 3), np.ascontiguousarray(m["faces"], dtype=np.uint32).reshape(-1, 3)
            nrm = _f32(m["normals"]).reshape(-1, 3) if m.get("normals") is not None else None
            uv = _f32(m["texcoords"]).reshape(-1, 2) if m.get("texcoords") is not None else None
            keep += [pos, faces, nrm, uv]
            md[i].vertex_count, md[i].face_count = pos.shape[0], faces.shape[0]
            md[i].positions = pos.ctypes.data_as(L.f32p)
            md[i].faces = faces.ctypes.data_as(L.u32p)
            md[i].normals = nrm.ctypes.data_as(L.f32p) if nrm is not None else None
            md[i].texcoords = uv.ctypes.data_as(L.f32p) if uv is not None else None
            md[i].bsdf, md[i].emitter = int(m["bsdf"]), int(m.get("emitter", -1))
        tex = []                       # bitmap textures (src/textures/bitmap.cpp): reflectance = dict(type="bitmap", data=(H,W,3))
        self._bsdf_texture = {}
        from . import bsdfs as B
        self._bsdf_records = [B.normalize(b) for b in bsdfs]      # plugin defaults / validation (src/bsdfs/*.cpp constructors)
        flat = B.flatten(self._bsdf_records)                      # + the children of blendbsdf / mask records
        bd = (L.BsdfDesc * max(len(flat), 1))()
        for i, n in enumerate(flat):
            bd[i].type, bd[i].twosided = n["type"], int(n["twosided"])
            bd[i].nested = (C.c_int32 * 2)(*n.get("nested", [-1, -1]))
            refl = n["reflectance"]
            if isinstance(refl, dict):
                kind = refl.get("type")
                if kind not in ("bitmap", "checkerboard"):
                    raise RuntimeError("Texture plugin '%s' is not supported by this backend (bitmap, checkerboard)" % kind)
                if kind == "bitmap":
                    data = _f32(refl["data"])
                    if data.ndim != 3 or data.shape[2] != 3:
                        raise RuntimeError("bitmap texture: expected (H, W, 3) linear RGB data")
                else:
                    data = None
                bd[i].reflectance = (C.c_float * 3)(0.5, 0.5, 0.5)
                bd[i].texture = len(tex)
                self._bsdf_texture[i] = len(tex)
                tex.append((kind, data, refl))
            else:
                bd[i].reflectance = (C.c_float * 3)(*[float(x) for x in refl])
                bd[i].texture = -1
            for name in ("specular_reflectance", "specular_transmittance", "eta", "k"):
                setattr(bd[i], name, (C.c_float * 3)(*n[name]))
            bd[i].int_ior, bd[i].ext_ior, bd[i].alpha_u, bd[i].alpha_v = n["int_ior"], n["ext_ior"], n["alpha_u"], n["alpha_v"]
            bd[i].distribution, bd[i].sample_visible, bd[i].nonlinear = n["distribution"], int(n["sample_visible"]), int(n["nonlinear"])
            bd[i].uniform_mask = n["uniform_mask"]
        td = (L.TextureDesc * max(len(tex), 1))()
        for i, (kind, t, spec) in enumerate(tex):
            if kind == "bitmap":
                td[i].kind, td[i].width, td[i].height = 0, t.shape[1], t.shape[0]
                td[i].data = t.ctypes.data_as(L.f32p)
            else:                       # src/textures/checkerboard.cpp: color0 = .4, color1 = .2 by default
                td[i].kind = 1
                td[i].color0 = (C.c_float * 3)(*B._rgb(spec.get("color0"), 0.4))
                td[i].color1 = (C.c_float * 3)(*B._rgb(spec.get("color1"), 0.2))
            if spec.get("to_uv") is not None:       # Transform4f::extract(): the upper-left 3x3 acts on (u, v, 1)
                m = _f32(spec["to_uv"]).reshape(4, 4)
                td[i].to_uv = (C.c_float * 6)(float(m[0, 0]), float(m[0, 1]), float(m[0, 2]), float(m[1, 0]), float(m[1, 1]), float(m[1, 2]))
        self._texture_shapes = [t.shape if t is not None else (0, 0, 3) for (_, t, _) in tex]
        ed = (L.EmitterDesc * max(len(emitters), 1))()
        from . import emitters as E
        for i, e in enumerate(emitters):
            n = E.normalize(e)             # plugin defaults / validation (src/emitters/*.cpp constructors)
            ed[i].type = n["type"]
            ed[i].to_world = (C.c_float * 16)(*n["to_world"].reshape(-1).tolist())
            ed[i].radiance = (C.c_float * 3)(*n["radiance"])
            ed[i].cutoff_angle, ed[i].beam_width = n["cutoff_angle"], n["beam_width"]
            if n["type"] == E.TYPE_IDS["envmap"]:          # src/emitters/envmap.cpp: lat-long image (linear RGB), scale, to_world
                img = _f32(n["data"])
                if img.ndim != 3 or img.shape[2] != 3:
                    raise RuntimeError("envmap: expected (H, W, 3) linear RGB data")
                keep.append(img)
                ed[i].envmap_data = img.ctypes.data_as(L.f32p)
                ed[i].envmap_height, ed[i].envmap_width = img.shape[0], img.shape[1]
                ed[i].envmap_scale = n["scale"]
        sd = L.SceneDesc(md, len(meshes), bd, len(flat), ed, len(emitters), td, len(tex), 0, None)
        if variant == "spectral":
            sd.spectral = 1
            sd.rgb2spec_path = srgb_coeff_path().encode()
        handle = C.c_void_p()
        L.check(lib.mtsamd_scene_create(C.byref(sd), self._device_index, C.byref(handle)))
        self._handle = handle
        self._shape_count = len(meshes)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                L.lib().mtsamd_scene_destroy(h)
            except Exception:
                pass
            self._handle = None

    # -- accessors (scene_v.cpp:37-86)
    def sensors(self): return self._sensors
    def integrator(self): return self._integrator
    def shape_count(self): return self._shape_count

    def bbox(self):
        out = (C.c_float * 6)()
        L.check(L.lib().mtsamd_scene_bbox(self._handle, out))
        return np.array(out[:3], dtype=np.float32), np.array(out[3:], dtype=np.float32)

    def info(self):
        out = (C.c_uint32 * 6)()
        L.check(L.lib().mtsamd_scene_info(self._handle, out))
        return dict(zip(("primitives", "bvh_nodes", "bvh_depth", "shapes", "emitters", "lds_nodes"), [int(x) for x in out]))

    def set_bsdf_reflectance(self, index, rgb):
        L.check(L.lib().mtsamd_scene_set_bsdf_reflectance(self._handle, int(index), (C.c_float * 3)(*[float(x) for x in rgb])))

    def texture_index(self, bsdf):
        """Index of the bitmap texture attached to BSDF `bsdf` (None if its reflectance is constant)."""
        return self._bsdf_texture.get(int(bsdf))

    def update_texture(self, texture, data):
        """parameters_changed() for a BitmapTexture's `data` (bitmap.cpp:295-299); data: (H,W,3) tensor or array."""
        shape = self._texture_shapes[int(texture)]
        if isinstance(data, torch.Tensor):
            t = data.detach().to(torch.device("cuda", self._device_index), torch.float32).contiguous()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError("texture data has shape %s, expected %s" % (tuple(t.shape), tuple(shape)))
            # device-to-device copy enqueued on the current stream: stream-ordered with whatever wrote `t` and with the next render; a
            # temporary `t` is safe too (the caching allocator reuses its memory on this stream only after the copy)
            L.check(L.lib().mtsamd_scene_update_texture(self._handle, int(texture), _ptr(t), _stream()))
        else:
            a = _f32(data)
            if tuple(a.shape) != tuple(shape):
                raise RuntimeError("texture data has shape %s, expected %s" % (tuple(a.shape), tuple(shape)))
            L.check(L.lib().mtsamd_scene_update_texture(self._handle, int(texture), a.ctypes.data_as(C.c_void_p), _stream()))
            torch.cuda.current_stream().synchronize()

    def update_envmap(self, data, rebuild_distribution=True):
        """parameters_changed() for the envmap emitter's `data` (envmap.cpp:220-253); data: (H, W, 3) linear RGB tensor or array.
        ``rebuild_distribution=False`` keeps the importance-sampling hierarchy of the previous texels."""
        if isinstance(data, torch.Tensor):
            data = data.detach().cpu().numpy()
        a = _f32(data)
        L.check(L.lib().mtsamd_scene_update_envmap(self._handle, a.ctypes.data_as(L.f32p), 1 if rebuild_distribution else 0))
        # the scene description follows the device: a ParameterMap built after an optimisation step starts from the new texels
        for em in self._dict.get("emitters", []):
            if em.get("type", "area") == "envmap":
                em["data"] = a.reshape(np.asarray(em["data"]).shape).copy()

    def set_bsdf_param(self, index, kind, values):
        """parameters_changed() after editing a constant parameter of a BSDF record (mtsamd_bsdf_param kinds: 0 (diffuse_)reflectance,
        1 specular_reflectance, 2 eta, 3 k, 4 alpha, 5 specular_transmittance)"""
        v = [float(x) for x in values] + [0.0, 0.0]
        L.check(L.lib().mtsamd_scene_set_bsdf_param(self._handle, int(index), int(kind), (C.c_float * 3)(*v[:3])))

    def set_emitter_radiance(self, index, rgb):
        L.check(L.lib().mtsamd_scene_set_emitter_radiance(self._handle, int(index), (C.c_float * 3)(*[float(x) for x in rgb])))

    # -- queries
    def _soa(self, ray, active):
        dev = torch.device("cuda", self._device_index)
        o = ray.o.to(dev, torch.float32).t().contiguous()
        d = ray.d.to(dev, torch.float32).t().contiguous()
        mint = ray.mint.to(dev, torch.float32).contiguous()
        maxt = ray.maxt.to(dev, torch.float32).contiguous()
        act = None
        if active is not None and active is not True:
            act = torch.as_tensor(active, device=dev).to(torch.uint8).contiguous()
        r = L.Rays(_ptr(o[0]), _ptr(o[1]), _ptr(o[2]), _ptr(d[0]), _ptr(d[1]), _ptr(d[2]), _ptr(mint), _ptr(maxt), _ptr(act))
        return r, (o, d, mint, maxt, act), o.shape[1], dev

    def ray_intersect(self, ray, active=True, full=True):
        """Scene::ray_intersect (scene.h:36).  full=False skips the SurfaceInteraction fill."""
        r, keep, n, dev = self._soa(ray, active)
        t = torch.empty(n, dtype=torch.float32, device=dev)
        prim = torch.empty(n, dtype=torch.int32, device=dev)
        shape = torch.empty(n, dtype=torch.int32, device=dev)
        if not full:
            u = torch.empty(n, dtype=torch.float32, device=dev)
            v = torch.empty(n, dtype=torch.float32, device=dev)
            L.check(L.lib().mtsamd_ray_intersect(self._handle, n, C.byref(r), _ptr(t), _ptr(prim), _ptr(shape), _ptr(u), _ptr(v), _stream()))
            return SurfaceInteraction3f(t=t, prim_index=prim, shape_index=shape, prim_uv=torch.stack([u, v], dim=1))
        si = torch.empty((26, n), dtype=torch.float32, device=dev)
        L.check(L.lib().mtsamd_ray_intersect_si(self._handle, n, C.byref(r), _ptr(t), _ptr(prim), _ptr(shape), _ptr(si), _stream()))
        g = lambda a, b: si[a:b].t().contiguous()
        return SurfaceInteraction3f(t=t, prim_index=prim, shape_index=shape, p=g(0, 3), n=g(3, 6), uv=g(6, 8), sh_frame_s=g(8, 11),
                                    sh_frame_t=g(11, 14), sh_frame_n=g(14, 17), dp_du=g(17, 20), dp_dv=g(20, 23), wi=g(23, 26))

    def ray_intersect_naive(self, ray, active=True):
        """Scene::ray_intersect_naive (scene.h:38-44): brute force, for tests."""
        r, keep, n, dev = self._soa(ray, active)
        t = torch.empty(n, dtype=torch.float32, device=dev)
        prim = torch.empty(n, dtype=torch.int32, device=dev)
        shape = torch.empty(n, dtype=torch.int32, device=dev)
        u = torch.empty(n, dtype=torch.float32, device=dev)
        v = torch.empty(n, dtype=torch.float32, device=dev)
        L.check(L.lib().mtsamd_ray_intersect_naive(self._handle, n, C.byref(r), _ptr(t), _ptr(prim), _ptr(shape), _ptr(u), _ptr(v), _stream()))
        return SurfaceInteraction3f(t=t, prim_index=prim, shape_index=shape, prim_uv=torch.stack([u, v], dim=1))

    def ray_test(self, ray, active=True):
        """Scene::ray_test (scene.h:62)"""
        r, keep, n, dev = self._soa(ray, active)
        hit = torch.empty(n, dtype=torch.uint8, device=dev)
        L.check(L.lib().mtsamd_ray_test(self._handle, n, C.byref(r), _ptr(hit), _stream()))
        return hit.bool()


# --------------------------------------------------------------------------------------------
class PathIntegrator:
    """src/integrators/path.cpp + MonteCarloIntegrator (src/librender/integrator.cpp:283-296)."""

    def __init__(self, max_depth=-1, rr_depth=5, paths_per_wave=0, pipeline=0, samples_per_pass=-1, timeout=-1.0, profile=False):
        if max_depth < 0 and max_depth != -1:
            raise RuntimeError("\"max_depth\" must be set to -1 (infinite) or a value >= 0")
        if rr_depth <= 0:
            raise RuntimeError("\"rr_depth\" must be set to a value greater than zero!")
        self.max_depth, self.rr_depth = int(max_depth), int(rr_depth)
        self.paths_per_wave = int(paths_per_wave)
        self.pipeline = int(pipeline)          # 0 automatic, 1 fused kernel, 2 split trace/shade kernels (same samples)
        # SamplingIntegrator properties (integrator.cpp:27-39): samples_per_pass (-1: all), timeout in seconds (-1: none)
        self.samples_per_pass, self.timeout = int(samples_per_pass), float(timeout)
        self.profile = bool(profile)           # per-launch HIP event timing of the split pipeline (stats: trace_*_ns)
        # scheduler knobs of mtsamd_render_desc (0 = library default; the image does not depend on them): a pass holds at most
        # 2^max_pass_log2 samples; finish_kernel 1 = never end a pass with k_finish, 2 = as soon as the sample cursors are dry
        self.max_pass_log2, self.finish_kernel = 0, 0
        self._scene = None
        self.stats = None

    def _desc(self, sensor, rows=None, partition=None):
        d = L.RenderDesc()
        sensor._fill_desc(d)
        d.max_depth, d.rr_depth = self.max_depth, self.rr_depth
        d.row_begin, d.row_end = (0, 0) if rows is None else (int(rows[0]), int(rows[1]))
        if partition is not None:       # (index, count, tile_rows): interleaved row tiles of the film
            d.part_index, d.part_count, d.part_tile_rows = (int(x) for x in partition)
        d.paths_per_wave = self.paths_per_wave
        d.pipeline = self.pipeline
        d.samples_per_pass, d.timeout, d.profile = self.samples_per_pass, self.timeout, int(self.profile)
        d.max_pass_log2, d.finish_kernel = int(self.max_pass_log2), int(self.finish_kernel)
        self._fill_integrator(d)
        return d

    def _fill_integrator(self, d):
        d.integrator = 0

    def aov_names(self):
        """SamplingIntegrator::aov_names (integrator.h:128-133)"""
        return []

    def aov_channels(self):
        """film channels of a render: X, Y, Z, A, W followed by the AOVs (integrator.cpp:72-77)"""
        return ["X", "Y", "Z", "A", "W"] + self.aov_names()

    def render(self, scene, sensor=None, rows=None, partition=None):
        """Integrator::render (integrator.h:42): renders into sensor.film(); returns False if cancelled.
        rows=(begin, end) / partition=(index, count, tile_rows) restrict the call to a part of the film
        (multi-GPU film partition); the film then holds that part's contribution only."""
        sensor = sensor if sensor is not None else scene.sensors()[0]
        film = sensor.film()
        film.prepare(self.aov_channels(), device="cuda:%d" % scene._device_index)
        d = self._desc(sensor, rows, partition)
        stats = (C.c_uint64 * 16)()
        self._scene = scene
        rc = L.lib().mtsamd_render(scene._handle, C.byref(d), _ptr(film._storage.data()), stats, _stream())
        self._scene = None
        if rc == -4:                     # MTSAMD_ERR_CANCELLED: render() returns false (integrator.cpp:175)
            return False
        L.check(rc)
        self.stats = dict(zip(("closest_hit_rays", "any_hit_rays", "samples", "iterations", "segments", "bounce_ns", "film_ns", "tri_tests",
                               "trace_closest_ns", "trace_closest_launches", "trace_any_ns", "trace_any_launches", "shade_ns", "shade_launches",
                               "passes", "timed_out"), [int(x) for x in stats]))
        return True

    def cancel(self):
        if self._scene is not None:
            L.lib().mtsamd_cancel(self._scene._handle)

    def sample(self, scene, sensor, first, count):
        """SamplingIntegrator::sample for whole sample indices: returns (rgb (N,3), mask (N,), position (N,2))."""
        d = self._desc(sensor)
        dev = torch.device("cuda", scene._device_index)
        rgba = torch.empty((count, 4), dtype=torch.float32, device=dev)
        pos = torch.empty((count, 2), dtype=torch.float32, device=dev)
        L.check(L.lib().mtsamd_sample_radiance(scene._handle, C.byref(d), int(first), int(count), _ptr(rgba), _ptr(pos), _stream()))
        return rgba[:, :3], rgba[:, 3] > 0.5, pos


class DirectIntegrator(PathIntegrator):
    """src/integrators/direct.cpp: direct illumination with multiple importance sampling of emitter and BSDF samples."""

    def __init__(self, shading_samples=None, emitter_samples=None, bsdf_samples=None, hide_emitters=False, paths_per_wave=0):
        super().__init__(paths_per_wave=paths_per_wave)
        if shading_samples is not None and (emitter_samples is not None or bsdf_samples is not None):      # direct.cpp:80-86
            raise RuntimeError("Cannot specify both 'shading_samples' and ('emitter_samples' and/or 'bsdf_samples').")
        base = 1 if shading_samples is None else int(shading_samples)
        self.emitter_samples = base if emitter_samples is None else int(emitter_samples)
        self.bsdf_samples = base if bsdf_samples is None else int(bsdf_samples)
        if self.emitter_samples < 0 or self.bsdf_samples < 0 or self.emitter_samples + self.bsdf_samples == 0:
            raise RuntimeError("Must have at least 1 BSDF or emitter sample!")
        self.hide_emitters = bool(hide_emitters)

    def _fill_integrator(self, d):
        d.integrator = 1
        d.emitter_samples, d.bsdf_samples, d.hide_emitters = self.emitter_samples, self.bsdf_samples, int(self.hide_emitters)


class DepthIntegrator(PathIntegrator):
    """src/integrators/depth.cpp: distance to the first intersection"""

    def _fill_integrator(self, d):
        d.integrator = 2


class MomentIntegrator(PathIntegrator):
    """src/integrators/moment.cpp: wraps a sampling integrator and adds its XYZ result and the second moments of it as AOVs
    (``<name>.X/Y/Z`` and ``m2_<name>.X/Y/Z``), from which a per-pixel variance estimate follows."""

    def __init__(self, nested, name="nested"):
        if not isinstance(nested, PathIntegrator) or isinstance(nested, MomentIntegrator):
            raise RuntimeError("Child objects must be of type 'SamplingIntegrator'!")
        super().__init__(paths_per_wave=nested.paths_per_wave, pipeline=nested.pipeline)
        self.nested, self.name = nested, name

    def aov_names(self):
        base = ["%s.%s" % (self.name, c) for c in "XYZ"]
        return base + ["m2_" + n for n in base]

    def _desc(self, sensor, rows=None, partition=None):
        d = self.nested._desc(sensor, rows, partition)
        d.moment = 1
        return d

    @staticmethod
    def mean_and_variance(film):
        """what test_renders.py:55-60 (bitmap_extract) takes from the developed film: the nested integrator's XYZ image and
        m2 - mean^2, both normalised by the accumulated filter weight"""
        raw = film.bitmap(raw=True)
        w = raw[..., 4:5]
        inv = torch.where(w != 0, 1.0 / torch.where(w != 0, w, torch.ones_like(w)), torch.zeros_like(w))
        mean, m2 = raw[..., 5:8] * inv, raw[..., 8:11] * inv
        return mean, m2 - mean * mean


def make_sensor(params):
    """Build PerspectiveCamera/HDRFilm/IndependentSampler from a scenes.*_sensor() dict."""
    rp = params.get("rfilter_param")
    flt = make_filter(params["rfilter"], *([] if rp is None else (list(rp) if isinstance(rp, (list, tuple)) else [rp])))
    cx, cy, cw, ch = params["crop"]
    film = HDRFilm(params["width"], params["height"], (cx, cy), (cw, ch), flt)
    sampler = IndependentSampler(params["sample_count"], params["seed"])
    kw = dict(to_world=params["to_world"], fov=params["fov"], near_clip=params["near_clip"], far_clip=params["far_clip"], film=film, sampler=sampler)
    if params.get("aperture_radius") is not None:
        return ThinLensCamera(aperture_radius=params["aperture_radius"], focus_distance=params.get("focus_distance"), **kw)
    return PerspectiveCamera(**kw)
