"""Differentiable rendering: the counterpart of ``mitsuba.python.autodiff`` / ``mitsuba.python.util.traverse``
(``src/python/python/autodiff.py:6-91,121-194,200-377``, ``util.py:16-179``) for the parameters this backend supports --
diffuse reflectances, constant (``'<bsdf>.reflectance.value'``, ``src/spectra/srgb.cpp:59-61``) or bitmap
(``'<bsdf>.reflectance.data'``, ``src/textures/bitmap.cpp:295-299``).

The reference records the whole render in Enoki's autodiff graph and calls ``ek.backward``.  Here the forward pass is
the ordinary wavefront render into an R,G,B,A,W film, and the backward pass is ``mtsamd_render_adjoint``: every camera
sample is replayed with the same PCG32 stream and its vertices are swept backwards (``k_adjoint`` in
``csrc/kernels.hip``).  PyTorch only provides the autograd plumbing (as ``render_torch`` does in the reference,
``autodiff.py:380-482``).
"""
import ctypes as C
import math
from contextlib import contextmanager

import numpy as np
import torch

from . import bsdfs as B

from . import _lib as L
from .render import PathIntegrator, _ptr, _stream

_DERIV_SEED_OFFSET = 0x9E3779B97F4A7C15      # decorrelates the derivative pass from the primal pass (unbiased=True)


def _is_spectral(scene):
    return getattr(scene, "_variant", getattr(scene, "variant", "rgb")) == "spectral"


class ParameterMap:
    """Dictionary-like view of the differentiable scene parameters (util.py:16-130): torch tensors on the scene's GPU.
    Writes take effect in the scene after :meth:`update` (``parameters_changed``)."""

    def __init__(self, scene):
        self._scene = scene
        self.rebuild_envmap_distribution = True      # what parameters_changed() does (envmap.cpp:220-253); False: tests of linearity
        self.fd_step = 0.0                           # BSDF-model parameters: step of the central difference of the model code (0: 1 % of the value)
        self.properties = {}
        self._kind = {}
        dev = torch.device("cuda", scene._device_index)
        emitters = scene._dict.get("emitters", [])
        # 'my_envmap.data' (envmap.cpp:214-218): differentiable in any scene (any BSDF).  Layout: (H, W, 3) linear RGB here; the
        # reference's traverse() exposes the flat H * W * 4 buffer of its bitmap (RGB + a padding channel that carries no parameter,
        # envmap.cpp:216): a script written like invert_bunny.py reshapes with .reshape(H, W, 4)[..., :3] ("parity unpinned": no
        # reference test fixes the layout)
        for e, em in enumerate(emitters):
            if em.get("type", "area") == "envmap":
                key = em.get("id", "emitter_%d" % e) + ".data"
                self.properties[key] = torch.as_tensor(np.ascontiguousarray(em["data"], np.float32), dtype=torch.float32, device=dev).clone()
                self._kind[key] = ("envmap", e, e)
        # reflectances and area-light radiances are differentiated by the diffuse path replay (mtsamd_render_adjoint): scenes of
        # diffuse BSDFs (plain or inside `twosided`) lit by area lights only
        diffuse_scene = all(b["type"] == 0 for b in scene._bsdf_records) and \
            all(em.get("type", "area") == "area" for em in emitters)
        for i, b in enumerate(scene._bsdf_records if diffuse_scene else []):
            if b["type"] != 0:            # only diffuse reflectances are exposed (the adjoint pass covers those)
                continue
            name = b.get("id", "bsdf_%d" % i)
            if b["twosided"]:             # TwoSidedBRDF::traverse exposes its nested BSDF as "brdf_0" (twosided.cpp:183-186)
                name += ".brdf_0"
            refl = b["reflectance"]
            if isinstance(refl, dict) and refl.get("type") != "bitmap":
                continue                  # procedural textures have no differentiable texels
            if _is_spectral(scene) and (isinstance(refl, dict) or b.get("uniform_mask", 0) & 1):
                continue                  # spectral variant: constants that are srgb colours only (_spectral_gradient)
            if isinstance(refl, dict):
                key = name + ".reflectance.data"
                self.properties[key] = torch.as_tensor(refl["data"], dtype=torch.float32, device=dev).clone()
                self._kind[key] = ("texture", scene.texture_index(i), i)
            else:
                key = name + ".reflectance.value"
                self.properties[key] = torch.as_tensor([float(x) for x in refl], dtype=torch.float32, device=dev)
                self._kind[key] = ("bsdf", i, i)
        # parameters of the BSDF models beyond `diffuse` (what their traverse() exposes, e.g. roughconductor.cpp:393-404, plastic.cpp:299-307):
        # differentiated by mtsamd_render_adjoint_param in ANY scene (any emitter, any depth): constants only.  Spectral variant: the same
        # keys, differentiated by central differences of whole renders (_spectral_gradient); `uniform` spectra are not exposed
        for i, b in enumerate(scene._bsdf_records if not diffuse_scene else []):
            name = b.get("id", "bsdf_%d" % i) + (".brdf_0" if b.get("twosided") and b["type"] != B.DIFFUSE else "")
            flat_index = i                # top-level records keep their place in the table (bsdfs.flatten)
            for pname, kind, types in (("reflectance", 0, (B.DIFFUSE,)), ("diffuse_reflectance", 0, (B.PLASTIC, B.ROUGHPLASTIC)),
                                       ("specular_reflectance", 1, (B.CONDUCTOR, B.ROUGHCONDUCTOR, B.PLASTIC, B.ROUGHPLASTIC, B.DIELECTRIC, B.ROUGHDIELECTRIC, B.THINDIELECTRIC)),
                                       ("specular_transmittance", 5, (B.DIELECTRIC, B.ROUGHDIELECTRIC, B.THINDIELECTRIC)),
                                       ("eta", 2, (B.CONDUCTOR, B.ROUGHCONDUCTOR)), ("k", 3, (B.CONDUCTOR, B.ROUGHCONDUCTOR))):
                if b["type"] not in types:
                    continue
                src = b["reflectance"] if kind == 0 else b[pname]
                if isinstance(src, dict):
                    continue              # textured: no constant parameter
                if _is_spectral(scene) and (kind in (2, 3) or (b.get("uniform_mask", 0) >> {0: 0, 1: 1, 5: 2}[kind]) & 1):
                    continue              # spectral variant: eta / k and `uniform` spectra are not srgb colours
                key = "%s.%s.value" % (name, pname)
                self.properties[key] = torch.as_tensor([float(x) for x in src], dtype=torch.float32, device=dev)
                self._kind[key] = ("bsdf_param", flat_index, kind)
            if b["type"] in (B.ROUGHCONDUCTOR, B.ROUGHDIELECTRIC) and b["alpha_u"] == b["alpha_v"]:
                key = name + ".alpha.value"
                self.properties[key] = torch.as_tensor([float(b["alpha_u"])], dtype=torch.float32, device=dev)
                self._kind[key] = ("bsdf_param", flat_index, 4)
        # 'shape.emitter.radiance.value' of area lights (docs/src/inverse_rendering/diff_render.rst:76)
        for i, m in enumerate(scene._dict["meshes"] if diffuse_scene else []):
            e = m.get("emitter", -1)
            if e is None or e < 0 or scene._dict["emitters"][e].get("type", "area") != "area":
                continue
            key = m.get("id", "shape_%d" % i) + ".emitter.radiance.value"
            rad = np.broadcast_to(np.asarray(scene._dict["emitters"][e]["radiance"], np.float32), (3,))
            self.properties[key] = torch.as_tensor(rad.copy(), dtype=torch.float32, device=dev)
            self._kind[key] = ("emitter", e, i)

    def __getitem__(self, k): return self.properties[k]
    def __contains__(self, k): return k in self.properties
    def __len__(self): return len(self.properties)
    def keys(self): return self.properties.keys()
    def items(self): return self.properties.items()

    def __setitem__(self, k, value):
        if k not in self.properties:
            raise KeyError(k)
        old = self.properties[k]
        new = torch.as_tensor(value, dtype=torch.float32, device=old.device).reshape(old.shape).clone()
        new.requires_grad_(old.requires_grad)
        self.properties[k] = new

    def keep(self, keys):
        """util.py:120-129"""
        keys = set(keys)
        self.properties = {k: v for k, v in self.properties.items() if k in keys}

    def all_differentiable(self):
        return True

    def update(self):
        """util.py:103-118: push the current values into the scene (parameters_changed).  A value the scene has already seen -- the same
        tensor at the same version -- is not pushed again: render() calls this before every primal pass, and reading a constant back
        to the host costs a device synchronisation."""
        seen = self.__dict__.setdefault("_pushed", {})
        for k, v in self.properties.items():
            last = seen.get(k)
            if last is not None and last[0] is v and last[1] == v._version:      # (the tensor itself is kept: an id can be reused)
                continue
            kind, idx, _ = self._kind[k]
            if _is_spectral(self._scene) and (kind == "bsdf" or (kind == "bsdf_param" and self._kind[k][2] != 4)):
                # spectral variant: an srgb colour lives in [0, 1]^3 (srgb.cpp:34-35); an optimiser step that leaves the box is projected
                # back onto it (the parameter tensor itself, so that the optimiser's state and the scene agree)
                with torch.no_grad():
                    v.clamp_(0.0, 1.0)
            seen[k] = (v, v._version)
            if kind == "texture":
                self._scene.update_texture(idx, v)
            elif kind == "emitter":
                self._scene.set_emitter_radiance(idx, v.detach().cpu().tolist())
            elif kind == "envmap":
                self._scene.update_envmap(v, rebuild_distribution=self.rebuild_envmap_distribution)
            elif kind == "bsdf_param":
                vals = [float(x) for x in v.detach().cpu().reshape(-1).tolist()]
                self._scene.set_bsdf_param(idx, self._kind[k][2], vals)
            else:
                self._scene.set_bsdf_reflectance(idx, v.detach().cpu().tolist())


def traverse(scene):
    """mitsuba.python.util.traverse (util.py:132-179) for the supported parameters."""
    return ParameterMap(scene)


def _desc(scene, sensor, integrator, spp, seed):
    d = integrator._desc(sensor)
    if spp is not None:
        d.sample_count = int(spp)
    d.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    d.film_rgb = 1
    d.rfilter_analytic = 1          # CUDA variants evaluate the filter analytically (imageblock.cpp:131-132)
    return d


def _render_film(scene, d):
    dev = torch.device("cuda", scene._device_index)
    film = torch.zeros((d.crop_height, d.crop_width, 5), dtype=torch.float32, device=dev)
    L.check(L.lib().mtsamd_render(scene._handle, C.byref(d), _ptr(film), None, _stream()))
    return film


def _image_of(film):
    """autodiff.py:80-91: values / (weight + 1e-8), flattened RGB."""
    return (film[..., :3] / (film[..., 4:5] + 1e-8)).reshape(-1)


def _spectral_gradient(scene, d, pmap, key, gi):
    """Spectral variant: d(loss)/d(parameter) of a constant colour, radiance or roughness by central differences of the rendered
    image at fixed random numbers (the derivative pass's seed): two renders per scalar component.  The reference differentiates its
    spectral variants with Enoki's autodiff like the others; this backend has no spectral path replay, so the
    spectral ParameterMap is limited to a handful of constants -- texels and envmaps raise."""
    kind, idx, extra = pmap._kind[key]
    if kind in ("texture", "envmap"):
        raise RuntimeError("the spectral variant differentiates constant colours, radiances and roughnesses only (%s is a %s)" % (key, kind))
    value = [float(x) for x in pmap[key].detach().cpu().reshape(-1).tolist()]

    def push(vals):
        if kind == "emitter":
            scene.set_emitter_radiance(idx, vals)
        elif kind == "bsdf_param":
            scene.set_bsdf_param(idx, extra, vals)
        else:
            scene.set_bsdf_reflectance(idx, vals)

    grad = torch.zeros(len(value), dtype=torch.float32, device=gi.device)
    colour = kind != "emitter" and not (kind == "bsdf_param" and extra == 4)
    for c in range(len(value)):
        h = float(pmap.fd_step) if pmap.fd_step > 0 else 0.01 * max(abs(value[c]), 0.05)
        hi, lo = value[c] + h, value[c] - h
        if colour:
            hi, lo = min(hi, 1.0), max(lo, 0.0)                       # srgb colours live in [0, 1] (srgb.cpp:34-35)
        elif kind == "bsdf_param":
            lo = max(lo, 1e-4)                                        # roughness
        images = []
        for x in (hi, lo):
            push(value[:c] + [x] + value[c + 1:])
            images.append(_image_of(_render_film(scene, d)))
        grad[c] = torch.dot(gi, images[0] - images[1]) / (hi - lo)
    push(value)
    return grad.reshape(pmap[key].shape)


class _Render(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scene, d, pmap, keys, *values):
        pmap.update()                 # parameters_changed(): the scene sees the current values
        film = _render_film(scene, d)
        ctx.scene, ctx.d, ctx.pmap, ctx.keys = scene, d, pmap, keys
        ctx.save_for_backward(film)
        return _image_of(film)

    @staticmethod
    def backward(ctx, grad_image):
        (film,) = ctx.saved_tensors
        scene, d, pmap, keys = ctx.scene, ctx.d, ctx.pmap, ctx.keys
        dev = film.device
        n_bsdf = len(scene._dict["bsdfs"])
        tex_floats = sum(h * w * 3 for (h, w, _) in scene._texture_shapes)
        g_bsdf = torch.zeros((n_bsdf, 3), dtype=torch.float32, device=dev)
        g_tex = torch.zeros(max(tex_floats, 1), dtype=torch.float32, device=dev)
        g_em = torch.zeros((max(len(scene._dict.get("emitters", [])), 1), 3), dtype=torch.float32, device=dev)
        gi = grad_image.to(dev, torch.float32).contiguous()
        if _is_spectral(scene):
            return (None, None, None, None) + tuple(_spectral_gradient(scene, d, pmap, k, gi.reshape(-1)) for k in keys)
        kinds = {pmap._kind[k][0] for k in keys}
        if kinds - {"envmap", "bsdf_param"}:
            L.check(L.lib().mtsamd_render_adjoint(scene._handle, C.byref(d), _ptr(gi), _ptr(film), _ptr(g_bsdf), _ptr(g_tex), _ptr(g_em), _stream()))
        g_env = None
        if "envmap" in kinds:
            g_env = torch.zeros_like(next(pmap[k] for k in keys if pmap._kind[k][0] == "envmap"))
            L.check(L.lib().mtsamd_render_adjoint_envmap(scene._handle, C.byref(d), _ptr(gi), _ptr(film), _ptr(g_env), _stream()))
        grads = []
        for k in keys:
            kind, idx, _ = pmap._kind[k]
            if kind == "bsdf_param":          # one replay per scalar component (forward-mode derivative carried beside the path)
                n = pmap[k].numel()
                g = torch.zeros(n, dtype=torch.float32, device=dev)
                for c in range(n):
                    L.check(L.lib().mtsamd_render_adjoint_param(scene._handle, C.byref(d), _ptr(gi), _ptr(film), int(idx), int(pmap._kind[k][2]), c,
                                                                float(pmap.fd_step), _ptr(g[c:c + 1]), _stream()))
                grads.append(g.reshape(pmap[k].shape))
                continue
            if kind == "envmap":
                grads.append(g_env)
            elif kind == "bsdf":
                grads.append(g_bsdf[idx].clone())
            elif kind == "emitter":
                grads.append(g_em[idx].clone())
            else:
                off, w, h = C.c_uint64(), C.c_int32(), C.c_int32()
                L.check(L.lib().mtsamd_scene_texture_info(scene._handle, idx, C.byref(w), C.byref(h), C.byref(off)))
                grads.append(g_tex[off.value: off.value + 3 * w.value * h.value].reshape(h.value, w.value, 3).clone())
        return (None, None, None, None) + tuple(grads)


_render_counter = {}


def render(scene, spp=None, unbiased=False, optimizer=None, sensor_index=0, params=None):
    """mitsuba.python.autodiff.render (autodiff.py:121-194): differentiable render returning the flattened RGB image
    (``len == H*W*3``).  ``params`` (or ``optimizer.params``) is the :class:`ParameterMap` whose tensors with
    ``requires_grad`` receive gradients; without one the call is a plain render.  Every call draws new random numbers
    (the reference's sampler keeps advancing its streams between calls; here the call counter is folded into the seed).
    """
    sensor = scene.sensors()[sensor_index]
    integrator = scene.integrator() if scene.integrator() is not None else PathIntegrator()
    if optimizer is not None and params is None:
        params = optimizer.params
    if unbiased:
        if optimizer is None and params is None:
            raise Exception("render(): unbiased=True requires that an optimizer is specified!")
        if not isinstance(spp, tuple):
            spp = (spp, spp)
    elif isinstance(spp, tuple):
        raise Exception("render(): unbiased=False requires that spp is either an integer or None!")
    call = _render_counter.get(id(scene), 0)
    _render_counter[id(scene)] = call + 1
    base = sensor.sampler().seed_value() + call * 0xD1B54A32D192ED03

    def differentiable(spp_, seed):
        d = _desc(scene, sensor, integrator, spp_, seed)
        keys = [k for k, v in params.items() if v.requires_grad] if params is not None else []
        if not keys:
            if params is not None:
                params.update()
            return _image_of(_render_film(scene, d))
        return _Render.apply(scene, d, params, keys, *[params[k] for k in keys])

    if not unbiased:
        return differentiable(spp, base)
    with torch.no_grad():
        if params is not None:
            params.update()
        image = _image_of(_render_film(scene, _desc(scene, sensor, integrator, spp[0], base)))
    image_diff = differentiable(spp[1], base + _DERIV_SEED_OFFSET)
    return image_diff + (image - image_diff).detach()        # ek.reattach(image, image_diff), autodiff.py:185-187


def render_torch(scene, params=None, **kwargs):
    """autodiff.py:380-482: the image as a torch tensor whose autograd graph reaches ``params``."""
    return render(scene, params=params, **kwargs)


class Optimizer:
    """autodiff.py:200-243"""

    def __init__(self, params, lr):
        self.set_learning_rate(lr)
        self.params = params
        if not params.all_differentiable():
            raise Exception("Optimizer.__init__(): all parameters should be differentiable!")
        self.state = {}
        for k, p in self.params.items():
            p.requires_grad_(True)
            self._reset(k)

    def set_learning_rate(self, lr):
        self.lr = lr

    @contextmanager
    def disable_gradients(self):
        for _, p in self.params.items():
            p.requires_grad_(False)
        try:
            yield
        finally:
            for _, p in self.params.items():
                p.requires_grad_(True)

    def _replace(self, key, value):
        value = value.detach()           # a fresh tensor (the result of the update expression): no copy needed
        value.requires_grad_(True)
        self.params.properties[key] = value


class SGD(Optimizer):
    """autodiff.py:246-306: v <- mu v + g ; p <- p - lr v"""

    def __init__(self, params, lr, momentum=0):
        assert momentum >= 0 and momentum < 1
        assert lr > 0
        self.momentum = momentum
        super().__init__(params, lr)

    def step(self):
        for k, p in list(self.params.items()):
            g = p.grad
            if g is None:
                continue
            if self.momentum != 0:
                self.state[k] = self.momentum * self.state[k] + g
                value = p.detach() - self.lr * self.state[k]
            else:
                value = p.detach() - self.lr * g
            self._replace(k, value)
        self.params.update()

    def _reset(self, key):
        if self.momentum == 0:
            return
        self.state[key] = torch.zeros_like(self.params[key])

    def __repr__(self):
        return "SGD[\n  lr = %.2g,\n  momentum = %.2g\n]" % (self.lr, self.momentum)


class Adam(Optimizer):
    """autodiff.py:309-377"""

    def __init__(self, params, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-8):
        super().__init__(params, lr)
        assert 0 <= beta_1 < 1 and 0 <= beta_2 < 1 and lr > 0 and epsilon > 0
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.t = 0

    def step(self):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.beta_2 ** self.t) / (1 - self.beta_1 ** self.t)
        for k, p in list(self.params.items()):
            g = p.grad
            if g is None:
                continue
            m_t, v_t = self.state[k]                       # updated in place: m <- b1 m + (1 - b1) g, v <- b2 v + (1 - b2) g^2
            m_t.mul_(self.beta_1).add_(g, alpha=1 - self.beta_1)
            v_t.mul_(self.beta_2).addcmul_(g, g, value=1 - self.beta_2)
            self._replace(k, torch.addcdiv(p.detach(), m_t, torch.sqrt(v_t).add_(self.epsilon), value=-lr_t))
        self.params.update()      # the reference leaves this to the next render's parameters_changed(); explicit here

    def _reset(self, key):
        p = self.params[key]
        self.state[key] = (torch.zeros_like(p), torch.zeros_like(p))

    def __repr__(self):
        return "Adam[\n  lr = %g,\n  betas = (%g, %g),\n  eps = %g\n]" % (self.lr, self.beta_1, self.beta_2, self.epsilon)
