"""Differentiable rendering on the GPU (BASELINE config 4): the adjoint kernel against the oracle's adjoint and against
finite differences of the GPU forward pass, the torch autograd plumbing, and a small inverse-rendering run in the
style of docs/examples/10_inverse_rendering/invert_cbox.py."""
import ctypes as C

import numpy as np
import pytest
import torch

from mitsuba2_amd import scenes

pytestmark = pytest.mark.gpu


def _scene(gpu, tex, w, h, spp, max_depth, rfilter="box", seed=5, two_lights=False):
    sd = scenes.cornell_box(texture=tex)
    sd["meshes"][5]["id"] = "lamp"
    if two_lights:                                     # a second area light: the emitter-selection branch of scene.cpp:141-189
        sd["emitters"] = list(sd["emitters"]) + [dict(type="area", radiance=np.array([2.0, 3.0, 4.0], np.float32))]
        sd["meshes"][7] = dict(sd["meshes"][7], emitter=1, id="glowing_box")
    names = ["white", "red", "green", "light", "textured"]
    for b, n in zip(sd["bsdfs"], names):
        b["id"] = n
    p = scenes.cornell_box_sensor(w, h, spp, seed=seed, max_depth=max_depth, rfilter=rfilter)
    sensor = gpu.make_sensor(p)
    scene = gpu.Scene(sd, sensor=sensor, integrator=gpu.PathIntegrator(max_depth=max_depth))
    return sd, p, scene


@pytest.mark.parametrize("rfilter,max_depth", [("box", 3), ("gaussian", 4), ("box", 8)])
def test_adjoint_matches_oracle(gpu, oracle, rfilter, max_depth):
    from mitsuba2_amd import autodiff, _lib as L
    rng = np.random.RandomState(1)
    tex = (0.3 + 0.5 * rng.rand(4, 5, 3)).astype(np.float32)
    sd, p, scene = _scene(gpu, tex, 24, 20, 4, max_depth, rfilter, two_lights=max_depth == 8)
    d = autodiff._desc(scene, scene.sensors()[0], scene.integrator(), None, p["seed"])
    film = autodiff._render_film(scene, d)
    desc = oracle.make_desc(p, analytic=True, film_rgb=True)
    S = oracle.OracleScene(sd, naive=True)
    image_o, film_o = S.render_image(desc)
    assert np.allclose(film.cpu().numpy()[..., 4], film_o[..., 4], rtol=1e-5, atol=1e-6)
    assert np.mean((film.cpu().numpy()[..., :3] - film_o[..., :3]) ** 2 / (film_o[..., :3] ** 2 + 1e-2)) < 1e-5
    dimage = np.random.RandomState(2).randn(20, 24, 3).astype(np.float32)
    gs_o, gt_o, ge_o = S.render_adjoint(desc, dimage, film_o, len(sd["meshes"]), tex.size, n_emitters=len(sd["emitters"]))
    g_bsdf = torch.zeros((len(sd["bsdfs"]), 3), device="cuda")
    g_tex = torch.zeros(tex.size, device="cuda")
    g_em = torch.zeros((len(sd["emitters"]), 3), device="cuda")
    di = torch.from_numpy(dimage).cuda()
    L.check(L.lib().mtsamd_render_adjoint(scene._handle, C.byref(d), C.c_void_p(di.data_ptr()), C.c_void_p(film.data_ptr()),
                                          C.c_void_p(g_bsdf.data_ptr()), C.c_void_p(g_tex.data_ptr()), C.c_void_p(g_em.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.abs(ge_o).min() > 1e-3                   # 'shape.emitter.radiance.value' of every area light
    assert np.allclose(g_em.cpu().numpy(), ge_o, rtol=2e-2, atol=2e-3 * np.abs(ge_o).max())
    gt = g_tex.cpu().numpy()
    assert np.abs(gt_o).max() > 1e-3
    assert np.allclose(gt, gt_o, rtol=2e-2, atol=2e-3 * np.abs(gt_o).max())
    # per-BSDF gradient = sum over the shapes that share it
    gb_o = np.zeros((len(sd["bsdfs"]), 3), np.float32)
    for si, m in enumerate(sd["meshes"]):
        gb_o[m["bsdf"]] += gs_o[si]
    gb = g_bsdf.cpu().numpy()
    assert np.allclose(gb, gb_o, rtol=2e-2, atol=2e-3 * np.abs(gb_o).max())
    assert (gb[4] == 0).all()                  # the textured BSDF has no constant-reflectance gradient


def test_autograd_and_finite_differences(gpu):
    from mitsuba2_amd import autodiff
    rng = np.random.RandomState(3)
    tex = (0.3 + 0.5 * rng.rand(6, 6, 3)).astype(np.float32)
    sd, p, scene = _scene(gpu, tex, 32, 32, 8, 4, "gaussian")
    params = autodiff.traverse(scene)
    assert set(params.keys()) == {"white.reflectance.value", "red.reflectance.value", "green.reflectance.value",
                                  "light.reflectance.value", "textured.reflectance.data", "lamp.emitter.radiance.value"}
    params.keep(["red.reflectance.value", "textured.reflectance.data", "lamp.emitter.radiance.value"])
    for k in list(params.keys()):
        params[k].requires_grad_(True)
    target = torch.from_numpy(np.random.RandomState(4).rand(32 * 32 * 3).astype(np.float32)).cuda()

    def loss_at(seed_call):
        autodiff._render_counter[id(scene)] = seed_call          # same random numbers for every evaluation
        img = autodiff.render(scene, params=params)
        return ((img - target) ** 2).sum() / img.numel(), img

    loss, img = loss_at(7)
    assert img.shape == (32 * 32 * 3,) and img.requires_grad
    loss.backward()
    g_red = params["red.reflectance.value"].grad.clone()
    g_tex = params["textured.reflectance.data"].grad.clone()
    g_lamp = params["lamp.emitter.radiance.value"].grad.clone()
    assert g_tex.shape == (6, 6, 3) and g_tex.abs().max() > 0 and g_red.abs().max() > 0 and g_lamp.abs().min() > 0
    eps = 2e-2
    with torch.no_grad():
        for key, idx, g in (("red.reflectance.value", (0,), g_red), ("red.reflectance.value", (2,), g_red),
                            ("lamp.emitter.radiance.value", (0,), g_lamp), ("lamp.emitter.radiance.value", (2,), g_lamp),
                            ("textured.reflectance.data", (2, 3, 1), g_tex), ("textured.reflectance.data", (4, 1, 0), g_tex)):
            base = params[key].detach().clone()
            vp, vm = base.clone(), base.clone()
            h = 0.5 if "radiance" in key else eps          # the loss is quadratic in the radiance: central differences are exact
            vp[idx] += h; vm[idx] -= h
            params[key] = vp; lp, _ = loss_at(7)
            params[key] = vm; lm, _ = loss_at(7)
            params[key] = base
            fd = (lp.item() - lm.item()) / (2 * h)
            assert abs(fd - g[idx].item()) <= 3e-2 * max(abs(fd), abs(g[idx].item())) + 1e-6, (key, idx, fd, g[idx].item())
    # unbiased mode: value of the primal render, gradient of a decorrelated one
    params["red.reflectance.value"].requires_grad_(True)
    img_u = autodiff.render(scene, params=params, unbiased=True, spp=(4, 2))
    assert img_u.requires_grad
    (img_u.sum()).backward()
    with pytest.raises(Exception):
        autodiff.render(scene, unbiased=False, spp=(1, 1))


def test_invert_cbox(gpu):
    """docs/examples/10_inverse_rendering/invert_cbox.py: recover the red wall's albedo from (.9,.9,.9)."""
    from mitsuba2_amd import autodiff
    sd = scenes.cornell_box()
    for b, n in zip(sd["bsdfs"], ["white", "red", "green", "light"]):
        b["id"] = n
    p = scenes.cornell_box_sensor(64, 64, 8, seed=1, max_depth=3, rfilter="box")
    scene = gpu.Scene(sd, sensor=gpu.make_sensor(p), integrator=gpu.PathIntegrator(max_depth=3))
    params = autodiff.traverse(scene)
    params.keep(["red.reflectance.value"])
    param_ref = params["red.reflectance.value"].clone()
    image_ref = autodiff.render(scene, spp=32).detach()
    params["red.reflectance.value"] = [0.9, 0.9, 0.9]
    params.update()
    opt = autodiff.Adam(params, lr=0.2)
    errs = []
    for it in range(40):
        image = autodiff.render(scene, optimizer=opt, unbiased=True, spp=4)
        ob_val = ((image - image_ref) ** 2).sum() / image.numel()
        ob_val.backward()
        opt.step()
        errs.append(((param_ref - params["red.reflectance.value"].detach()) ** 2).sum().item())
    assert errs[-1] < 0.02 * errs[0], errs[::5]
    # SGD with momentum also runs
    sgd = autodiff.SGD(params, lr=0.5, momentum=0.9)
    image = autodiff.render(scene, optimizer=sgd, spp=2)
    (((image - image_ref) ** 2).sum() / image.numel()).backward()
    sgd.step()


def _envmap_scene(gpu, w, h, spp, max_depth, with_area, rfilter="box", seed=9):
    from test_adjoint_cpu import _envmap_scene as build
    sd, img = build(with_area)
    p = scenes.cornell_box_sensor(w, h, spp, seed=seed, max_depth=max_depth, rfilter=rfilter)
    scene = gpu.Scene(sd, sensor=gpu.make_sensor(p), integrator=gpu.PathIntegrator(max_depth=max_depth))
    return sd, img, p, scene


def test_invert_cbox_texture(gpu):
    """BASELINE config 4 as worded ("optimise diffuse-albedo texture on cbox"): the texels of the bitmap albedo on the back wall and the
    floor ('textured.reflectance.data', src/textures/bitmap.cpp:250-299) are recovered from a uniform grey by the loop of invert_cbox.py.
    A 16 x 16 texture seen by a 96 x 96 film: every texel is covered by many pixels, so the inverse problem is well posed."""
    from mitsuba2_amd import autodiff
    yy, xx = np.meshgrid(np.linspace(0, 1, 16, dtype=np.float32), np.linspace(0, 1, 16, dtype=np.float32), indexing="ij")
    tex = np.stack([0.5 + 0.35 * np.sin(6 * xx) * np.cos(5 * yy), 0.45 + 0.3 * np.cos(4 * xx + 3 * yy), 0.4 + 0.3 * np.sin(7 * yy)], -1).astype(np.float32)
    sd = scenes.cornell_box(texture=tex)
    for b, n in zip(sd["bsdfs"], ["white", "red", "green", "light", "textured"]):
        b["id"] = n
    p = scenes.cornell_box_sensor(96, 96, 8, seed=2, max_depth=3, rfilter="box")
    scene = gpu.Scene(sd, sensor=gpu.make_sensor(p), integrator=gpu.PathIntegrator(max_depth=3))
    params = autodiff.traverse(scene)
    key = "textured.reflectance.data"
    params.keep([key])
    ref = params[key].clone()
    assert tuple(ref.shape) == (16, 16, 3)
    image_ref = autodiff.render(scene, spp=64).detach()
    params[key] = torch.full_like(ref, 0.5)
    params.update()
    opt = autodiff.Adam(params, lr=0.03)
    errs = []
    for it in range(150):
        image = autodiff.render(scene, optimizer=opt, unbiased=True, spp=8)
        (((image - image_ref) ** 2).sum() / image.numel()).backward()
        opt.step()
        errs.append(((ref - params[key].detach()) ** 2).mean().item())
    # texels the camera sees (the floor is seen at a grazing angle, parts of both walls are hidden by the boxes): the bulk is recovered
    assert errs[-1] < 0.25 * errs[0], errs[::15]
    params[key] = ref
    params.update()


@pytest.mark.parametrize("with_area,rfilter,max_depth", [(False, "box", 4), (True, "gaussian", 6)])
def test_envmap_adjoint_matches_oracle(gpu, oracle, with_area, rfilter, max_depth):
    """mtsamd_render_adjoint_envmap ('my_envmap.data', invert_bunny.py) against the oracle's path replay: glossy + dielectric boxes,
    MIS between emitter and BSDF sampling, Russian roulette, with and without a second (area) emitter"""
    from mitsuba2_amd import autodiff, _lib as L
    sd, img, p, scene = _envmap_scene(gpu, 28, 24, 8, max_depth, with_area, rfilter)
    d = autodiff._desc(scene, scene.sensors()[0], scene.integrator(), None, p["seed"])
    film = autodiff._render_film(scene, d)
    desc = oracle.make_desc(p, analytic=True, film_rgb=True)
    S = oracle.OracleScene(sd, naive=True)
    image_o, film_o = S.render_image(desc)
    # a few of the 5 k samples take another branch on the two sides (libm vs OCML in the microfacet lobes)
    assert np.mean((film.cpu().numpy()[..., :3] - film_o[..., :3]) ** 2 / (film_o[..., :3] ** 2 + 1e-2)) < 1e-3
    dimage = np.random.RandomState(2).randn(24, 28, 3).astype(np.float32)
    g_o = S.render_adjoint_envmap(desc, dimage, film_o, img.shape)
    g = torch.zeros(img.shape, device="cuda")
    di = torch.from_numpy(dimage).cuda()
    L.check(L.lib().mtsamd_render_adjoint_envmap(scene._handle, C.byref(d), C.c_void_p(di.data_ptr()), C.c_void_p(film.data_ptr()),
                                                 C.c_void_p(g.data_ptr()), None))
    torch.cuda.synchronize()
    g = g.cpu().numpy()
    assert np.abs(g_o).max() > 1e-2
    # a sample whose path takes another branch on the two sides (libm vs OCML in the microfacet lobes) moves a few texels' sums
    assert np.allclose(g, g_o, rtol=5e-2, atol=2e-2 * np.abs(g_o).max())
    assert abs(g.sum() - g_o.sum()) < 1e-2 * np.abs(g_o).sum()


def test_envmap_autograd_and_inversion(gpu):
    """the workflow of docs/examples/10_inverse_rendering/invert_bunny.py on a small scene: traverse -> keep 'my_envmap.data' -> render
    a reference -> uniform lighting -> Adam on the texels"""
    from mitsuba2_amd import autodiff
    sd, img, p, scene = _envmap_scene(gpu, 48, 40, 8, 4, False, "gaussian")
    params = autodiff.traverse(scene)
    # the envmap texels + (round 3) the constant parameters of the scene's BSDF models
    assert "my_envmap.data" in params and all(k.split(".")[-2] in ("reflectance", "diffuse_reflectance", "specular_reflectance", "specular_transmittance",
                                                                   "eta", "k", "alpha", "my_envmap") for k in params.keys())
    params.keep(["my_envmap.data"])
    param_ref = params["my_envmap.data"].clone()
    assert param_ref.shape == (6, 10, 3)
    # central differences through the autograd function (distribution held fixed: the image is linear in the texels)
    params.rebuild_envmap_distribution = False
    params["my_envmap.data"].requires_grad_(True)
    target = torch.from_numpy(np.random.RandomState(4).rand(40 * 48 * 3).astype(np.float32)).cuda()

    def loss_at():
        autodiff._render_counter[id(scene)] = 11
        im = autodiff.render(scene, params=params)
        return ((im - target) ** 2).sum() / im.numel()

    loss = loss_at()
    loss.backward()
    g = params["my_envmap.data"].grad.clone()
    assert g.shape == (6, 10, 3) and (g.abs() > 0).float().mean() > 0.5
    with torch.no_grad():
        for idx in ((1, 6, 0), (4, 3, 1), (0, 0, 2)):
            base = params["my_envmap.data"].detach().clone()
            vp, vm = base.clone(), base.clone()
            vp[idx] += 0.25; vm[idx] -= 0.25
            params["my_envmap.data"] = vp; lp = loss_at().item()
            params["my_envmap.data"] = vm; lm = loss_at().item()
            params["my_envmap.data"] = base
            fd = (lp - lm) / 0.5
            assert abs(fd - g[idx].item()) <= 3e-2 * max(abs(fd), abs(g[idx].item())) + 1e-6, (idx, fd, g[idx].item())
    # inversion, as the example does it (the distribution follows the texels again)
    params.rebuild_envmap_distribution = True
    params["my_envmap.data"] = param_ref
    params.update()
    image_ref = autodiff.render(scene, spp=64).detach()
    params["my_envmap.data"] = torch.full_like(param_ref, 1.0)
    params.update()
    opt = autodiff.Adam(params, lr=0.1)
    errs = []
    for it in range(80):
        image = autodiff.render(scene, optimizer=opt, unbiased=True, spp=8)
        ((image - image_ref) ** 2).sum().div(image.numel()).backward()
        opt.step()
        errs.append(((autodiff.render(scene, spp=16).detach() - image_ref) ** 2).mean().item() if it % 20 == 19 or it == 0 else None)
    vals = [e for e in errs if e is not None]
    assert vals[-1] < 0.3 * vals[0] and all(b < a for a, b in zip(vals, vals[1:])), vals


def test_twosided_diffuse_adjoint_matches_oracle(gpu, oracle):
    """the adjoint path replay through `twosided` diffuse BSDFs (twosided.cpp:94-175; the primal render of such a scene runs the general
    kernels): a Cornell box whose red wall is wound the other way, every BSDF inside `twosided` -- gradient against the oracle's,
    parameter names as TwoSidedBRDF::traverse exposes them (twosided.cpp:183-186)"""
    from mitsuba2_amd import autodiff, _lib as L
    from test_adjoint_cpu import _twosided_cbox
    sd, wall = _twosided_cbox()
    p = scenes.cornell_box_sensor(24, 20, 4, seed=5, max_depth=5, rfilter="box")
    sensor = gpu.make_sensor(p)
    scene = gpu.Scene(sd, sensor=sensor, integrator=gpu.PathIntegrator(max_depth=5))
    d = autodiff._desc(scene, sensor, scene.integrator(), None, p["seed"])
    film = autodiff._render_film(scene, d)
    desc = oracle.make_desc(p, analytic=True, film_rgb=True)
    S = oracle.OracleScene(sd, naive=True)
    image_o, film_o = S.render_image(desc)
    assert np.mean((film.cpu().numpy()[..., :3] - film_o[..., :3]) ** 2 / (film_o[..., :3] ** 2 + 1e-2)) < 1e-5
    dimage = np.random.RandomState(2).randn(20, 24, 3).astype(np.float32)
    gs_o, _ = S.render_adjoint(desc, dimage, film_o, len(sd["meshes"]), 0)
    g_bsdf = torch.zeros((len(sd["bsdfs"]), 3), device="cuda")
    di = torch.from_numpy(dimage).cuda()
    L.check(L.lib().mtsamd_render_adjoint(scene._handle, C.byref(d), C.c_void_p(di.data_ptr()), C.c_void_p(film.data_ptr()),
                                          C.c_void_p(g_bsdf.data_ptr()), None, None, None))
    torch.cuda.synchronize()
    gb_o = np.zeros((len(sd["bsdfs"]), 3), np.float32)
    for si, m in enumerate(sd["meshes"]):
        gb_o[m["bsdf"]] += gs_o[si]
    assert np.abs(gs_o[wall]).max() > 1e-3                   # the back-facing wall takes part
    assert np.allclose(g_bsdf.cpu().numpy(), gb_o, rtol=2e-2, atol=2e-3 * np.abs(gb_o).max())
    params = autodiff.traverse(scene)
    assert "b1.brdf_0.reflectance.value" in params and "b0.brdf_0.reflectance.value" in params
    # a scene with any other BSDF is still refused by the replay
    sd2 = scenes.cornell_box(); sd2["bsdfs"] = list(sd2["bsdfs"]); sd2["bsdfs"][1] = {"type": "conductor"}
    scene2 = gpu.Scene(sd2, sensor=sensor, integrator=gpu.PathIntegrator(max_depth=5))
    with pytest.raises(RuntimeError, match="diffuse BSDFs"):
        L.check(L.lib().mtsamd_render_adjoint(scene2._handle, C.byref(d), C.c_void_p(di.data_ptr()), C.c_void_p(film.data_ptr()),
                                              C.c_void_p(g_bsdf.data_ptr()), None, None, None))


def _material_scene(gpu, materials, w=48, h=48, spp=64, max_depth=5, seed=13):
    """Cornell box whose tall / short boxes carry the given BSDF dictionaries (ids 'tall' / 'small'), floor two-sided diffuse"""
    sd = scenes.cornell_box()
    sd["bsdfs"] = list(sd["bsdfs"]) + [dict(materials[0], id="tall"), dict(materials[1], id="small")]
    sd["meshes"][6] = dict(sd["meshes"][6], bsdf=len(sd["bsdfs"]) - 2)
    sd["meshes"][7] = dict(sd["meshes"][7], bsdf=len(sd["bsdfs"]) - 1)
    p = scenes.cornell_box_sensor(w, h, spp, seed=seed, max_depth=max_depth, rfilter="box")
    scene = gpu.Scene(sd, sensor=gpu.make_sensor(p), integrator=gpu.PathIntegrator(max_depth=max_depth))
    return sd, p, scene


def _loss_and_grad(autodiff, scene, params, key, weights, spp):
    """loss = <weights, image> of one differentiable render with a FIXED seed (the call counter is reset), d loss / d params[key]"""
    autodiff._render_counter[id(scene)] = 0
    params[key].requires_grad_(True)
    img = autodiff.render(scene, params=params, spp=spp)
    loss = (img * weights).sum()
    loss.backward()
    g = params[key].grad.clone()
    params[key].grad = None
    params[key].requires_grad_(False)
    return float(loss.item()), g


def _loss_only(autodiff, scene, params, key, value, weights, spp):
    autodiff._render_counter[id(scene)] = 0
    params[key] = value
    params.update()
    with torch.no_grad():
        return float((autodiff.render(scene, params=params, spp=spp) * weights).sum().item())


def test_bsdf_parameter_gradients(gpu):
    """a21 beyond diffuse reflectances (round 3): d(image)/d(parameter) for the parameters the reference's traverse() exposes of the f-2
    models (roughconductor.cpp:393-404: alpha, eta, k, specular_reflectance; plastic.cpp:299-307: diffuse_reflectance, ...), through
    mtsamd_render_adjoint_param: a forward-mode derivative carried beside every replayed path, detached sampling.  Checked against central
    finite differences of the primal render with common random numbers: exact for parameters the sampled directions do not depend on
    (a smooth conductor's specular_reflectance: the image is a polynomial in it), statistical for the others (the finite difference
    moves the sampled directions with the parameter -- the reference's attached estimator -- the adjoint holds them fixed; both
    estimate the same derivative)."""
    from mitsuba2_amd import autodiff
    torch.manual_seed(3)
    mats = ({"type": "roughconductor", "alpha": 0.3, "distribution": "ggx", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14], "specular_reflectance": [0.9, 0.8, 0.7]},
            {"type": "plastic", "diffuse_reflectance": [0.2, 0.5, 0.3], "int_ior": 1.6})
    sd, p, scene = _material_scene(gpu, mats)
    params = autodiff.traverse(scene)
    for k in ("tall.alpha.value", "tall.eta.value", "tall.k.value", "tall.specular_reflectance.value", "small.diffuse_reflectance.value",
              "small.specular_reflectance.value"):
        assert k in params, (k, list(params.keys()))
    n_pix = 48 * 48
    weights = torch.ones(n_pix * 3, device="cuda") / n_pix                      # loss = mean radiance per channel, summed
    spp = 64
    for key, steps in (("tall.alpha.value", 0.03), ("small.diffuse_reflectance.value", 0.03), ("tall.k.value", 0.15), ("tall.specular_reflectance.value", 0.03)):
        params.keep([key]) if False else None
        base = params[key].detach().clone()
        _, g = _loss_and_grad(autodiff, scene, params, key, weights, spp)
        assert torch.isfinite(g).all() and float(g.abs().max()) > 0
        for c in range(base.numel()):
            vp, vm = base.clone(), base.clone()
            vp.view(-1)[c] += steps; vm.view(-1)[c] -= steps
            fd = (_loss_only(autodiff, scene, params, key, vp, weights, spp) - _loss_only(autodiff, scene, params, key, vm, weights, spp)) / (2 * steps)
            got = float(g.view(-1)[c])
            assert abs(got - fd) < 0.2 * abs(fd) + 2e-4, (key, c, got, fd)
        params[key] = base
        params.update()
    # a smooth conductor: sampled directions do not depend on the parameter, the image is a polynomial in it -> tight agreement
    mats2 = ({"type": "conductor", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14], "specular_reflectance": [0.8, 0.7, 0.6]}, {"type": "diffuse", "reflectance": [0.5, 0.5, 0.5]})
    sd2, p2, scene2 = _material_scene(gpu, mats2, spp=16)
    params2 = autodiff.traverse(scene2)
    key = "tall.specular_reflectance.value"
    base = params2[key].detach().clone()
    _, g = _loss_and_grad(autodiff, scene2, params2, key, weights, 16)
    for c in range(3):
        vp, vm = base.clone(), base.clone()
        vp[c] += 0.02; vm[c] -= 0.02
        fd = (_loss_only(autodiff, scene2, params2, key, vp, weights, 16) - _loss_only(autodiff, scene2, params2, key, vm, weights, 16)) / 0.04
        assert abs(float(g[c]) - fd) < 1e-2 * abs(fd) + 1e-5, (c, float(g[c]), fd)


@pytest.mark.parametrize("key,kind,step", [("tall.alpha.value", 4, 0.003), ("tall.k.value", 3, 0.03), ("tall.specular_reflectance.value", 1, 0.01),
                                           ("small.diffuse_reflectance.value", 0, 0.005)])
def test_bsdf_parameter_adjoint_matches_oracle(gpu, oracle, key, kind, step):
    """mtsamd_render_adjoint_param against the oracle's restatement of the same forward-mode replay (mo_render_adjoint_param: the same
    terms in the same order, the same central difference of the model code): every sample takes the same path on both sides
    (bit-identical primal, section 2), so the two sums agree to the accumulated rounding of 1.5e5 fp32 terms"""
    import ctypes as C
    from mitsuba2_amd import autodiff, _lib as L
    mats = ({"type": "roughconductor", "alpha": 0.3, "distribution": "ggx", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14], "specular_reflectance": [0.9, 0.8, 0.7]},
            {"type": "plastic", "diffuse_reflectance": [0.2, 0.5, 0.3], "int_ior": 1.6})
    sd, p, scene = _material_scene(gpu, mats, w=40, h=32, spp=16, max_depth=5)
    sensor = gpu.make_sensor(p)
    d = autodiff._desc(scene, sensor, gpu.PathIntegrator(max_depth=5), 16, 77)
    film = autodiff._render_film(scene, d)
    rng = np.random.RandomState(4)
    dimage = rng.uniform(-1.0, 1.0, (32, 40, 3)).astype(np.float32)
    di = torch.from_numpy(dimage).cuda()
    bsdf_index = len(sd["bsdfs"]) - (2 if key.startswith("tall") else 1)
    shapes = [i for i, m in enumerate(sd["meshes"]) if m["bsdf"] == bsdf_index]
    S = oracle.OracleScene(sd)
    desc = oracle.make_desc(dict(p, seed=77), analytic=True, film_rgb=True)
    _, film_o = S.render_image(desc)
    assert np.allclose(film.cpu().numpy(), film_o, rtol=2e-5, atol=1e-6)
    for comp in range(1 if kind == 4 else 3):
        g = torch.zeros(1, device="cuda")
        L.check(L.lib().mtsamd_render_adjoint_param(scene._handle, C.byref(d), C.c_void_p(di.data_ptr()), C.c_void_p(film.data_ptr()), bsdf_index, kind, comp,
                                                    step, C.c_void_p(g.data_ptr()), None))
        torch.cuda.synchronize()
        want = S.render_adjoint_param(desc, dimage, film_o, shapes, kind, comp, step)
        assert abs(want) > 1e-3
        assert abs(float(g.item()) - want) < 2e-3 * abs(want) + 1e-5, (key, comp, float(g.item()), want)


def test_invert_roughness(gpu):
    """the loop of invert_cbox.py on a parameter of a microfacet model: the GGX roughness of the tall box is recovered from a wrong start
    (roughconductor.cpp:393-404 exposes `alpha`).  The renders are seeded by the call counter: the run is deterministic."""
    from mitsuba2_amd import autodiff
    mats = ({"type": "roughconductor", "alpha": 0.15, "distribution": "ggx", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14]},
            {"type": "diffuse", "reflectance": [0.5, 0.5, 0.5]})
    sd, p, scene = _material_scene(gpu, mats, w=64, h=64, spp=16, max_depth=4, seed=21)
    params = autodiff.traverse(scene)
    key = "tall.alpha.value"
    params.keep([key])
    image_ref = autodiff.render(scene, spp=512).detach()
    params[key] = [0.45]
    params.update()
    opt = autodiff.Adam(params, lr=0.02)
    trace = []
    for it in range(40):
        image = autodiff.render(scene, optimizer=opt, unbiased=True, spp=16)
        (((image - image_ref) ** 2).sum() / image.numel()).backward()
        opt.step()
        trace.append(float(params[key].item()))
    assert trace[5] < 0.45 and min(trace) > 0.05            # it moves the right way and stays inside the parameter's domain
    assert abs(float(np.mean(trace[-10:])) - 0.15) < 0.06, trace[::4]


def test_spectral_variant_gradients_and_inversion(gpu, oracle):
    """The spectral variant through mitsuba2_amd.autodiff (round 3): constant srgb colours and area-light radiances are differentiated by
    central differences of the rendered image at fixed random numbers (two renders per component; there is no spectral path replay).
    Checked against the same central difference of the ORACLE's spectral renders (identical samples, so the two differences agree to
    rounding), then used for a short inverse-rendering run on the red wall."""
    import copy
    from mitsuba2_amd import autodiff
    sd = scenes.cornell_box()
    for b, n in zip(sd["bsdfs"], ["white", "red", "green", "light"]):
        b["id"] = n
    sd["meshes"][5]["id"] = "lamp"
    p = scenes.cornell_box_sensor(20, 16, 16, seed=3, max_depth=4, rfilter="box")
    sensor = gpu.make_sensor(p)
    scene = gpu.Scene(sd, variant="spectral", sensor=sensor, integrator=gpu.PathIntegrator(max_depth=4))
    params = autodiff.traverse(scene)
    assert set(params.keys()) == {"white.reflectance.value", "red.reflectance.value", "green.reflectance.value", "light.reflectance.value",
                                  "lamp.emitter.radiance.value"}
    params.keep(["red.reflectance.value", "lamp.emitter.radiance.value"])
    params.fd_step = 0.01
    for v in params.properties.values():
        v.requires_grad_(True)
    dimage = torch.from_numpy(np.random.RandomState(7).randn(16 * 20 * 3).astype(np.float32)).cuda()
    image = autodiff.render(scene, spp=16, params=params)
    (image * dimage).sum().backward()
    g_red = params["red.reflectance.value"].grad.cpu().numpy()
    g_lamp = params["lamp.emitter.radiance.value"].grad.cpu().numpy()
    # the oracle's central differences, same seed (render() folds its call counter into the seed: call 0 = the sensor's seed)
    desc = oracle.make_desc(dict(p, seed=sensor.sampler().seed_value()), analytic=True, film_rgb=True)
    path = gpu.srgb_coeff_path()

    def oracle_image(red, lamp):
        s2 = copy.deepcopy(sd)
        s2["bsdfs"][1]["reflectance"] = np.asarray(red, np.float32)
        s2["emitters"][0]["radiance"] = np.asarray(lamp, np.float32)
        return oracle.OracleScene(s2, naive=True, spectral_path=path).render_image(desc)[0].reshape(-1)

    red0, lamp0 = np.asarray(sd["bsdfs"][1]["reflectance"], np.float64), np.asarray(sd["emitters"][0]["radiance"], np.float64)
    di = dimage.cpu().numpy().astype(np.float64)
    for c in range(3):
        e = np.zeros(3); e[c] = 0.01
        fd = float(di @ (oracle_image(red0 + e, lamp0).astype(np.float64) - oracle_image(red0 - e, lamp0))) / 0.02
        assert abs(g_red[c] - fd) <= 2e-3 * max(abs(fd), 1e-2), (c, g_red[c], fd)
        fd = float(di @ (oracle_image(red0, lamp0 + e).astype(np.float64) - oracle_image(red0, lamp0 - e))) / 0.02
        assert abs(g_lamp[c] - fd) <= 2e-3 * max(abs(fd), 1e-2), (c, g_lamp[c], fd)
    # texels are not differentiable in the spectral variant, and say so
    tex_scene = gpu.Scene(scenes.cornell_box(texture=np.full((2, 2, 3), 0.5, np.float32)), variant="spectral", sensor=sensor, integrator=gpu.PathIntegrator(max_depth=4))
    assert not any(k.endswith(".data") for k in autodiff.traverse(tex_scene).keys())
    # inverse rendering in the spectral variant: recover the red wall's colour from an image of it
    with torch.no_grad():
        target = autodiff.render(scene, spp=64).clone()
    params = autodiff.traverse(scene)
    params.keep(["red.reflectance.value"])
    params["red.reflectance.value"] = [0.3, 0.3, 0.3]
    params.update()
    opt = autodiff.Adam(params, lr=0.05)
    first = last = None
    for it in range(30):
        img = autodiff.render(scene, spp=16, optimizer=opt)
        loss = ((img - target) ** 2).mean()
        loss.backward()
        opt.step()
        with torch.no_grad():
            params["red.reflectance.value"].clamp_(0.01, 0.99)
        first = float(loss.detach()) if first is None else first
        last = float(loss.detach())
    got = params["red.reflectance.value"].detach().cpu().numpy()
    assert last < 0.2 * first, (first, last)
    assert np.abs(got - red0).max() < 0.12, got
