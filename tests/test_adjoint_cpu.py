"""The oracle's adjoint (reverse-mode derivative of Image = RGB / (W + 1e-8) w.r.t. diffuse reflectances) against
central finite differences of the oracle's own forward render with identical sampler seeds.  The reference's
gradients come from Enoki's autodiff, which is absent: no golden gradient exists in-tree ("parity unpinned"), so the
derivative is pinned by the forward pass it differentiates."""
import numpy as np
import pytest

from mitsuba2_amd import scenes


def _setup(oracle, rfilter, max_depth, tex_res=(4, 5)):
    rng = np.random.RandomState(1)
    tex = (0.3 + 0.5 * rng.rand(tex_res[0], tex_res[1], 3)).astype(np.float32)
    sd = scenes.cornell_box(texture=tex)
    p = scenes.cornell_box_sensor(12, 10, 4, seed=5, max_depth=max_depth, rfilter=rfilter)
    desc = oracle.make_desc(p, analytic=True, film_rgb=True)
    return sd, tex, desc


@pytest.mark.parametrize("rfilter,max_depth", [("box", 3), ("gaussian", 4), ("box", 8)])
def test_oracle_adjoint_matches_finite_differences(oracle, rfilter, max_depth):
    sd, tex, desc = _setup(oracle, rfilter, max_depth)
    S = oracle.OracleScene(sd, naive=True)
    image, film = S.render_image(desc)
    rng = np.random.RandomState(2)
    dimage = rng.randn(*image.shape).astype(np.float32)          # arbitrary cotangent: loss = <dimage, image>
    n_shapes = len(sd["meshes"])
    gs, gt = S.render_adjoint(desc, dimage, film, n_shapes, tex.size)
    gt = gt.reshape(tex.shape)
    loss = lambda img: float(np.sum(img.astype(np.float64) * dimage))

    # (a) texels: central differences (the image is multilinear in the texels)
    eps = 1e-2
    checked = 0
    for (ty, tx, c) in [(0, 0, 0), (1, 2, 1), (3, 4, 2), (2, 1, 0), (1, 3, 2)]:
        tp, tm = tex.copy(), tex.copy()
        tp[ty, tx, c] += eps; tm[ty, tx, c] -= eps
        S.update_texture(4, tp); lp = loss(S.render_image(desc)[0])
        S.update_texture(4, tm); lm = loss(S.render_image(desc)[0])
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - gt[ty, tx, c]) <= 2e-3 * max(abs(fd), abs(gt).max()) + 1e-5, (ty, tx, c, fd, gt[ty, tx, c])
        checked += abs(fd) > 1e-6
    S.update_texture(4, tex)
    assert checked >= 3

    # (b) constant reflectance of the red wall (bsdf 1 -> the "left" quad) and of the white bsdf (ceiling, right-side boxes)
    for bsdf, c in ((1, 0), (1, 2), (0, 1)):
        base = np.array(sd["bsdfs"][bsdf]["reflectance"], np.float32)
        shapes = [i for i, m in enumerate(sd["meshes"]) if m["bsdf"] == bsdf]
        vp, vm = base.copy(), base.copy()
        vp[c] += eps; vm[c] -= eps
        S.set_bsdf_reflectance(bsdf, vp); lp = loss(S.render_image(desc)[0])
        S.set_bsdf_reflectance(bsdf, vm); lm = loss(S.render_image(desc)[0])
        S.set_bsdf_reflectance(bsdf, base)
        fd = (lp - lm) / (2 * eps)
        g = gs[shapes, c].sum()
        # a BSDF shared by several vertices of one path makes the image polynomial, not linear: FD has O(eps^2) error
        assert abs(fd - g) <= 5e-3 * max(abs(fd), 1e-3) + 1e-5, (bsdf, c, fd, g)


@pytest.mark.parametrize("rfilter,max_depth,two_lights", [("box", 3, False), ("gaussian", 6, True)])
def test_oracle_emitter_radiance_gradient(oracle, rfilter, max_depth, two_lights):
    """'shape.emitter.radiance.value' (diff_render.rst:76): the image is linear in every area light's radiance, so central
    differences are exact up to rounding; the second case has two lights (emitter selection, scene.cpp:141-189) and
    Russian roulette"""
    sd, tex, desc = _setup(oracle, rfilter, max_depth)
    if two_lights:
        sd["emitters"] = list(sd["emitters"]) + [dict(type="area", radiance=np.array([2.0, 3.0, 4.0], np.float32))]
        sd["meshes"][7] = dict(sd["meshes"][7], emitter=1)                 # one of the boxes glows too
    S = oracle.OracleScene(sd, naive=True)
    image, film = S.render_image(desc)
    dimage = np.random.RandomState(3).randn(*image.shape).astype(np.float32)
    _, _, ge = S.render_adjoint(desc, dimage, film, len(sd["meshes"]), tex.size, n_emitters=len(sd["emitters"]))
    loss = lambda img: float(np.sum(img.astype(np.float64) * dimage))
    assert np.abs(ge).min() > 0
    for e in range(len(sd["emitters"])):
        base = np.array(sd["emitters"][e]["radiance"], np.float32)
        for c in range(3):
            eps = 0.05 * base[c]
            vp, vm = base.copy(), base.copy()
            vp[c] += eps; vm[c] -= eps
            S.set_emitter_radiance(e, vp); lp = loss(S.render_image(desc)[0])
            S.set_emitter_radiance(e, vm); lm = loss(S.render_image(desc)[0])
            S.set_emitter_radiance(e, base)
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - ge[e, c]) <= 2e-3 * max(abs(fd), np.abs(ge).max()) + 1e-5, (e, c, fd, ge[e, c])


def _envmap_scene(with_area):
    """open Cornell box (no ceiling / back wall) with a glossy and a glass box under an `envmap` emitter, optionally the area light"""
    cb = scenes.cornell_box()
    keep = [i for i, m in enumerate(cb["meshes"]) if i not in (1, 2) and (with_area or m.get("emitter", -1) < 0)]
    cb["meshes"] = [dict(cb["meshes"][i]) for i in keep]
    cb["bsdfs"] = list(cb["bsdfs"]) + [{"type": "roughconductor", "alpha": 0.2, "distribution": "ggx", "eta": 0.0, "k": 1.0}, {"type": "dielectric"}]
    cb["meshes"][-1]["bsdf"] = len(cb["bsdfs"]) - 2
    cb["meshes"][-2]["bsdf"] = len(cb["bsdfs"]) - 1
    rng = np.random.RandomState(7)
    img = rng.uniform(0.2, 1.0, size=(6, 10, 3)).astype(np.float32)
    img[1:3, 6:8] += 8.0
    env = {"type": "envmap", "id": "my_envmap", "data": img, "scale": 0.8, "to_world": scenes.look_at([0, 0, 0], [1, 0.2, 0.3], [0, 1, 0])}
    if with_area:
        cb["emitters"] = [env] + list(cb["emitters"])
        for m in cb["meshes"]:
            if m.get("emitter", -1) >= 0:
                m["emitter"] = 1
    else:
        cb["emitters"] = [env]
    return cb, img


@pytest.mark.parametrize("with_area,rfilter,max_depth", [(False, "box", 4), (True, "gaussian", 6)])
def test_oracle_envmap_gradient(oracle, with_area, rfilter, max_depth):
    """'my_envmap.data' (docs/examples/10_inverse_rendering/invert_bunny.py): with the sampling distribution held fixed the image
    is linear in the texels, so central differences of the oracle's forward render are exact up to rounding -- through glossy and
    dielectric BSDFs, MIS with BSDF sampling, Russian roulette and (second case) emitter selection"""
    sd, img = _envmap_scene(with_area)
    p = scenes.cornell_box_sensor(14, 12, 8, seed=9, max_depth=max_depth, rfilter=rfilter)
    desc = oracle.make_desc(p, analytic=True, film_rgb=True)
    S = oracle.OracleScene(sd, naive=True)
    image, film = S.render_image(desc)
    dimage = np.random.RandomState(3).randn(*image.shape).astype(np.float32)
    g = S.render_adjoint_envmap(desc, dimage, film, img.shape)
    loss = lambda im: float(np.sum(im.astype(np.float64) * dimage))
    assert (np.abs(g) > 0).mean() > 0.5
    checked = 0
    for (ty, tx, c) in [(1, 6, 0), (2, 7, 1), (0, 0, 2), (4, 3, 1), (5, 9, 0), (3, 5, 2)]:
        h = 0.25
        tp, tm = img.copy(), img.copy()
        tp[ty, tx, c] += h; tm[ty, tx, c] -= h
        S.update_envmap(tp, rebuild_warp=False); lp = loss(S.render_image(desc)[0])
        S.update_envmap(tm, rebuild_warp=False); lm = loss(S.render_image(desc)[0])
        fd = (lp - lm) / (2 * h)
        assert abs(fd - g[ty, tx, c]) <= 2e-3 * max(abs(fd), np.abs(g).max()) + 1e-5, (ty, tx, c, fd, g[ty, tx, c])
        checked += abs(fd) > 1e-6
    assert checked >= 4
    S.update_envmap(img, rebuild_warp=False)
    assert np.array_equal(S.render_image(desc)[0], image)
    # parameters_changed() (envmap.cpp:220-253): rebuilding the distribution from the same texels changes nothing
    S.update_envmap(img, rebuild_warp=True)
    assert np.array_equal(S.render_image(desc)[0], image)


def test_adjoint_argument_checks(oracle):
    sd, tex, desc = _setup(oracle, "box", 3)
    S = oracle.OracleScene(sd, naive=True)
    image, film = S.render_image(desc)
    desc.max_depth = -1
    with pytest.raises(RuntimeError):
        S.render_adjoint(desc, np.zeros_like(image), film, len(sd["meshes"]), tex.size)


def _twosided_cbox(tex=None):
    """Cornell box whose red wall is wound the other way (every hit on it is a back-face hit) with all diffuse BSDFs inside `twosided`"""
    sd = scenes.cornell_box(texture=tex)
    sd["bsdfs"] = [dict(type="twosided", id=b.get("id", "b%d" % i), bsdf=dict(b, id="inner%d" % i)) if b.get("type", "diffuse") == "diffuse" else b
                   for i, b in enumerate(sd["bsdfs"])]
    wall = next(i for i, m in enumerate(sd["meshes"]) if m["bsdf"] == 1)
    sd["meshes"][wall] = dict(sd["meshes"][wall], faces=np.ascontiguousarray(np.asarray(sd["meshes"][wall]["faces"]).reshape(-1, 3)[:, ::-1]))
    return sd, wall


def test_oracle_adjoint_twosided_diffuse(oracle):
    """`twosided` around `diffuse` (twosided.cpp:94-175) in the adjoint path replay: finite differences of the forward render, and the
    flipped wall really is lit (a one-sided wall seen from behind is black and has a zero gradient)"""
    sd, wall = _twosided_cbox()
    p = scenes.cornell_box_sensor(12, 10, 4, seed=5, max_depth=4, rfilter="box")
    desc = oracle.make_desc(p, analytic=True, film_rgb=True)
    S = oracle.OracleScene(sd, naive=True)
    image, film = S.render_image(desc)
    dimage = np.random.RandomState(2).randn(*image.shape).astype(np.float32)
    gs, _ = S.render_adjoint(desc, dimage, film, len(sd["meshes"]), 0)
    loss = lambda img: float(np.sum(img.astype(np.float64) * dimage))
    eps = 1e-2
    assert np.abs(gs[wall]).max() > 1e-3                     # the back-facing wall scatters
    for bsdf, c in ((1, 0), (1, 2), (0, 1)):
        base = np.array(sd["bsdfs"][bsdf]["bsdf"]["reflectance"], np.float32)
        shapes = [i for i, m in enumerate(sd["meshes"]) if m["bsdf"] == bsdf]
        vp, vm = base.copy(), base.copy()
        vp[c] += eps; vm[c] -= eps
        S.set_bsdf_reflectance(bsdf, vp); lp = loss(S.render_image(desc)[0])
        S.set_bsdf_reflectance(bsdf, vm); lm = loss(S.render_image(desc)[0])
        S.set_bsdf_reflectance(bsdf, base)
        fd = (lp - lm) / (2 * eps)
        g = gs[shapes, c].sum()
        assert abs(fd - g) <= 5e-3 * max(abs(fd), 1e-3) + 1e-5, (bsdf, c, fd, g)
    # the same box with one-sided BSDFs: the wall seen from behind neither scatters nor receives a gradient
    one = scenes.cornell_box()
    one["meshes"][wall] = sd["meshes"][wall]
    S1 = oracle.OracleScene(one, naive=True)
    image1, film1 = S1.render_image(desc)
    gs1, _ = S1.render_adjoint(desc, dimage, film1, len(one["meshes"]), 0)
    assert np.all(gs1[wall] == 0.0) and not np.allclose(image1, image)
    # a scene with any other BSDF is refused
    gl = scenes.cornell_box(); gl["bsdfs"] = list(gl["bsdfs"]); gl["bsdfs"][1] = {"type": "conductor"}
    with pytest.raises(RuntimeError, match="adjoint failed"):
        oracle.OracleScene(gl, naive=True).render_adjoint(desc, dimage, film, len(gl["meshes"]), 0)


def test_oracle_parameter_adjoint_matches_finite_differences(oracle):
    """mo_render_adjoint_param (the checker of mtsamd_render_adjoint_param: forward-mode derivative carried beside the replayed path,
    detached sampling) against central differences of the oracle's own render with identical seeds, for parameters the sampled directions
    do not depend on -- there the detached and the attached estimator coincide sample by sample: the complex IOR and the specular
    reflectance of a rough conductor, and a diffuse reflectance in the same (general) scene."""
    sd = scenes.cornell_box()
    rough = {"type": "roughconductor", "alpha": 0.3, "distribution": "ggx", "eta": [0.2, 0.92, 1.1], "k": [3.9, 2.45, 2.14], "specular_reflectance": [0.9, 0.8, 0.7]}
    sd["bsdfs"] = list(sd["bsdfs"]) + [rough]
    sd["meshes"][6] = dict(sd["meshes"][6], bsdf=len(sd["bsdfs"]) - 1)
    p = scenes.cornell_box_sensor(16, 12, 8, seed=9, max_depth=4, rfilter="box")
    desc = oracle.make_desc(p, analytic=True, film_rgb=True)
    rng = np.random.RandomState(3)
    dimage = rng.uniform(0.2, 1.0, (12, 16, 3)).astype(np.float32)
    loss = lambda img: float(np.sum(img.astype(np.float64) * dimage))

    def render_loss(bsdfs):
        S = oracle.OracleScene(dict(sd, bsdfs=bsdfs))
        return loss(S.render_image(desc)[0])

    S0 = oracle.OracleScene(sd)
    image, film = S0.render_image(desc)
    tall = [i for i, m in enumerate(sd["meshes"]) if m["bsdf"] == len(sd["bsdfs"]) - 1]
    white = [i for i, m in enumerate(sd["meshes"]) if m["bsdf"] == 0]
    for shapes, bsdf_index, name, kind, comp, eps in ((tall, len(sd["bsdfs"]) - 1, "k", 3, 1, 0.05), (tall, len(sd["bsdfs"]) - 1, "eta", 2, 0, 0.02),
                                                      (tall, len(sd["bsdfs"]) - 1, "specular_reflectance", 1, 2, 0.02), (white, 0, "reflectance", 0, 1, 0.01)):
        g = S0.render_adjoint_param(desc, dimage, film, shapes, kind, comp, 0.01 * max(abs(sd["bsdfs"][bsdf_index][name][comp]), 0.05))
        lp_lm = []
        for sign in (+1.0, -1.0):
            bs = [dict(b) for b in sd["bsdfs"]]
            v = np.array(bs[bsdf_index][name], np.float32).copy()
            v[comp] += sign * eps
            bs[bsdf_index][name] = v
            lp_lm.append(render_loss(bs))
        fd = (lp_lm[0] - lp_lm[1]) / (2 * eps)
        assert abs(fd) > 1e-4 and abs(g - fd) <= 1e-2 * abs(fd) + 1e-5, (name, comp, g, fd)
