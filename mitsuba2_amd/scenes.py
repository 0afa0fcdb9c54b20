"""Synthetic scene descriptions used by the benchmark, the smoke test and the parity tests.

The reference's scene data (``resources/data``: ``cbox.xml``, the OBJ/PLY meshes) is an empty
submodule, so the scenes are generated in code:

* ``cornell_box``   -- the classic Cornell-box measurements: 36 triangles, diffuse BSDFs, one
  2-triangle area light, perspective camera looking down +z.  Numbers are the published Cornell
  data; the layout follows what ``cbox.xml`` describes (``docs/src/inverse_rendering/diff_render.rst``).
* ``stairs``        -- the staircase mesh of the reference's kd-tree tests
  (``src/librender/tests/mesh_generation.py:27-59``), restated.
* ``bumpy_sphere``  -- a procedurally displaced sphere over a ground plane with an area light,
  for large-mesh traversal tests.

A scene description is a plain dict: ``{"meshes": [...], "bsdfs": [...], "emitters": [...]}`` with
numpy arrays, consumed by :class:`mitsuba2_amd.render.Scene` and by the test oracle binding.
"""
import numpy as np

F32 = np.float32


def _quad(p):
    """Two triangles (0,1,2), (0,2,3) from 4 corner points."""
    p = np.asarray(p, dtype=F32)
    return p, np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)


def _orient(pos, faces, towards=None, away_from=None):
    """Flip triangle windings so that geometric normals point towards / away from a point."""
    pos = np.asarray(pos, dtype=np.float64)
    out = faces.copy()
    for i, f in enumerate(faces):
        p0, p1, p2 = pos[f[0]], pos[f[1]], pos[f[2]]
        n = np.cross(p1 - p0, p2 - p0)
        c = (p0 + p1 + p2) / 3.0
        ref = (np.asarray(towards) - c) if towards is not None else (c - np.asarray(away_from))
        if np.dot(n, ref) < 0:
            out[i] = [f[0], f[2], f[1]]
    return out


def _box(quads, center):
    pos, faces = [], []
    for q in quads:
        p, f = _quad(q)
        faces.append(f + len(pos) * 4)
        pos.append(p)
    pos = np.concatenate(pos).astype(F32)
    faces = np.concatenate(faces).astype(np.uint32)
    return pos, _orient(pos, faces, away_from=center)


def cornell_box(texture=None):
    """36-triangle Cornell box.  Returns a scene dict; camera via :func:`cornell_box_sensor`.
    texture: optional (H, W, 3) linear-RGB array used as the diffuse albedo of the back wall and the floor
    (the five room quads then carry texcoords (0,0) (1,0) (1,1) (0,1)) -- the differentiable-rendering setup."""
    room_center = np.array([278.0, 274.4, 279.6])
    white, red, green = [0.725, 0.71, 0.68], [0.63, 0.065, 0.05], [0.14, 0.45, 0.091]
    meshes = []

    def add(pos, faces, bsdf, emitter=-1):
        meshes.append(dict(positions=np.ascontiguousarray(pos, dtype=F32), faces=np.ascontiguousarray(faces, dtype=np.uint32),
                           normals=None, texcoords=None, bsdf=bsdf, emitter=emitter))

    walls = {
        "floor": ([[552.8, 0, 0], [0, 0, 0], [0, 0, 559.2], [549.6, 0, 559.2]], 0),
        "ceiling": ([[556, 548.8, 0], [556, 548.8, 559.2], [0, 548.8, 559.2], [0, 548.8, 0]], 0),
        "back": ([[549.6, 0, 559.2], [0, 0, 559.2], [0, 548.8, 559.2], [556, 548.8, 559.2]], 0),
        "right": ([[0, 0, 559.2], [0, 0, 0], [0, 548.8, 0], [0, 548.8, 559.2]], 2),
        "left": ([[552.8, 0, 0], [549.6, 0, 559.2], [556, 548.8, 559.2], [556, 548.8, 0]], 1),
    }
    quad_uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=F32)
    for name, (q, bsdf) in walls.items():
        p, f = _quad(q)
        if texture is not None and name in ("floor", "back"):
            bsdf = 4
        add(p, _orient(p, f, towards=room_center), bsdf)
        if texture is not None:
            meshes[-1]["texcoords"] = quad_uv.copy()
    # area light, 0.5 below the ceiling (the cbox scene translates the luminaire by (0,-0.5,0))
    p, f = _quad([[343, 548.3, 227], [343, 548.3, 332], [213, 548.3, 332], [213, 548.3, 227]])
    add(p, _orient(p, f, towards=room_center), 3, emitter=0)
    short = [
        [[130, 165, 65], [82, 165, 225], [240, 165, 272], [290, 165, 114]],
        [[290, 0, 114], [290, 165, 114], [240, 165, 272], [240, 0, 272]],
        [[130, 0, 65], [130, 165, 65], [290, 165, 114], [290, 0, 114]],
        [[82, 0, 225], [82, 165, 225], [130, 165, 65], [130, 0, 65]],
        [[240, 0, 272], [240, 165, 272], [82, 165, 225], [82, 0, 225]],
        [[130, 0, 65], [290, 0, 114], [240, 0, 272], [82, 0, 225]],
    ]
    tall = [
        [[423, 330, 247], [265, 330, 296], [314, 330, 456], [472, 330, 406]],
        [[423, 0, 247], [423, 330, 247], [472, 330, 406], [472, 0, 406]],
        [[472, 0, 406], [472, 330, 406], [314, 330, 456], [314, 0, 456]],
        [[314, 0, 456], [314, 330, 456], [265, 330, 296], [265, 0, 296]],
        [[265, 0, 296], [265, 330, 296], [423, 330, 247], [423, 0, 247]],
        [[423, 0, 247], [472, 0, 406], [314, 0, 456], [265, 0, 296]],
    ]
    p, f = _box(short, [185.5, 82.5, 169.0]); add(p, f, 0)
    p, f = _box(tall, [368.5, 165.0, 351.25]); add(p, f, 0)
    bsdfs = [dict(type="diffuse", reflectance=np.array(c, dtype=F32)) for c in (white, red, green, [0.78, 0.78, 0.78])]
    if texture is not None:
        bsdfs.append(dict(type="diffuse", reflectance=dict(type="bitmap", data=np.ascontiguousarray(texture, dtype=F32))))
    emitters = [dict(type="area", radiance=np.array([18.387, 13.9873, 6.75357], dtype=F32))]
    return dict(meshes=meshes, bsdfs=bsdfs, emitters=emitters)


def look_at(origin, target, up):
    """Transform::look_at (include/mitsuba/core/transform.h:241-269), float32 arithmetic."""
    o, t, u = (np.asarray(v, dtype=F32) for v in (origin, target, up))

    def normalize(v):
        return (v * (F32(1.0) / np.sqrt(np.dot(v, v).astype(F32)))).astype(F32)

    d = normalize(normalize((t - o).astype(F32)))
    left = normalize(np.cross(u, d).astype(F32))
    new_up = np.cross(d, left).astype(F32)
    m = np.eye(4, dtype=F32)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
    return m


def cornell_box_sensor(width=256, height=256, spp=16, seed=0, max_depth=-1, rr_depth=5,
                       rfilter="gaussian", rfilter_param=None):
    """Sensor/film/sampler/integrator parameters of the synthetic cbox (dict of plain values)."""
    if rfilter_param is None and rfilter in ("gaussian", "box"):
        rfilter_param = 0.5
    return dict(to_world=look_at([278, 273, -800], [278, 273, -799], [0, 1, 0]), fov=39.3077, near_clip=10.0, far_clip=2800.0,
                width=width, height=height, crop=(0, 0, width, height), rfilter=rfilter, rfilter_param=rfilter_param,
                sample_count=spp, seed=seed, max_depth=max_depth, rr_depth=rr_depth)


def stairs(num_steps):
    """create_stairs (src/librender/tests/mesh_generation.py:27-59)."""
    size_step = 1.0 / num_steps
    v = np.zeros((4 * num_steps, 3))
    f = np.zeros((4 * num_steps - 2, 3))
    for i in range(num_steps):
        h = i * size_step
        s1 = i * size_step
        s2 = (i + 1) * size_step
        k = 4 * i
        v[k + 0] = [0.0, s1, h]
        v[k + 1] = [1.0, s1, h]
        v[k + 2] = [0.0, s2, h]
        v[k + 3] = [1.0, s2, h]
        f[k] = [k, k + 1, k + 2]
        f[k + 1] = [k + 1, k + 3, k + 2]
        if i < num_steps - 1:
            f[k + 2] = [k + 2, k + 3, k + 5]
            f[k + 3] = [k + 5, k + 4, k + 2]
    mesh = dict(positions=np.ascontiguousarray(v, dtype=F32), faces=np.ascontiguousarray(f, dtype=np.uint32), normals=None,
                texcoords=None, bsdf=0, emitter=-1)
    return dict(meshes=[mesh], bsdfs=[dict(type="diffuse", reflectance=np.array([0.5, 0.5, 0.5], dtype=F32))], emitters=[])


def bumpy_sphere(n_theta=64, n_phi=128, with_normals=True, seed=1):
    """Displaced sphere (2*n_theta*n_phi triangles approx.) over a ground quad, lit by an area light."""
    rng = np.random.RandomState(seed)
    th = np.linspace(0.0, np.pi, n_theta + 1)
    ph = np.linspace(0.0, 2 * np.pi, n_phi, endpoint=False)
    T, P = np.meshgrid(th, ph, indexing="ij")
    r = 1.0 + 0.08 * np.sin(5 * T) * np.cos(7 * P) + 0.02 * rng.rand(*T.shape)
    r[0, :] = r[0, 0]
    r[-1, :] = r[-1, 0]
    pos = np.stack([r * np.sin(T) * np.cos(P), r * np.cos(T) + 1.2, r * np.sin(T) * np.sin(P)], axis=-1).reshape(-1, 3)
    faces = []
    for i in range(n_theta):
        for j in range(n_phi):
            a = i * n_phi + j
            b = i * n_phi + (j + 1) % n_phi
            c = (i + 1) * n_phi + j
            d = (i + 1) * n_phi + (j + 1) % n_phi
            if i > 0:
                faces.append([a, b, c])
            if i < n_theta - 1:
                faces.append([b, d, c])
    faces = np.array(faces, dtype=np.uint32)
    faces = _orient(pos, faces, away_from=[0.0, 1.2, 0.0])
    normals = None
    if with_normals:
        n = pos - np.array([0.0, 1.2, 0.0])
        normals = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(F32)
    meshes = [dict(positions=np.ascontiguousarray(pos, dtype=F32), faces=faces, normals=normals, texcoords=None, bsdf=0, emitter=-1)]
    p, f = _quad([[-6, 0, -6], [-6, 0, 6], [6, 0, 6], [6, 0, -6]])
    meshes.append(dict(positions=p, faces=_orient(p, f, towards=[0, 5, 0]), normals=None, texcoords=None, bsdf=1, emitter=-1))
    p, f = _quad([[-1, 4, -1], [1, 4, -1], [1, 4, 1], [-1, 4, 1]])
    meshes.append(dict(positions=p, faces=_orient(p, f, towards=[0, 0, 0]), normals=None, texcoords=None, bsdf=2, emitter=0))
    bsdfs = [dict(type="diffuse", reflectance=np.array(c, dtype=F32)) for c in ([0.7, 0.4, 0.3], [0.5, 0.5, 0.5], [0, 0, 0])]
    emitters = [dict(type="area", radiance=np.array([20.0, 20.0, 20.0], dtype=F32))]
    return dict(meshes=meshes, bsdfs=bsdfs, emitters=emitters)


def matpreview(n_theta=256, n_phi=512, env_res=(256, 512)):
    """Material-preview style scene (BASELINE config 3 with the materials of a matpreview scene): the displaced sphere as a `roughplastic`
    object on a `checkerboard` ground plane, lit by a procedural sky `envmap` (smooth gradient + a sun lobe) and the area light."""
    sd = bumpy_sphere(n_theta, n_phi)
    sd["bsdfs"][0] = dict(type="roughplastic", alpha=0.1, distribution="ggx", diffuse_reflectance=np.array([0.1, 0.27, 0.36], dtype=F32), int_ior=1.49)
    sd["bsdfs"][1] = dict(type="diffuse", reflectance=dict(type="checkerboard", color0=[0.4, 0.4, 0.4], color1=[0.2, 0.2, 0.2],
                                                            to_uv=np.array([[8, 0, 0, 0], [0, 8, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=F32)))
    h, w = env_res
    v, u = np.meshgrid((np.arange(h) + 0.5) / h, (np.arange(w) + 0.5) / w, indexing="ij")
    theta, phi = v * np.pi, u * 2 * np.pi
    d = np.stack([np.sin(theta) * np.sin(phi), np.cos(theta), -np.sin(theta) * np.cos(phi)], -1)
    sun = np.array([0.4, 0.7, -0.6]); sun /= np.linalg.norm(sun)
    sky = 0.25 + 0.55 * np.clip(d[..., 1], 0, 1)[..., None] * np.array([0.55, 0.7, 1.0]) + 0.08 * np.clip(-d[..., 1], 0, 1)[..., None] * np.array([0.5, 0.45, 0.4])
    sky = sky + 12.0 * np.exp(-(1 - np.clip(d @ sun, -1, 1)) * 200.0)[..., None] * np.array([1.0, 0.9, 0.75])
    sd["emitters"].append(dict(type="envmap", data=np.ascontiguousarray(sky, dtype=F32), scale=1.0))
    return sd


def bumpy_sphere_sensor(width=128, height=96, spp=8, seed=0, max_depth=-1, rr_depth=5):
    return dict(to_world=look_at([0, 2.5, -5.5], [0, 1.0, 0], [0, 1, 0]), fov=40.0, near_clip=0.01, far_clip=1e4,
                width=width, height=height, crop=(0, 0, width, height), rfilter="gaussian", rfilter_param=0.5,
                sample_count=spp, seed=seed, max_depth=max_depth, rr_depth=rr_depth)
