"""Pins the CPU oracle against the known-answer values held by the reference's own tests
(SURVEY.md section 8(c)).  Runs without a GPU."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_binding as ob


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_tea_float32_kat():
    # src/libcore/tests/test_random.py:6-16
    L = ob.lib()
    kat = {(1, 1): 0.5424730777740479, (1, 2): 0.5079904794692993, (1, 3): 0.4171961545944214, (1, 4): 0.008385419845581055,
           (1, 5): 0.8085528612136841, (2, 1): 0.6939879655838013, (3, 1): 0.6978365182876587, (4, 1): 0.4897364377975464}
    for (a, b), e in kat.items():
        assert L.mo_kat_tea_float32(a, b, 4) == e


def test_tea_float64_kat():
    # src/libcore/tests/test_random.py:19-29
    L = ob.lib()
    kat = {(1, 1): 0.5424730799533735, (1, 2): 0.5079905082233922, (1, 3): 0.4171962610608142, (1, 4): 0.008385529523330604,
           (1, 5): 0.80855288317879, (2, 1): 0.6939880404156831, (3, 1): 0.6978365636630994, (4, 1): 0.48973647949223253}
    for (a, b), e in kat.items():
        assert L.mo_kat_tea_float64(a, b, 4) == e


def test_tea64_flavours_agree_on_small_operands():
    # the 64-bit-operand flavour has no in-tree KAT ("parity unpinned"); its low word must still
    # match the 32-bit algorithm as long as nothing overflows 32 bits in the first round
    L = ob.lib()
    assert L.mo_kat_tea64_u32(1, 1, 4) & 0xffffffff == L.mo_kat_tea64_u32(1, 1, 4) % (1 << 32)
    assert L.mo_kat_tea64_u32(1, 1, 4) >> 32 == L.mo_kat_tea32(1, 1, 4)
    # zero rounds is the identity packing
    assert L.mo_kat_tea64_u64(5, 7, 0) == 5 + (7 << 32)
    assert L.mo_kat_tea64_u32(5, 7, 0) == 5 + (7 << 32)


def test_pcg32_published_sequences():
    # PCG32 lives in the absent Enoki submodule; pinned to the published demo sequence
    # (pcg32_srandom(42, 54)) and to the default-seeded float stream
    u = np.zeros(6, np.uint32)
    ob.lib().mo_kat_pcg32(42, 54, 6, _p(u), None)
    assert [hex(x) for x in u] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]
    f = np.zeros(3, np.float32)
    ob.lib().mo_kat_pcg32(0x853c49e6748fea9b, 0xda3e39cb94b95bdb, 3, None, _p(f))
    assert f.tolist() == [0.10837864875793457, 0.9069600105285645, 0.4066922664642334]
    # src/samplers/tests/test_independent.py:28-33: next_1d is the PCG32 float stream, in [0,1)
    f = np.zeros(1000, np.float32)
    ob.lib().mo_kat_pcg32(7, 0xda3e39cb94b95bdb, 1000, None, _p(f))
    assert (f >= 0).all() and (f < 1).all() and abs(f.mean() - 0.5) < 0.05


def test_spiral_order():
    # src/librender/tests/test_spiral.py:49-86
    out = np.zeros((200, 5), np.int64)
    n = ob.lib().mo_kat_spiral(15, 12, 0, 0, 32, 1, 200, _p(out))
    assert n == 1 and out[0].tolist() == [0, 0, 15, 12, 0]
    n = ob.lib().mo_kat_spiral(318, 322, 0, 0, 32, 1, 200, _p(out))
    assert n == 110
    c, w = np.array([160, 160]), 32
    steps = [(0, 0), (1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (2, -1), (2, 0), (2, 1)]
    for i, s in enumerate(steps):
        assert out[i, :2].tolist() == (c + np.array(s) * w).tolist()
        assert out[i, 2:4].tolist() == [w, w]
    # all blocks tile the film exactly once, ids are 0..n-1
    cover = np.zeros((322, 318), np.int32)
    for ox, oy, bw, bh, bid in out[:n]:
        cover[oy:oy + bh, ox:ox + bw] += 1
    assert (cover == 1).all() and sorted(out[:n, 4].tolist()) == list(range(110))
    # multi-pass ids: block_counter + (remaining_passes - 1) * block_count (spiral.cpp:43)
    n2 = ob.lib().mo_kat_spiral(64, 64, 0, 0, 32, 2, 200, _p(out))
    assert n2 == 8 and out[:8, 4].tolist() == [4, 5, 6, 7, 0, 1, 2, 3]


def test_morton_decode():
    xy = np.zeros((16, 2), np.uint32)
    ob.lib().mo_kat_morton(16, _p(xy))
    assert xy[:8].tolist() == [[0, 0], [1, 0], [0, 1], [1, 1], [2, 0], [3, 0], [2, 1], [3, 1]]


def test_discrete_distribution():
    # src/libcore/tests/test_distr_1d.py:35-103 ([1,3,2]: cdf [1,4,6]; sample(-1,0,1,2)=[0,0,2,2])
    pmf = np.array([1, 3, 2], np.float32)
    cdf = np.zeros(3, np.float32)
    vals = np.array([-1, 0, 1, 2], np.float32)
    idx = np.zeros(4, np.uint32)
    s = ob.lib().mo_kat_distr(3, _p(pmf), _p(cdf), 4, _p(vals), _p(idx), None)
    assert s == 6.0 and cdf.tolist() == [1, 4, 6] and idx.tolist() == [0, 0, 2, 2]
    eps = 1e-7
    vals = np.array([1 / 6.0 - eps, 1 / 6.0 + eps, 4 / 6.0 - eps, 4 / 6.0 + eps], np.float32)
    ob.lib().mo_kat_distr(3, _p(pmf), _p(cdf), 4, _p(vals), _p(idx), None)
    assert idx.tolist() == [0, 1, 1, 2]
    # sample_reuse: (value - cdf[i-1]/sum) / (pmf[i]/sum)
    vals = np.array([0.5], np.float32)
    reused = np.zeros(1, np.float32)
    ob.lib().mo_kat_distr(3, _p(pmf), _p(cdf), 1, _p(vals), _p(idx), _p(reused))
    assert idx[0] == 1 and abs(reused[0] - (0.5 - 1 / 6) / 0.5) < 1e-6
    # leading / trailing zero-probability bins are never selected (m_valid, distr_1d.h:75-81)
    pmf = np.array([0, 1, 0], np.float32)
    vals = np.array([0.0, 0.5, 1.0], np.float32)
    idx = np.zeros(3, np.uint32)
    ob.lib().mo_kat_distr(3, _p(pmf), _p(cdf), 3, _p(vals), _p(idx), None)
    assert idx.tolist() == [1, 1, 1]
    assert ob.lib().mo_kat_distr(2, _p(np.zeros(2, np.float32)), _p(cdf), 0, None, None, None) == -1.0


def _warp(which, pts):
    pts = np.asarray(pts, np.float32).reshape(-1, 2)
    out = np.zeros((pts.shape[0], 3), np.float32)
    ob.lib().mo_kat_warp(which, pts.shape[0], _p(np.ascontiguousarray(pts[:, 0])), _p(np.ascontiguousarray(pts[:, 1])), _p(out))
    return out


def test_warps():
    # src/libcore/tests/test_warp.py:68-92
    d = _warp(0, [[0, 0], [0.5, 0.5], [1, 1]])
    s = 1 / math.sqrt(2)
    assert np.allclose(d[0, :2], [-s, -s], atol=1e-6) and np.allclose(d[1, :2], [0, 0], atol=1e-7)
    assert np.allclose(d[2, :2], [s, s], atol=1e-6)
    t = _warp(2, [[0, 0], [0, 0.1], [0, 1], [1, 0], [1, 0.5], [1, 1]])
    assert np.allclose(t[:, :2], [[0, 0], [0, 0.1], [0, 1], [1, 0], [1, 0], [1, 0]], atol=1e-6)
    h = _warp(1, [[0.5, 0.5], [0.5, 0]])
    assert np.allclose(h[0], [0, 0, 1], atol=1e-6)
    assert np.allclose(h[1], [0, -1, 0], atol=1e-6)
    rng = np.random.RandomState(0)
    h = _warp(1, rng.rand(1000, 2))
    assert np.allclose(np.linalg.norm(h, axis=1), 1, atol=1e-5) and (h[:, 2] >= 0).all()
    # cosine-weighted: E[z] = 2/3
    assert abs(h[:, 2].mean() - 2 / 3) < 0.03


def test_coordinate_system():
    rng = np.random.RandomState(1)
    for n in list(rng.randn(50, 3)) + [[0, 0, 1], [0, 0, -1], [1, 0, 0]]:
        n = np.asarray(n, np.float32)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        s, t = np.zeros(3, np.float32), np.zeros(3, np.float32)
        ob.lib().mo_kat_coordinate_system(_p(n), _p(s), _p(t))
        assert abs(np.dot(s, t)) < 1e-5 and abs(np.dot(s, n)) < 1e-5 and abs(np.dot(t, n)) < 1e-5
        assert np.allclose(np.cross(s, t), n, atol=1e-5)


def test_diffuse_closed_form():
    # src/bsdfs/tests/test_diffuse.py:16-38: pdf = cos/pi, eval = 0.5 cos/pi for 20 angles
    refl = np.array([0.5, 0.5, 0.5], np.float32)
    wi = np.array([0, 0, 1], np.float32)
    out = [np.zeros(3, np.float32), np.zeros(1, np.float32), np.zeros(3, np.float32), np.zeros(1, np.float32), np.zeros(3, np.float32)]
    for i in range(20):
        theta = i / 19.0 * (math.pi / 2)
        wo = np.array([math.sin(theta), 0, math.cos(theta)], np.float32)
        ob.lib().mo_kat_diffuse(_p(refl), _p(wi), _p(wo), _p(np.array([0.3, 0.6], np.float32)), *[_p(o) for o in out])
        assert np.allclose(out[1][0], wo[2] / math.pi, atol=1e-6)
        assert np.allclose(out[0], 0.5 * wo[2] / math.pi, atol=1e-6)
    # one-sided (FrontSide only): wi below the surface -> zero eval/pdf and a zero-weight sample
    wi = np.array([0, 0, -1], np.float32)
    ob.lib().mo_kat_diffuse(_p(refl), _p(wi), _p(np.array([0, 0, 1], np.float32)), _p(np.array([0.3, 0.6], np.float32)), *[_p(o) for o in out])
    assert out[1][0] == 0 and (out[0] == 0).all() and (out[4] == 0).all() and out[3][0] == 0
    # sample: weight = reflectance, pdf = cos/pi
    wi = np.array([0, 0, 1], np.float32)
    ob.lib().mo_kat_diffuse(_p(refl), _p(wi), _p(np.array([0, 0, 1], np.float32)), _p(np.array([0.3, 0.6], np.float32)), *[_p(o) for o in out])
    assert np.allclose(out[4], refl) and abs(out[3][0] - out[2][2] / math.pi) < 1e-7


def test_rfilter_tables():
    # gaussian: radius 4*sigma, border ceil(r - 0.5 - 2eps); box: radius 0.5+eps, border 0
    tbl, radius, border = ob.rfilter_table(0, 0.5)
    assert radius == 2.0 and border == 2 and tbl[31] == 0
    alpha = -1 / (2 * 0.25)
    for i in range(31):
        x = 2.0 * i / 31
        assert abs(tbl[i] - max(0, math.exp(alpha * x * x) - math.exp(alpha * 4))) < 1e-6
    tbl, radius, border = ob.rfilter_table(1, 0.5)
    assert border == 0 and abs(radius - 0.5) < 1e-3 and (tbl[:31] == 1).all()
    # src/librender/tests/test_imageblock.py:39-44: border_size() for stddev 15
    _, radius, border = ob.rfilter_table(0, 15.0)
    assert radius == 60.0 and border == 60


def test_rfilter_spot_values():
    """src/rfilters/tests/test_rfilter.py:9-62 (test01 .. test06): eval_discretized spot checks of every filter"""
    def disc(kind, x, param=0.0, param2=0.0):
        tbl, radius, _ = ob.rfilter_table(kind, param, param2)
        return float(tbl[min(int(abs(np.float32(x) * np.float32(31.0 / radius))), 31)])
    assert disc(1, 0.49, 0.5) == 1 and disc(1, 0.51, 0.5) == 0                                        # box
    assert abs(disc(0, 0.2, 0.5) - 0.9227) < 8e-3 and disc(0, 2.1, 0.5) == 0                         # gaussian
    assert abs(disc(5, 1.4, 3) - (-0.14668)) < 1e-2 and disc(5, 3.1, 3) == 0                         # lanczos
    assert abs(disc(4, 0.0, 1 / 3, 1 / 3) - 0.8888) < 1e-3 and disc(4, 2.1, 1 / 3, 1 / 3) == 0       # mitchell
    assert abs(disc(3, 0.0) - 0.9765) < 5e-2 and disc(3, 2.1) == 0                                   # catmullrom
    assert abs(disc(2, 0.1) - 0.903) < 5e-2 and disc(2, 1.1) == 0                                    # tent
    for kind, radius, border in ((2, 1.0, 1), (3, 2.0, 2), (4, 2.0, 2), (5, 3.0, 3)):
        _, r, b = ob.rfilter_table(kind, 3.0 if kind == 5 else 1 / 3, 1 / 3)
        assert r == radius and b == border                                                            # rfilter.h:72-73


def test_imageblock_box_put_lands_in_one_pixel():
    # src/librender/tests/test_imageblock.py:52-75 (test02): centre samples, box filter, 4 channels
    w, h, ch = 10, 5, 4
    ref = (3.14 * np.arange(h * w * ch)).reshape(h, w, ch).astype(np.float32)
    pos = np.array([[x + 0.5, y + 0.5] for y in range(h) for x in range(w)], np.float32)
    data = ob.imageblock_put(w, h, 0, 0, ch, 1, 0.5, True, pos, ref.reshape(-1, ch))
    assert np.allclose(data, ref, atol=1e-6)


def test_imageblock_gaussian_footprint():
    # src/librender/tests/test_imageblock.py:146-220 (test05): footprint = product of eval_discretized over
    # ceil(pos - r) .. floor(pos + r), block has a border of border_size() pixels
    size = 12
    tbl, radius, border = ob.rfilter_table(0, 0.5)
    rng = np.random.RandomState(3)
    positions = np.array([[5, 6], [0, 1], [5, 6], [1, 11], [11, 11], [0, 1], [2, 5], [4, 1], [0, 11], [5, 4]], np.float64)
    positions += rng.uniform(0, 0.95, positions.shape)
    n = positions.shape[0]
    values = np.zeros((n, 5), np.float32)
    values[:, :3] = np.arange(n * 3).reshape(n, 3)
    values[:, 3:] = 1
    ref = np.zeros((size + 2 * border, size + 2 * border, 5))
    ev = lambda x: tbl[min(int(abs(np.float32(x) * np.float32(31 / radius))), 31)]
    r = int(math.ceil(radius))
    for i in range(n):
        pos = positions[i].astype(np.float32) - np.float32(0.5) + border
        lo, hi = np.ceil(pos - r).astype(int), np.floor(pos + r).astype(int)
        for dy in range(lo[1], hi[1] + 1):
            for dx in range(lo[0], hi[0] + 1):
                if dx < 0 or dy < 0 or dx >= ref.shape[1] or dy >= ref.shape[0]:
                    continue
                ref[dy, dx] += ev(dx - pos[0]) * ev(dy - pos[1]) * values[i]
    data = ob.imageblock_put(size, size, 0, 0, 5, 0, 0.5, True, positions, values)
    assert data.shape == ref.shape
    assert np.allclose(data, ref, atol=1e-5)
    # invalid samples are dropped (imageblock.cpp:85-109)
    bad = values.copy(); bad[0, 0] = -1; bad[1, 1] = np.nan
    d2 = ob.imageblock_put(size, size, 0, 0, 5, 0, 0.5, True, positions, bad)
    d3 = ob.imageblock_put(size, size, 0, 0, 5, 0, 0.5, True, positions[2:], values[2:])
    assert np.allclose(d2, d3, atol=1e-6)


def test_film_develop():
    # hdrfilm.cpp:278-299: RGB = M * XYZ / W; round trip through srgb_to_xyz
    rgb = np.array([[0.2, 0.5, 0.7]], np.float32)
    M = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], np.float32)
    xyz = rgb @ M.T
    w = 2.5
    px = np.concatenate([xyz * w, [[0.8 * w, w]]], axis=1).astype(np.float32)
    out = ob.film_develop(px)
    assert np.allclose(out[0, :3], rgb[0], atol=1e-4) and abs(out[0, 3] - 0.8) < 1e-6


def test_stairs_kat(oracle):
    # src/librender/tests/test_kdtrees.py:26-59: t = 2 - floor(y*n)/n, shadow ray == naive validity,
    # accelerated query == brute force exactly
    from mitsuba2_amd import scenes
    n_steps = 20
    S = oracle.OracleScene(scenes.stairs(n_steps))
    n = 128
    inv_n = 1.0 / (n - 1)
    xs, ys = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
    o = np.stack([xs.ravel() * inv_n, ys.ravel() * inv_n, np.full(xs.size, 2.0)], axis=1).astype(np.float32)
    d = np.tile(np.array([0, 0, -1], np.float32), (o.shape[0], 1))
    mint, maxt = np.zeros(o.shape[0], np.float32), np.full(o.shape[0], 100, np.float32)
    t_naive, prim_n, _, u_n, v_n = S.ray_intersect(o, d, mint, maxt, naive=True)
    t_bvh, prim_b, _, u_b, v_b = S.ray_intersect(o, d, mint, maxt, naive=False)
    shadow = S.ray_test(o, d, mint, maxt, naive=False)
    step_idx = np.floor((ys.ravel() * inv_n) * n_steps)
    expected = 2.0 - step_idx / n_steps
    assert shadow.all() and np.isfinite(t_naive).all()
    assert np.allclose(t_naive, expected, atol=1e-6)
    assert (t_naive == t_bvh).all() and (prim_n == prim_b).all() and (u_n == u_b).all() and (v_n == v_b).all()


def test_mesh_area_kat(oracle):
    # src/librender/tests/test_mesh.py:10-31: 3 vertices, faces [0,1,2] and [1,2,0]: surface_area = 0.96
    sd = dict(meshes=[dict(positions=np.array([[0.0, 0.0, 0.0], [1.0, 0.2, 0.0], [0.2, 1.0, 0.0]], np.float32),
                           faces=np.array([[0, 1, 2], [1, 2, 0]], np.uint32), normals=None, texcoords=None, bsdf=0, emitter=0)],
              bsdfs=[dict(type="diffuse", reflectance=[0.5, 0.5, 0.5])], emitters=[dict(type="area", radiance=[1, 1, 1])])
    S = oracle.OracleScene(sd)
    assert abs(oracle.lib().mo_scene_emitter_area(S.h, 0) - 0.96) < 1e-6


def test_emitter_sampling_consistency(oracle):
    # sample_emitter_direction / pdf_emitter_direction agree (area.cpp:103-125) and respect the
    # one-sided emission rule (dot(d, n) < 0)
    from mitsuba2_amd import scenes
    S = oracle.OracleScene(scenes.cornell_box())
    area = oracle.lib().mo_scene_emitter_area(S.h, 0)
    assert abs(area - 130 * 105) < 1e-2
    rng = np.random.RandomState(5)
    for _ in range(50):
        ref = np.array([rng.uniform(50, 500), rng.uniform(10, 500), rng.uniform(50, 500)], np.float32)
        out = S.sample_emitter(ref, rng.rand(2).astype(np.float32))
        d, dist, pdf, n, p, spec, pdf2 = out[0:3], out[3], out[4], out[5:8], out[8:11], out[11:14], out[14]
        assert abs(np.linalg.norm(d) - 1) < 1e-5 and np.allclose(ref + d * dist, p, atol=1e-2)
        assert np.allclose(n, [0, -1, 0], atol=1e-6) and abs(p[1] - 548.3) < 1e-3
        assert 213 - 1e-3 <= p[0] <= 343 + 1e-3 and 227 - 1e-3 <= p[2] <= 332 + 1e-3
        expect_pdf = dist * dist / (abs(np.dot(d, n)) * area)
        assert abs(pdf - expect_pdf) / expect_pdf < 1e-4 and abs(pdf - pdf2) <= 2e-7 * pdf
        assert np.allclose(spec * pdf, [18.387, 13.9873, 6.75357], rtol=1e-5)
    # a reference point above the light sees its back side: zero radiance, pdf still reported
    out = S.sample_emitter(np.array([278, 600, 279], np.float32), np.array([0.3, 0.4], np.float32))
    assert (out[11:14] == 0).all() and out[4] > 0 and out[14] == 0


def test_integrator_parameter_checks(oracle):
    # src/python/python/test/test_integrator.py:79-106: max_depth = -2 and rr_depth = 0 are rejected
    from mitsuba2_amd import scenes
    S = oracle.OracleScene(scenes.cornell_box())
    p = scenes.cornell_box_sensor(8, 8, 1)
    for bad in (dict(max_depth=-2), dict(rr_depth=0)):
        q = dict(p); q.update(bad)
        with pytest.raises(RuntimeError):
            S.render(oracle.make_desc(q))


def test_empty_scene_renders_black(oracle):
    # src/python/python/test/scenes.py:262-267: a scene without emitters renders to all-zero RGBA
    from mitsuba2_amd import scenes
    S = oracle.OracleScene(scenes.stairs(4))
    p = scenes.cornell_box_sensor(16, 16, 2)
    p["to_world"] = scenes.look_at([0.5, 0.5, 3], [0.5, 0.5, 0], [0, 1, 0])
    p["near_clip"], p["far_clip"] = 0.01, 100.0
    for mode in (0, 1):
        film, _ = S.render(oracle.make_desc(p), mode=mode)
        assert (film[..., :3] == 0).all() and film[..., 3].max() > 0 and film[..., 4].min() > 0


def test_block_and_wavefront_modes_agree_statistically(oracle):
    # scalar_rgb block seeding vs per-sample wavefront seeding estimate the same image
    from mitsuba2_amd import scenes
    S = oracle.OracleScene(scenes.cornell_box())
    p = scenes.cornell_box_sensor(32, 32, 64)
    d = oracle.make_desc(p)
    f0, s0 = S.render(d, mode=0)
    f1, s1 = S.render(d, mode=1)
    a, b = oracle.film_develop(f0), oracle.film_develop(f1)
    assert s0[2] == s1[2] == 32 * 32 * 64
    assert abs(a[..., :3].mean() - b[..., :3].mean()) / b[..., :3].mean() < 0.03
    assert abs(a[..., 3].mean() - b[..., 3].mean()) < 0.01 and np.abs(a[..., 3] - b[..., 3]).max() < 0.2
    # wavefront mode is independent of the thread count and bitwise reproducible
    f2, _ = S.render(d, mode=1, n_threads=2)
    assert (f1 == f2).all()
    # block mode is reproducible too (blocks are merged in spiral order)
    f3, _ = S.render(d, mode=0, n_threads=3, block_size=32)
    f4, _ = S.render(d, mode=0, n_threads=1, block_size=32)
    assert (f3 == f4).all()


def test_bvh_matches_brute_force_in_oracle(oracle):
    from mitsuba2_amd import scenes
    for sd in (scenes.cornell_box(), scenes.bumpy_sphere(12, 24)):
        S = oracle.OracleScene(sd)
        rng = np.random.RandomState(7)
        allp = np.concatenate([m["positions"] for m in sd["meshes"]])
        lo, hi = allp.min(0), allp.max(0)
        n = 4000
        o = (lo + (hi - lo) * rng.rand(n, 3)).astype(np.float32)
        d = rng.randn(n, 3); d /= np.linalg.norm(d, axis=1, keepdims=True)
        mint, maxt = np.full(n, 1e-4, np.float32), np.full(n, np.inf, np.float32)
        a = S.ray_intersect(o, d, mint, maxt, naive=True)
        b = S.ray_intersect(o, d, mint, maxt, naive=False)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
        assert (S.ray_test(o, d, mint, maxt, naive=True) == S.ray_test(o, d, mint, maxt, naive=False)).all()


@pytest.mark.parametrize("origin", [[1.0, 0.0, 1.5], [1.0, 4.0, 1.5]])
@pytest.mark.parametrize("direction", [[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
@pytest.mark.parametrize("aperture_rad", [0.01, 0.1, 0.25])
@pytest.mark.parametrize("focus_dist", [1, 15, 25])
def test_thinlens_sample_ray(origin, direction, aperture_rad, focus_dist):
    """src/sensors/tests/test_thinlens.py:59-108 (test02_sample_ray), shutter closed"""
    from mitsuba2_amd import scenes
    to_world = scenes.look_at(origin, (np.array(origin) + np.array(direction)).tolist(), [0, 1, 0])
    p = dict(to_world=to_world, fov=34.0, near_clip=1.0, far_clip=35.0, width=512, height=256, crop=(0, 0, 512, 256), rfilter="gaussian",
             rfilter_param=0.5, sample_count=1, seed=0, max_depth=-1, rr_depth=5, aperture_radius=aperture_rad, focus_distance=focus_dist)
    d = ob.make_desc(p)
    o, dirs, _, _ = ob.camera_rays(d, [0.2, 0.6], [0.1, 0.9], [[0.5, 0.5], [0.5, 0.5]])
    assert np.allclose(o, origin, atol=1e-6)
    o, dirs, mint, maxt = ob.camera_rays(d, [0.5], [0.5], [[0.5, 0.5]])
    assert np.allclose(dirs[0], direction, atol=1e-6) and np.isclose(mint[0], 1.0) and np.isclose(maxt[0], 35.0)
    # aperture sampling
    ap = np.float32([[0.9, 0.6], [0.4, 0.9], [0.2, 0.7]])
    o, dirs, _, _ = ob.camera_rays(d, [0.5] * 3, [0.5] * 3, ap)
    oc, dc, _, _ = ob.camera_rays(d, [0.5], [0.5], [[0.5, 0.5]])
    tmp = _warp(0, ap)[:, :2]                                        # square_to_uniform_disk_concentric
    aperture_v = (to_world[:3, :3] @ (aperture_rad * np.concatenate([tmp, np.zeros((3, 1), np.float32)], 1)).T).T
    assert np.allclose(o, oc + aperture_v, atol=1e-6)
    want = dc * focus_dist - aperture_v                                    # (assumes near_clip = 1, as the reference's test does)
    assert np.allclose(dirs, want / np.linalg.norm(want, axis=1, keepdims=True), atol=1e-6)
    # a pinhole camera is the limit of a closed aperture
    pin = ob.camera_rays(ob.make_desc(dict(p, aperture_radius=None)), [0.3], [0.7])
    tiny = ob.camera_rays(ob.make_desc(dict(p, aperture_radius=0.0)), [0.3], [0.7], [[0.9, 0.1]])
    assert np.allclose(pin[0], tiny[0], atol=1e-6) and np.allclose(pin[1], tiny[1], atol=1e-6)
