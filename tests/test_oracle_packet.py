"""The packet_rgb-equivalent CPU baseline (oracle/mo_packet.c + the packet flavour of mo_render.c) against the scalar oracle.

The reference's packet mode (src/librender/integrator.cpp:204-212) walks a block's pixel_count * spp sample indices 8 at a time with
8 sampler streams; its kd-tree traversal keeps one stack for the packet, a lane mask per entry and votes on the visiting order
(include/mitsuba/render/kdtree.h:2176-2300).  Our restatement of that schedule (mode 2: ray queries 8 wide on AVX2) must produce the
very film of the same schedule traced ray by ray with the scalar code (mode 3) -- bit for bit, since the triangle test keeps the
scalar operation order and the box tests only cull."""
import time

import numpy as np
import pytest

from mitsuba2_amd import scenes


@pytest.mark.parametrize("name", ["cbox", "sphere", "sphere_textured_env"])
def test_packet_film_is_the_scalar_film_of_the_same_schedule(oracle, name):
    if name == "cbox":
        sd, p = scenes.cornell_box(), scenes.cornell_box_sensor(80, 56, 6, seed=5)
    elif name == "sphere":
        sd, p = scenes.bumpy_sphere(40, 80), scenes.bumpy_sphere_sensor(72, 48, 4, seed=2)
    else:
        sd, p = scenes.bumpy_sphere(16, 32, with_normals=False), scenes.bumpy_sphere_sensor(48, 40, 5, seed=7)
        sd["emitters"].append(dict(type="constant", radiance=np.array([0.3, 0.4, 0.5], np.float32)))
    S = oracle.OracleScene(sd)
    d = oracle.make_desc(p)
    packet, st_p = S.render(d, mode=oracle.PACKET_MODE, n_threads=4)
    check, st_c = S.render(d, mode=oracle.PACKET_CHECK_MODE, n_threads=4)
    assert (packet == check).all(), float(np.abs(packet - check).max())
    assert (st_p == st_c).all() and st_p[2] == p["width"] * p["height"] * p["sample_count"]
    # thread count and block size do not matter (blocks are merged in spiral order)
    again, _ = S.render(d, mode=oracle.PACKET_MODE, n_threads=1, block_size=16)
    check16, _ = S.render(d, mode=oracle.PACKET_CHECK_MODE, n_threads=2, block_size=16)
    assert (again == check16).all()
    # a different estimate of the same image as the scalar_rgb schedule: same weights, statistically the same radiance
    scalar, st_s = S.render(d, mode=oracle.SCALAR_MODE, n_threads=4)
    assert abs(packet[..., 4].mean() - scalar[..., 4].mean()) < 0.02 * scalar[..., 4].mean()
    assert abs(packet[..., 1].mean() - scalar[..., 1].mean()) < 0.1 * scalar[..., 1].mean()
    assert st_s[2] == st_p[2] and abs(int(st_s[0]) - int(st_p[0])) < 0.05 * st_s[0]


def test_spectral_packet_film_is_the_scalar_film_of_the_same_schedule(oracle):
    """the same for the spectral variant (BASELINE config 3's CPU baseline): every lane carries its own four wavelengths, the ray queries
    are 8 wide, the shading is the scalar spectral code -- bit for bit the film of the schedule traced lane by lane"""
    from mitsuba2_amd import render
    path = render.srgb_coeff_path()
    for sd, p in ((scenes.cornell_box(), scenes.cornell_box_sensor(64, 48, 4, seed=3)), (scenes.bumpy_sphere(32, 64), scenes.bumpy_sphere_sensor(64, 40, 3, seed=4))):
        S = oracle.OracleScene(sd, spectral_path=path)
        d = oracle.make_desc(p)
        packet, st_p = S.render(d, mode=oracle.PACKET_MODE, n_threads=4)
        check, st_c = S.render(d, mode=oracle.PACKET_CHECK_MODE, n_threads=4)
        assert (packet == check).all(), float(np.abs(packet - check).max())
        assert (st_p == st_c).all() and packet[..., :3].max() > 0


def test_packet_queries_match_scalar_queries_on_incoherent_rays(oracle):
    # the 8-wide query itself, on random rays through a 7 k-triangle mesh (ragged last packet included): t, primitive, u, v and the
    # any-hit flag equal the scalar BVH walk's
    sd = scenes.bumpy_sphere(40, 80)
    S = oracle.OracleScene(sd)
    rng = np.random.RandomState(3)
    n = 20003
    o = (rng.rand(n, 3) * 6 - 3).astype(np.float32); o[:, 1] += 1.0
    d = rng.randn(n, 3); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[::97, 0] = 0.0                                   # axis-parallel components
    mint = np.full(n, 1e-4, np.float32)
    maxt = np.where(rng.rand(n) < 0.3, rng.rand(n) * 5, np.inf).astype(np.float32)
    t, prim, shape, u, v = S.ray_intersect(o, d, mint, maxt, naive=False)
    tp, pp, up, vp, hp = S.packet_intersect(o, d, mint, maxt)
    assert (tp == t).all() and (pp == prim).all() and (up == u).all() and (vp == v).all()
    assert (hp == S.ray_test(o, d, mint, maxt, naive=False)).all()
    assert np.isfinite(t).sum() > n // 10


def test_packet_mode_limits(oracle):
    S = oracle.OracleScene(scenes.cornell_box())
    p = scenes.cornell_box_sensor(16, 16, 2)
    d = oracle.make_desc(p)
    d.integrator = 1
    with pytest.raises(RuntimeError):
        S.render(d, mode=oracle.PACKET_MODE)           # `direct` is not part of the packet baseline
