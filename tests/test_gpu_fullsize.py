"""BASELINE.json's full-size configuration (cbox 1024 x 1024 @ 256 spp, 2.7e8 samples) checked through size-independent
properties, since the oracle needs minutes for it: the film must not depend on how the work is cut (film partition into
interleaved tiles, scheduler geometry), the weight channel is the same sum of filter taps for every interior pixel, the image
statistics agree with a small oracle render of the same scene, and the spectral / RGB variants agree in luminance."""
import numpy as np
import pytest
import torch

import oracle_binding as ob

pytestmark = pytest.mark.gpu


def test_full_size_cbox_properties():
    from mitsuba2_amd import render as R, scenes
    cb = scenes.cornell_box()
    sp = scenes.cornell_box_sensor(1024, 1024, spp=256, seed=0)
    scene, sensor = R.Scene(cb), R.make_sensor(sp)
    integ = R.PathIntegrator()
    assert integ.render(scene, sensor)
    full = sensor.film().bitmap(raw=True).clone()
    st = integ.stats
    assert st["samples"] == 1024 * 1024 * 256 and 3.5 < st["segments"] / st["samples"] < 5.0
    assert torch.isfinite(full).all() and (full[..., :3] >= 0).all()
    # weight channel: every interior pixel receives the same expected filter mass; alpha == weight where the box covers the film
    w = full[4:-4, 4:-4, 4]
    assert abs(float(w.std() / w.mean())) < 2e-2
    # film partition: two interleaved-tile parts add up to the whole (same samples, different accumulation order)
    parts = torch.zeros_like(full)
    for part in range(2):
        assert integ.render(scene, sensor, partition=(part, 2, 32))
        parts += sensor.film().bitmap(raw=True)
    assert torch.allclose(parts, full, rtol=2e-5, atol=1e-4)
    # scheduler geometry does not matter
    small = R.PathIntegrator(paths_per_wave=128)
    assert small.render(scene, sensor)
    assert torch.allclose(sensor.film().bitmap(raw=True), full, rtol=2e-5, atol=1e-4)
    # the headline configuration itself against the oracle (round 3): the oracle renders, with the same per-sample seeds, the window of
    # film rows 506..517 (all 1024 columns, 3.1e6 samples); the film pixels of rows 508..515, columns 2..1021 receive splats from
    # exactly that window (gaussian radius 2), so there both films hold the same sums of the same samples
    win = ob.OracleScene(cb).render_window(ob.make_desc(sp), 506, 518, 0, 1024)[508:516, 2:1022]
    got_w = full[508:516, 2:1022].cpu().numpy()
    assert win[..., 4].min() > 0
    assert np.allclose(got_w[..., 3:], win[..., 3:], rtol=2e-5, atol=1e-3)          # alpha and weight
    rgb_g, rgb_r = ob.film_develop(got_w)[..., :3], ob.film_develop(win)[..., :3]
    relmse = float(np.mean((rgb_g - rgb_r) ** 2 / (rgb_r ** 2 + 1e-2)))
    assert relmse < 1e-3                                                           # north_star tolerance
    assert relmse < 1e-9, relmse                                                   # bit-identical samples: only the order of the fp32 sums differs
    # against the oracle on a 64 x 64 @ 256 spp render of the same scene: mean radiance and per-channel balance
    lo = scenes.cornell_box_sensor(64, 64, spp=256, seed=0)
    ref, _ = ob.OracleScene(cb).render(ob.make_desc(lo), mode=1)
    ref_rgb = ob.film_develop(ref)[..., :3]
    got_rgb = sensor.film().bitmap().cpu().numpy()[..., :3]
    down = got_rgb.reshape(64, 16, 64, 16, 3).mean((1, 3))
    assert np.allclose(down.mean((0, 1)), ref_rgb.mean((0, 1)), rtol=2e-2)
    assert np.mean((down - ref_rgb) ** 2 / (ref_rgb ** 2 + 1e-2)) < 1e-2          # different pixel footprints + the oracle's noise


@pytest.mark.parametrize("variant", ["rgb", "spectral"])
def test_full_size_mesh_properties(variant):
    """BASELINE config 3 geometry (261 k triangles with vertex normals, 1920 x 1080) through the schedules: the default split
    pipeline (two launch chains, any-hit overlapped with the next closest-hit, dynamic ray fetch, quantised nodes, spill stack)
    must return the very film of the fused kernel, whatever the scheduler geometry; ray queries on the same scene agree
    between the walk kernel, the full-SurfaceInteraction kernel and brute force on a sample of rays."""
    from mitsuba2_amd import render as R, scenes
    sd = scenes.bumpy_sphere(256, 512)
    sp = scenes.bumpy_sphere_sensor(1920, 1080, 4)
    scene = R.Scene(sd, variant=variant)
    assert scene.info()["primitives"] > 250000
    films = []
    for kw in (dict(pipeline=0), dict(pipeline=1), dict(pipeline=2, paths_per_wave=100)):
        sensor = R.make_sensor(sp)
        integ = R.PathIntegrator(**kw)
        assert integ.render(scene, sensor)
        films.append(sensor.film().bitmap(raw=True))
        assert integ.stats["samples"] == 1920 * 1080 * 4
    assert torch.equal(films[0], films[1]) and torch.equal(films[0], films[2])
    f = films[0].cpu().numpy()
    assert np.isfinite(f).all() and f[..., 4].min() > 0 and f[..., :3].max() > 0
    if variant == "rgb":
        rng = np.random.RandomState(2)
        n = 1 << 18
        s = rng.rand(n, 2).astype(np.float32)
        ray = R.make_sensor(sp).sample_ray(torch.from_numpy(s).cuda())
        fast, full = scene.ray_intersect(ray, full=False), scene.ray_intersect(ray, full=True)
        assert torch.equal(fast.t, full.t) and torch.equal(fast.prim_index, full.prim_index)
        sub = R.Ray3f(o=ray.o[:4096].contiguous(), d=ray.d[:4096].contiguous(), mint=ray.mint[:4096].contiguous(), maxt=ray.maxt[:4096].contiguous())
        slow = scene.ray_intersect_naive(sub)
        assert torch.equal(fast.t[:4096], slow.t) and torch.equal(fast.prim_index[:4096], slow.prim_index)
        assert torch.equal(scene.ray_test(ray), torch.isfinite(fast.t))


def test_config3_full_size_against_oracle_rows():
    """BASELINE config 3 as stated: ~250 k triangles, spectral variant, 1920 x 1080 @ 1024 spp = 2.1e9 camera samples in 2 passes
    (546 + 534 film rows: a pass holds up to 2^30 samples).  The oracle cannot render that (hours); it renders, in wavefront mode with
    the very same per-sample seeds, the seven film rows 543..549 around the border between the passes (13.8 M samples).  Film rows
    545..547 receive splats from exactly those source rows (gaussian radius 2), so there the full-size GPU film and the oracle's film
    hold the same sum."""
    from mitsuba2_amd import render as R, scenes
    sd = scenes.bumpy_sphere(256, 512)
    sp = scenes.bumpy_sphere_sensor(1920, 1080, 1024)
    scene = R.Scene(sd, variant="spectral")
    assert scene.info()["primitives"] > 250000
    sensor = R.make_sensor(sp)
    integ = R.PathIntegrator()
    assert integ.render(scene, sensor)
    st = integ.stats
    assert st["samples"] == 1920 * 1080 * 1024 and st["passes"] == 2
    film = sensor.film().bitmap(raw=True)
    assert torch.isfinite(film).all()
    got = film[545:548].cpu().numpy()
    S = ob.OracleScene(sd, spectral_path=R.srgb_coeff_path())
    ref = S.render_rows(ob.make_desc(sp), 543, 550)[545:548]
    assert ref[..., 3].mean() > 0.1 * ref[..., 4].mean()                 # the rows cross the object
    assert np.allclose(got[..., 3:], ref[..., 3:], rtol=1e-4, atol=1e-3)    # alpha and weight: sums of 1024 x ~12 filter taps
    rgb_g, rgb_r = ob.film_develop(got)[..., :3], ob.film_develop(ref)[..., :3]
    relmse = float(np.mean((rgb_g - rgb_r) ** 2 / (rgb_r ** 2 + 1e-2)))
    assert relmse < 1e-3                                                 # north_star tolerance
    assert relmse < 1e-8, relmse                                         # bit-identical samples (round 3): only the order of the fp32 sums differs


def test_config5_rank_partition_at_full_spp_against_oracle_rows():
    """BASELINE config 5 as stated -- cbox 4096 x 4096 @ 4096 spp, film tile-partitioned over 8 GPUs -- at the per-pixel sample count of
    the configuration: rank 3 of 8 renders its partition (interleaved 16-row tiles, mitsuba2_amd/dist.py: 512 rows = 8.6e9 camera
    samples, 8 passes of 2^30) exactly as `bench.py --config cbox4k --gpus 8` would on that rank.  The oracle renders, with the same
    per-sample seeds, the window of film rows 564..569 x columns 1536..2303 inside the first tile of the rank's SECOND pass (1.9e7 samples);
    the film pixels of rows 566 and 567, columns 1538..2301 receive splats from exactly that window (gaussian radius 2) and all of its
    rows belong to this rank, so there the partition's film and the oracle's hold the same sum.  Rounds 1-2 checked this shape at 2 spp against itself."""
    from mitsuba2_amd import render as R, scenes, dist as mdist
    cb = scenes.cornell_box()
    sp = scenes.cornell_box_sensor(4096, 4096, spp=4096, seed=0)
    scene, sensor = R.Scene(cb), R.make_sensor(sp)
    integ = R.PathIntegrator()
    part = mdist.film_partition(3, 8)
    assert integ.render(scene, sensor, partition=part)
    st = integ.stats
    own = mdist.owned_rows(4096, 3, 8)
    assert st["samples"] == len(own) * 4096 * 4096 == 512 * 4096 * 4096 and st["passes"] == 8
    # local rows 64..79 (the first tile of pass 2) are the global rows of tile 3 + 8 * 4 = 35
    assert own[64] == 560 and own[79] == 575
    film = sensor.film().bitmap(raw=True)
    assert torch.isfinite(film).all()
    got = film[566:568, 1538:2302].cpu().numpy()
    # nothing of another rank's tiles: rows beyond the 2-pixel apron of this rank's tiles stay empty
    assert float(film[544 + 2: 560 - 2].abs().max()) == 0.0
    ref = ob.OracleScene(cb).render_window(ob.make_desc(sp), 564, 570, 1536, 2304)[566:568, 1538:2302]
    assert ref[..., 4].min() > 0
    assert np.allclose(got[..., 3:], ref[..., 3:], rtol=2e-4, atol=1e-2)   # alpha and weight: sums of 4096 x ~12 filter taps
    rgb_g, rgb_r = ob.film_develop(got)[..., :3], ob.film_develop(ref)[..., :3]
    relmse = float(np.mean((rgb_g - rgb_r) ** 2 / (rgb_r ** 2 + 1e-2)))
    assert relmse < 1e-3                                                 # north_star tolerance
    assert relmse < 1e-8, relmse                                         # bit-identical samples, fp32 sums of 4096 x 12 terms in two orders
