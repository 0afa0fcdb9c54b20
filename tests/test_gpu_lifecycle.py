"""Life cycle of a render call on the GPU: several passes (the path every film above 2^28 samples takes), cancel() from another
thread, the `timeout` and `samples_per_pass` properties of SamplingIntegrator (src/librender/integrator.cpp:27-66,122,175;
include/mitsuba/render/integrator.h:44-51,143-146) and the per-rank shape of BASELINE config 5 (4096 x 4096 film cut into
interleaved 32-row tiles over 8 ranks)."""
import os
import threading
import time

import numpy as np
import pytest
import torch

from mitsuba2_amd import scenes

pytestmark = pytest.mark.gpu


class _knobs:
    """scoped scheduler knobs of mtsamd_render_desc (max_pass_log2, finish_kernel) on an integrator and the one it wraps"""

    def __init__(self, integ, **kv):
        self.targets = [integ] + ([integ.nested] if hasattr(integ, "nested") else [])
        self.kv = kv

    def __enter__(self):
        self.old = [{k: getattr(t, k) for k in self.kv} for t in self.targets]
        for t in self.targets:
            for k, v in self.kv.items():
                setattr(t, k, v)

    def __exit__(self, *a):
        for t, old in zip(self.targets, self.old):
            for k, v in old.items():
                setattr(t, k, v)


def _film(gpu, integ, scene, p, **kw):
    sensor = gpu.make_sensor(p)
    assert integ.render(scene, sensor, **kw)
    return sensor.film().bitmap(raw=True).clone(), dict(integ.stats)


@pytest.mark.parametrize("what", ["path", "path_mesh_spectral", "direct", "moment", "partition"])
def test_multi_pass_film_equals_one_pass_film(gpu, what):
    """A render above the pass capacity is cut into passes of whole film rows (api.cpp, mtsamd_render); each pass re-seeds the
    per-wave cursors and splats filter aprons that overlap its neighbours'.  mtsamd_render_desc::max_pass_log2 forces the cut on a small film:
    same samples, same per-pixel accumulation order inside a pass, only the order of the row-border additions differs."""
    variant = "rgb"
    if what == "path_mesh_spectral":
        sd, p, variant = scenes.bumpy_sphere(48, 96), scenes.bumpy_sphere_sensor(96, 64, 8), "spectral"
    else:
        sd, p = scenes.cornell_box(), scenes.cornell_box_sensor(96, 64, 16, seed=4)
    scene = gpu.Scene(sd, variant=variant)
    integ = {"direct": gpu.DirectIntegrator(), "moment": gpu.MomentIntegrator(gpu.PathIntegrator())}.get(what, gpu.PathIntegrator())
    kw = dict(partition=(1, 3, 8)) if what == "partition" else {}
    one, st1 = _film(gpu, integ, scene, p, **kw)
    assert st1["passes"] == 1
    with _knobs(integ, max_pass_log2=13):        # 8192 samples per pass: 5 film rows of 96 px x 16 spp
        many, stn = _film(gpu, integ, scene, p, **kw)
    assert stn["passes"] >= 4 and stn["samples"] == st1["samples"]
    assert stn["closest_hit_rays"] == st1["closest_hit_rays"] and stn["any_hit_rays"] == st1["any_hit_rays"]
    assert torch.allclose(many, one, rtol=2e-5, atol=2e-5)
    assert float((many - one).abs().max()) < 1e-3 * float(one.abs().max())


def test_samples_per_pass_semantics(gpu):
    # integrator.cpp:59-66: sample_count must be a multiple of samples_per_pass; values above sample_count mean one pass
    sd, p = scenes.cornell_box(), scenes.cornell_box_sensor(32, 32, 12)
    scene = gpu.Scene(sd)
    ref, _ = _film(gpu, gpu.PathIntegrator(), scene, p)
    with pytest.raises(RuntimeError, match=r"sample_count \(12\) must be a multiple of samples_per_pass \(5\)"):
        gpu.PathIntegrator(samples_per_pass=5).render(scene, gpu.make_sensor(p))
    for spp_pass, passes in ((4, 4), (12, 1), (64, 1)):
        # a pass holds at most crop_width * crop_height * samples_per_pass samples (whole film rows at the full sample count):
        # 32 * 32 * 4 samples = 10 rows of 32 px x 12 spp -> 4 passes
        got, st = _film(gpu, gpu.PathIntegrator(samples_per_pass=spp_pass), scene, p)
        assert st["passes"] == passes
        # the RNG streams are seeded per global sample index: the samples cannot change, only the order in which the splats of
        # neighbouring passes meet at a row border
        assert torch.equal(got, ref) if passes == 1 else torch.allclose(got, ref, rtol=2e-5, atol=2e-5)


def test_cancel_from_another_thread(gpu):
    """Integrator::cancel may be called asynchronously (integrator.h:44-51); render() then returns false (integrator.cpp:175)
    and the handle stays usable: the next render is bit-identical to one on a fresh scene."""
    sd = scenes.cornell_box()
    scene = gpu.Scene(sd)
    big = scenes.cornell_box_sensor(1024, 1024, 1024)        # ~0.4 s of GPU work in 4 passes
    small = scenes.cornell_box_sensor(64, 64, 8, seed=3)
    integ = gpu.PathIntegrator()
    fresh, _ = _film(gpu, gpu.PathIntegrator(), gpu.Scene(sd), small)
    result = {}

    def worker():
        sensor = gpu.make_sensor(big)
        t0 = time.perf_counter()
        result["ok"] = integ.render(scene, sensor)
        result["dt"] = time.perf_counter() - t0

    _film(gpu, integ, scene, scenes.cornell_box_sensor(1024, 1024, 256))      # same scheduler geometry: the workspace exists
    th = threading.Thread(target=worker)
    th.start()
    deadline = time.perf_counter() + 20.0
    while integ._scene is None and time.perf_counter() < deadline:      # wait until render() is inside the library
        time.sleep(0.001)
    time.sleep(0.05)
    while th.is_alive():                         # render() clears the flag when it starts (m_stop = false, integrator.cpp:54)
        integ.cancel()
        time.sleep(0.002)
    th.join(60.0)
    assert not th.is_alive()
    assert result["ok"] is False, result
    again, st = _film(gpu, integ, scene, small)
    assert torch.equal(again, fresh)
    assert st["samples"] == 64 * 64 * 8
    # cancel() without a render in flight is a no-op
    integ.cancel()
    again2, _ = _film(gpu, integ, scene, small)
    assert torch.equal(again2, fresh)


@pytest.mark.parametrize("what", ["cbox", "cbox_spectral", "mesh", "mesh_spectral", "mesh_glossy"])
def test_finish_kernel_leaves_the_film_unchanged(gpu, what):
    """End of a pass (kernels.hip, k_finish): once every sample has been generated the remaining paths are run to their end by one
    fused launch instead of further launch rounds.  A path performs the same floating-point operations in the same order either way:
    with the switch taken as early as possible (threshold = the whole pool) and never (0), film and ray counts must be identical."""
    variant = "spectral" if what.endswith("spectral") else "rgb"
    if what.startswith("cbox"):
        sd, p = scenes.cornell_box(), scenes.cornell_box_sensor(160, 128, 32, seed=3)
    else:
        sd, p = scenes.bumpy_sphere(64, 128), scenes.bumpy_sphere_sensor(192, 128, 16)
        if what == "mesh_glossy":
            sd["bsdfs"][0] = {"type": "roughplastic", "alpha": 0.2, "diffuse_reflectance": [0.3, 0.4, 0.5]}
    scene = gpu.Scene(sd, variant=variant)
    integ = gpu.PathIntegrator(max_depth=12)
    with _knobs(integ, finish_kernel=1):
        never, st0 = _film(gpu, integ, scene, p)
    with _knobs(integ, finish_kernel=2):
        early, st1 = _film(gpu, integ, scene, p)
    assert st1["iterations"] < st0["iterations"]                 # the fused launch really replaced launch rounds
    for k in ("samples", "segments", "closest_hit_rays", "any_hit_rays"):
        assert st0[k] == st1[k], (k, st0[k], st1[k])
    assert torch.equal(never, early)


def test_timeout_stops_between_passes(gpu):
    """`timeout` (integrator.cpp:38, should_stop integrator.h:143-146): work stops being scheduled once the timer runs out;
    render() still returns true (only cancel() sets m_stop) and the film holds what was put before."""
    sd = scenes.cornell_box()
    scene = gpu.Scene(sd)
    p = scenes.cornell_box_sensor(512, 512, 256)
    integ = gpu.PathIntegrator()
    integ.max_pass_log2 = 22                     # 16 passes of 32 rows
    full, st_full = _film(gpu, integ, scene, p)
    t0 = time.perf_counter()
    full, st_full = _film(gpu, integ, scene, p)
    dt = time.perf_counter() - t0
    assert st_full["passes"] == 16 and st_full["timed_out"] == 0
    integ = gpu.PathIntegrator(timeout=dt / 3.0)
    integ.max_pass_log2 = 22
    part, st = _film(gpu, integ, scene, p)
    assert st["timed_out"] == 1 and st["passes"] < 16, (st, dt)
    done_rows = 32 * st["passes"]
    w = part[..., 4].cpu().numpy()
    assert (w[done_rows + 2:] == 0).all()                    # nothing was put for the rows of the abandoned passes
    if done_rows > 4:
        assert torch.equal(part[:done_rows - 2], full[:done_rows - 2])
    # a generous timeout changes nothing
    ok, st_ok = _film(gpu, gpu.PathIntegrator(timeout=600.0), scene, p)
    assert st_ok["timed_out"] == 0


def test_config5_per_rank_shape_adds_up(gpu):
    """BASELINE config 5 on one GPU: 4096 x 4096 film, every one of the 8 ranks' partitions (interleaved 32-row tiles), at
    least two passes per rank, reduced sample count.  The eight partial films add up to the unpartitioned film."""
    sd = scenes.cornell_box()
    scene = gpu.Scene(sd)
    p = scenes.cornell_box_sensor(4096, 4096, 2, seed=1)
    integ = gpu.PathIntegrator()
    with _knobs(integ, max_pass_log2=23):        # one film row = 2^13 samples: 4 passes of 1024 rows for the whole film
        whole, st = _film(gpu, integ, scene, p)
        assert st["passes"] == 4 and st["samples"] == 4096 * 4096 * 2
    total = torch.zeros_like(whole)
    with _knobs(integ, max_pass_log2=21):        # a rank owns 512 rows = 2^22 samples: 2 passes of 256 rows (8 tiles)
        for r in range(8):
            part, st_r = _film(gpu, integ, scene, p, partition=(r, 8, 32))
            assert st_r["samples"] == 4096 * 4096 * 2 // 8 and st_r["passes"] == 2
            rows = torch.zeros(4096, dtype=torch.bool)
            for t in range(r, 128, 8):
                rows[32 * t: 32 * t + 32] = True
            # a rank's splats stay within its tiles plus the 2-pixel filter apron
            far = torch.ones(4096, dtype=torch.bool)
            for t in range(r, 128, 8):
                far[max(0, 32 * t - 2): 32 * t + 34] = False
            assert float(part[far.cuda()].abs().max()) == 0.0
            total += part
    assert torch.allclose(total, whole, rtol=2e-5, atol=1e-4)


def _rccl_worker(rank, world, port, out_path):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from mitsuba2_amd import dist as mdist, render, scenes
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    scene = render.Scene(scenes.cornell_box(), device=0)
    sensor = render.make_sensor(scenes.cornell_box_sensor(96, 64, 8, seed=2))
    film = mdist.render_distributed(render.PathIntegrator(), scene, sensor, all_ranks=True)      # all_reduce on the device tensor
    t = torch.ones(4, device="cuda") * (rank + 1)
    dist.all_reduce(t)
    torch.save({"film": film.cpu(), "sum": t.cpu()}, out_path)
    dist.destroy_process_group()


def test_rccl_backend_runs_the_film_reduce(gpu, tmp_path):
    """The `nccl` backend of torch.distributed IS RCCL on ROCm.  The GPU box has one card, so this is a one-rank group -- but it is the
    code path of the N-GPU bench: process-group initialisation on the device, the film reduce on a device tensor without host
    staging, and a plain all_reduce.  (Two ranks on one card are refused by RCCL; the two-rank logic runs over gloo.)"""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_worker, args=(1, port, out), nprocs=1, join=True)
    got = torch.load(out, weights_only=True)
    scene = gpu.Scene(scenes.cornell_box())
    ref, _ = _film(gpu, gpu.PathIntegrator(), scene, scenes.cornell_box_sensor(96, 64, 8, seed=2))
    assert torch.equal(got["film"], ref.cpu()) and torch.equal(got["sum"], torch.ones(4))
