#!/usr/bin/env python3
"""Benchmarks of the path-tracing hot path on MI355X, one JSON line on stdout (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cbox|mesh|autodiff|cbox4k]

`--gpus N > 1` without a torch.distributed environment starts the N ranks itself (a `torch.distributed.run` child,
before anything touches the GPU); under `python -m torch.distributed.run ... bench.py --gpus N` the ranks are used as given.

Configurations (BASELINE.json `configs`):
  cbox      [1] synthetic Cornell box, RGB, 1024x1024 @ 256 spp per GPU -- the headline line.  N > 1: the film is cut into
            interleaved 16-row tiles (mitsuba2_amd/dist.py), every rank keeps 2^28 camera samples (weak scaling: 256 N spp),
            the per-rank XYZAW films are summed with one RCCL reduce.
  mesh      [2] 261 k-triangle displaced sphere, spectral variant, 1920x1080 @ 1024 spp, 1 GPU, all-diffuse materials;
            mesh_matpreview: the same with a roughplastic object, a checkerboard ground and an envmap sky (general BSDF / emitter kernels).
  autodiff  [3] one inverse-rendering iteration (primal + derivative render + adjoint + Adam) on the Cornell box, the setup of
            docs/examples/10_inverse_rendering/invert_cbox.py; metric = ms per iteration.
  cbox4k    [4] 4096x4096 @ 4096 spp Cornell box, film tile-partitioned over the N ranks (strong scaling: the work is fixed).
With the default `--config cbox` the line also carries the other configurations under "other_configs" (mesh and autodiff at
N = 1, cbox4k at every N), each with its own small fixed step count, so that every BASELINE config has a driver-visible number.

A step is one complete render (every sample traced to termination, film splatted, N > 1: films reduced).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
STATE_BYTES = 88               # per in-flight path: ray 32 B + path state 56 B (kernels.h PoolView)


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return min(n, 64)


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    return ob


# ------------------------------------------------------------------------------------------------------------------
# CPU baselines: the oracle (checker code) timed as a *reported baseline* on a bounded sample of the same workload
def cpu_baseline(sd, sensor_fn, label, spectral_path=None, seconds=10.0, max_spp=64, packet=True):
    """The CPU port (oracle, scalar_rgb block mode: spiral 32x32 blocks, one PCG32 stream per block, BVH, all host cores) and
    its 8-wide packet variant (packet_rgb / packet_spectral-equivalent restatement: Morton-ordered 8-ray packets, lane voting at inner
    nodes, kdtree.h:2176-2300) on the same film at a reduced sample count."""
    ob = _oracle()
    S = ob.OracleScene(sd, spectral_path=spectral_path)
    cores = host_cores()

    def timed(mode, spp):
        p = sensor_fn(spp)
        t0 = time.perf_counter()
        _, stats = S.render(ob.make_desc(p), mode=mode, n_threads=cores)
        return time.perf_counter() - t0, stats, p

    dt1, _, _ = timed(0, 1)                        # calibrate on 1 spp
    spp = int(max(1, min(max_spp, round(seconds / max(dt1, 1e-3)))))
    dt, stats, p = timed(0, spp)
    samples = p["crop"][2] * p["crop"][3] * spp
    out = {"value": samples / dt / 1e6, "unit": "Msample/s", "cores": cores, "kind": "port",
           "sample": "%s %dx%d@%dspp, oracle scalar block mode (spiral 32x32 blocks, BVH), %.1f s" % (label, p["crop"][2], p["crop"][3], spp, dt),
           "mray_per_s": float(stats[0] + stats[1]) / dt / 1e6, "scalar": samples / dt / 1e6}
    if packet and hasattr(ob, "PACKET_MODE"):
        dtp, stp, _ = timed(ob.PACKET_MODE, spp)
        out["packet"] = samples / dtp / 1e6
        out["packet_mray_per_s"] = float(stp[0] + stp[1]) / dtp / 1e6
        out["sample"] += "; packet restatement (8-wide AVX2 packets, lane voting, same blocks and spp) %.1f s" % dtp
    return out


# ------------------------------------------------------------------------------------------------------------------
def same_run_parity(render, scenes, scene, sd, device):
    """Parity of the very kernels that were just timed, on a reduced render (same scene, 128x128 @ 256 spp): against the oracle in
    wavefront mode (same per-sample seeds: tight) and against the oracle in scalar block mode (independent streams: relMSE next to
    its noise floor 2 var / N, and the reference's per-pixel z-test, src/librender/tests/test_renders.py:60-134)."""
    import numpy as np
    ob = _oracle()
    import render_stats
    p = scenes.cornell_box_sensor(128, 128, 256, seed=0)
    sensor = render.make_sensor(p)
    assert render.PathIntegrator().render(scene, sensor)
    raw = sensor.film().bitmap(raw=True).cpu().numpy()
    S = ob.OracleScene(sd)
    cores = host_cores()
    wf = S.render(ob.make_desc(p), mode=1, n_threads=cores)[0]
    blk = S.render(ob.make_desc(p), mode=0, n_threads=cores)[0]
    rel = lambda a, b: float(np.mean((a - b) ** 2 / (b ** 2 + 1e-2)))
    got, ref_wf, ref_blk = ob.film_develop(raw)[..., :3], ob.film_develop(wf)[..., :3], ob.film_develop(blk)[..., :3]
    msensor = render.make_sensor(p)
    assert render.MomentIntegrator(render.PathIntegrator()).render(scene, msensor)
    mean, var = render.MomentIntegrator.mean_and_variance(msensor.film())
    mean, var = mean.cpu().numpy(), var.cpu().numpy()
    xyz_blk = blk[..., :3] / blk[..., 4:5]
    pv = render_stats.z_test(mean, 256, xyz_blk, var)
    alpha = 1.0 - (1.0 - 0.01) ** (1.0 / (128 * 128))
    # the block-mode image is itself a 256-spp estimate: the difference of two independent estimates has twice the variance
    pv2 = render_stats.z_test(mean, 128, xyz_blk, var)
    return {"render": "cbox 128x128@256spp, same kernels as the timed region",
            "relmse_vs_oracle_same_seeds": rel(got, ref_wf),
            "relmse_vs_oracle_scalar_block_mode": rel(got, ref_blk),
            # a gaussian-filtered pixel averages N_eff = spp * (int w)^2 / int w^2 = spp * 4 pi stddev^2 = pi * spp samples (stddev 0.5)
            "relmse_noise_floor_two_independent_estimates": float(np.mean(2.0 * var / (256.0 * np.pi) / (xyz_blk ** 2 + 1e-2))),
            "z_test_pixels_passing": float((pv2 > alpha).mean()), "z_test_required": 0.9975,
            "z_test_pixels_passing_if_reference_were_exact": float((pv > alpha).mean())}


# ------------------------------------------------------------------------------------------------------------------
class Ranks:
    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = args.backend
        if self.world != args.gpus:
            sys.stderr.write("bench.py: WORLD_SIZE=%d but --gpus %d\n" % (self.world, args.gpus))
            sys.exit(2)
        n_dev = torch.cuda.device_count()
        self.selftest = args.config == "launcher-selftest"      # exercises launcher + collectives only (CPU test of the N-rank path)
        if not self.selftest and (n_dev < 1 or (self.backend == "nccl" and n_dev < self.world)):
            sys.stderr.write("bench.py: %d ranks need %d GPUs, %d visible (use --backend gloo to rehearse ranks on fewer cards)\n"
                             % (self.world, self.world, n_dev))
            sys.exit(2)
        self.device = local % max(n_dev, 1)         # gloo rehearsal: several ranks may share a card
        if not self.selftest:
            torch.cuda.set_device(self.device)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.device))
            else:
                dist.init_process_group(self.backend)
        self.cdev = "cuda" if self.backend == "nccl" else "cpu"

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        if not self.selftest:
            self.torch.cuda.synchronize()

    def max(self, x):
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, xs):
        if self.world == 1:
            return [float(x) for x in xs]
        t = self.torch.tensor(list(xs), dtype=self.torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    def gather(self, x):
        if self.world == 1:
            return [float(x)]
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.cdev)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


STAT_KEYS = ("closest_hit_rays", "any_hit_rays", "samples", "iterations", "segments", "bounce_ns", "film_ns", "tri_tests",
             "trace_closest_ns", "trace_closest_launches", "trace_any_ns", "trace_any_launches", "shade_ns", "shade_launches", "passes")


def timed_renders(R, integ, scene, sensor, partition, steps, warmup):
    """W untimed + K timed steps bracketed by barrier + synchronize; returns (max-over-ranks seconds, summed stats, per-rank
    render ms, per-rank reduce ms)."""
    from mitsuba2_amd import dist as mdist
    acc = {k: 0 for k in STAT_KEYS}
    render_s = reduce_s = 0.0

    def step(timed):
        nonlocal render_s, reduce_s
        t0 = time.perf_counter()
        ok = integ.render(scene, sensor, partition=partition)
        assert ok
        R.torch.cuda.synchronize()
        t1 = time.perf_counter()
        mdist.reduce_film(sensor.film().bitmap(raw=True))      # RCCL over xGMI: per-rank ImageBlocks -> rank 0
        R.torch.cuda.synchronize()
        t2 = time.perf_counter()
        if timed:
            render_s += t1 - t0; reduce_s += t2 - t1
            for k in acc:
                acc[k] += integ.stats[k]

    for _ in range(warmup):
        step(False)
    R.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    R.barrier()
    dt = R.max(time.perf_counter() - t0)
    return dt, acc, render_s, reduce_s


# ------------------------------------------------------------------------------------------------------------------
def run_cbox(args, R, strong=False):
    """configs[1] (weak scaling, 1024^2 @ 256 spp per GPU) or, strong=True, configs[4] (4096^2 @ 4096 spp over the N ranks)."""
    from mitsuba2_amd import render, scenes, dist as mdist
    n = R.world
    if strong:
        width = height = args.width4k
        spp_total = args.spp4k
        steps, warmup = (args.steps, args.warmup) if args.config == "cbox4k" else (1, 0)
    else:
        width, height = args.width, args.height
        spp_total = args.spp * n                # weak scaling: 2^28 camera samples per GPU and step
        steps, warmup = args.steps, args.warmup
    sd = scenes.cornell_box()
    scene = render.Scene(sd, device=R.device)
    integ = render.PathIntegrator(paths_per_wave=args.paths_per_wave, pipeline=args.pipeline)
    partition = mdist.film_partition(R.rank, n)
    if strong and args.config != "cbox4k":      # allocate the workspace of the full-size passes outside the timed step
        integ.render(scene, render.make_sensor(scenes.cornell_box_sensor(width, height, max(1, spp_total // 16))), partition=partition)
    sensor = render.make_sensor(scenes.cornell_box_sensor(width, height, spp_total))
    dt, acc, render_s, reduce_s = timed_renders(R, integ, scene, sensor, partition, steps, warmup)
    tot_closest, tot_any, tot_samples = R.sum([acc["closest_hit_rays"], acc["any_hit_rays"], acc["samples"]])
    rank_ms = R.gather(render_s / steps * 1e3)
    red_ms = R.gather(reduce_s / steps * 1e3)
    if R.rank != 0:
        return None
    # roofline of the dominant kernel on rank 0 -- the one that advances every in-flight path by one segment: k_shade<PathState,
    # false, true, true> (closest hit + shading + in-kernel shadow ring, the default schedule for LDS-resident scenes) or k_bounce
    # with --pipeline 1: algorithmic bytes per launch / average launch time.  Per segment the kernel reads one 88-B path record and
    # writes one (a survivor or a regenerated camera path: survivors + generated == segments over a whole render); per finished
    # sample it writes its film record.  Launch durations come from HIP events recorded on the render stream inside mtsamd_render.
    launches = max(acc["iterations"], 1)
    alg_bytes = 2 * STATE_BYTES * acc["segments"] + SAMPLE_RECORD_BYTES * acc["samples"]
    bounce_s = acc["bounce_ns"] * 1e-9
    achieved = alg_bytes / max(bounce_s, 1e-12) / 1e9
    kernel = {0: "k_shade<mtsamd::PathState, false, true, true>", 4: "k_shade<mtsamd::PathState, false, true, true>",
              1: "k_bounce<true, false>", 2: "k_trace<false, false> + k_shade<mtsamd::PathState, false, false, false> + k_trace<true, false>",
              3: "k_shade<mtsamd::PathState, false, true, false> + k_trace<true, true>"}[args.pipeline]
    concurrent = min(4, max(1, int(os.environ.get("MTSAMD_STREAMS", "2")))) if args.pipeline in (0, 4) else 1
    # HBM bytes per launch: PMC counters cannot be read inside this process; the ratio counted / algorithmic bytes of the same kernel
    # (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3 passes, scripts/collect_profiles.sh) is committed under profiles/
    # and scaled to this run's bytes per launch.  `traffic_source` says so; it is an OFFLINE figure.
    traffic, traffic_src = None, None
    pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bounce_kernel.json")))
    valu = {}
    if pmc and not strong:
        try:
            prof = json.load(open(pmc[-1]))
            traffic = prof["traffic_over_algorithmic"] * (alg_bytes / launches)
            traffic_src = "offline profile " + os.path.relpath(pmc[-1], ROOT) + " (counted / algorithmic bytes of this kernel), scaled to this run"
            # what really bounds the kernel of an LDS-resident scene: VALU issue; same offline profile
            valu = {"valu_per_wave_segment": prof.get("valu_instructions_per_wave_segment"), "lane_utilisation": prof.get("valu_lane_utilisation"),
                    "bound_in_practice": "VALU issue: the 36-triangle scene lives in LDS, HBM only carries the path-state streams"}
        except (OSError, ValueError, KeyError):
            pass
    out = {
        "metric": ("Msample/s, cbox %dx%d@%dspp per GPU, path integrator (max_depth=-1, rr_depth=5)" % (width, height, args.spp)) if not strong else
                  ("Msample/s, cbox %dx%d@%dspp, film tile-partitioned over %d GPU(s)" % (width, height, spp_total, n)),
        "value": tot_samples / dt / 1e6,
        "unit": "Msample/s",
        "n_gpus": n, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "synthetic Cornell box (36 triangles, diffuse, 1 area light), %dx%d film, %d spp total%s, "
                               "gaussian rfilter, independent sampler" % (width, height, spp_total, "" if strong else " (%d per GPU)" % args.spp),
                   "partition": ("interleaved 16-row film tiles + %s" % ("RCCL reduce over xGMI" if args.backend == "nccl" else "%s reduce through host memory (rehearsal)" % args.backend)) if n > 1 else "single GPU",
                   "backend": (args.backend if n > 1 else None)},
        "mray_per_s": (tot_closest + tot_any) / dt / 1e6,
        "segments_per_sample": acc["segments"] / max(acc["samples"], 1),
        "per_rank_render_ms": rank_ms, "per_rank_reduce_ms": red_ms,
        # a "launch" is one iteration of the scheduler: with the default schedule it is issued as `concurrent` part-size kernels on
        # their own streams, which run side by side (rocprofv3 lists them separately; each lasts about one iteration)
        "kernel_ms": {"bounce_kernel_per_step": acc["bounce_ns"] / steps * 1e-6, "k_film_tiles_per_step": acc["film_ns"] / steps * 1e-6,
                      "bounce_kernel_launches_per_step": launches / steps, "bounce_kernel_avg_launch_us": bounce_s / launches * 1e6,
                      "bounce_kernel_concurrent": concurrent, "passes_per_step": acc["passes"] / steps},
        "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "alg_bytes_per_launch": alg_bytes / launches, **valu},
    }
    if n == 1 and not strong and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sd, lambda spp: scenes.cornell_box_sensor(width, height, spp), "cbox", seconds=args.cpu_seconds)
        out["vs_cpu"] = {"gpu_over_scalar": out["value"] / out["cpu_baseline"]["scalar"]}
        if "packet" in out["cpu_baseline"]:
            out["vs_cpu"]["gpu_over_packet"] = out["value"] / out["cpu_baseline"]["packet"]
    if n == 1 and not strong and not args.no_parity:
        out["parity"] = same_run_parity(render, scenes, scene, sd, R.device)
    return out


SAMPLE_RECORD_BYTES = 24       # per finished camera sample: 16 B (X, Y, Z, alpha) + 8 B film position


def run_mesh(args, R, matpreview=False):
    """configs[2]: ~250 k-triangle matpreview-style mesh, spectral variant, 1920x1080 @ 1024 spp on one GPU (2 passes).  Roofline of
    the BVH traversal kernel k_trace<false,false>: 48 B per closest-hit ray + the geometry once per launch, over the
    average launch duration measured with HIP events on the stream of each launch.  Two material sets on the same geometry, camera and
    area light: all-diffuse (`mesh`: the kernels specialised for one-sided diffuse BSDFs and area lights) and the materials of a
    material-preview scene (`mesh_matpreview`: roughplastic object, checkerboard ground, envmap sky: the general BSDF / emitter kernels)."""
    import numpy as np
    from mitsuba2_amd import render, scenes
    explicit = args.config in ("mesh", "mesh_matpreview")
    steps, warmup = (args.steps, args.warmup) if explicit else ((1, 1) if matpreview else (2, 1))
    variant = args.variant
    sd = scenes.matpreview(256, 512) if matpreview else scenes.bumpy_sphere(256, 512)
    scene = render.Scene(sd, device=R.device, variant=variant)
    info = scene.info()
    w, h, spp = args.mesh_width, args.mesh_height, args.mesh_spp
    sensor = render.make_sensor(scenes.bumpy_sphere_sensor(w, h, spp))
    integ = render.PathIntegrator(profile=True)
    dt, acc, _, _ = timed_renders(R, integ, scene, sensor, None, steps, warmup)
    launches = max(acc["trace_closest_launches"], 1)
    geometry = 32 * info["bvh_nodes"] + 48 * info["primitives"]
    alg_bytes = 48 * acc["closest_hit_rays"] + geometry * launches
    trace_s = acc["trace_closest_ns"] * 1e-9
    achieved = alg_bytes / max(trace_s, 1e-12) / 1e9
    any_launches = max(acc["trace_any_launches"], 1)
    any_bytes = 36 * acc["any_hit_rays"] + geometry * any_launches
    # Fabric traffic of the walks: PMC counters cannot be read inside this process; the ratio counted / algorithmic bytes of each kernel comes
    # from the committed rocprofv3 --pmc summary of the same scene AND variant (1920x1080 @ 64 spp: the kernels are the same) -- an OFFLINE
    # figure.  FETCH_SIZE counts the L2's memory-side requests, Infinity-Cache hits included (MI355X_MICROARCH.md): for a 17.7 MB scene the
    # excess over the algorithmic bytes is node / triangle lines that miss the 4 MB L2 of an XCD, served on-die -- not HBM reads.
    traffic, traffic_src, pmc_extra = None, None, {}
    pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mesh_spectral_pmc.json" if variant == "spectral" else "r[0-9][0-9]_mesh_pmc.json")))
    if pmc and not matpreview:
        try:
            prof = json.load(open(pmc[-1]))
            traffic = prof["k_trace_closest"]["traffic_over_algorithmic"] * (alg_bytes / launches)
            traffic_src = ("offline profile " + os.path.relpath(pmc[-1], ROOT) + " (counted / algorithmic bytes of this kernel on the %s 64-spp render; "
                           "L2-miss traffic, Infinity-Cache hits included), scaled to this run" % variant)
            pmc_extra = {"k_trace_any_traffic_over_algorithmic": prof.get("k_trace_any", {}).get("traffic_over_algorithmic"),
                         "k_shade_bytes_per_segment": prof.get("k_shade", {}).get("bytes_per_segment"),
                         "k_shade_traffic_over_algorithmic": prof.get("k_shade", {}).get("traffic_over_algorithmic"),
                         "k_trace_closest_lane_utilisation": prof["kernels"].get("k_trace<false, false>", {}).get("valu_lane_utilisation")}
        except (KeyError, ValueError, OSError):
            traffic, traffic_src = None, None
    out = {
        "metric": "Msample/s, %d k-triangle mesh, %s variant, %dx%d@%dspp, path integrator (max_depth=-1, rr_depth=5)" % (info["primitives"] // 1000, variant, w, h, spp),
        "value": acc["samples"] / dt / 1e6, "unit": "Msample/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "procedural displaced sphere (%d triangles, vertex normals) over a ground quad, one area light, %s, %s variant, "
                               "%dx%d film, %d spp, gaussian rfilter, independent sampler" % (info["primitives"],
                               "roughplastic (GGX) object, checkerboard ground, 256x512 envmap sky" if matpreview else "diffuse BSDFs", variant, w, h, spp),
                   "bvh_nodes": info["bvh_nodes"], "bvh_depth": info["bvh_depth"]},
        "mray_per_s": (acc["closest_hit_rays"] + acc["any_hit_rays"]) / dt / 1e6,
        "segments_per_sample": acc["segments"] / max(acc["samples"], 1),
        "tri_tests_per_ray": acc["tri_tests"] / max(acc["closest_hit_rays"] + acc["any_hit_rays"], 1),
        # the two launch chains run up to four kernels side by side: the summed kernel durations exceed the wall time of the loop
        "kernel_ms": {"loop_per_step": acc["bounce_ns"] / steps * 1e-6, "k_film_tiles_per_step": acc["film_ns"] / steps * 1e-6,
                      "k_trace_closest_sum_per_step": acc["trace_closest_ns"] / steps * 1e-6, "k_trace_any_sum_per_step": acc["trace_any_ns"] / steps * 1e-6,
                      "k_shade_sum_per_step": acc["shade_ns"] / steps * 1e-6, "k_trace_closest_launches_per_step": launches / steps,
                      "k_trace_closest_avg_launch_us": trace_s / launches * 1e6, "passes_per_step": acc["passes"] / steps},
        "roofline": {"bound": "hbm", "kernel": "k_trace<false, false>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "alg_bytes_per_launch": alg_bytes / launches,
                     "gray_per_s_in_kernel": acc["closest_hit_rays"] / max(trace_s, 1e-12) / 1e9,
                     "k_trace_any": {"achieved": any_bytes / max(acc["trace_any_ns"] * 1e-9, 1e-12) / 1e9,
                                     "gray_per_s_in_kernel": acc["any_hit_rays"] / max(acc["trace_any_ns"] * 1e-9, 1e-12) / 1e9},
                     "offline_counters": pmc_extra,
                     "bound_in_practice": "VALU issue for half-empty waves (lane utilisation 0.48) and the latency of dependent node fetches; the 17.7 MB of geometry are served by L2 and the Infinity Cache, HBM carries the ray / hit streams only"},
    }
    if not args.no_cpu_baseline:
        path = render.srgb_coeff_path() if variant == "spectral" else None
        out["cpu_baseline"] = cpu_baseline(sd, lambda s: scenes.bumpy_sphere_sensor(w, h, s), "mesh %s%s" % ("matpreview " if matpreview else "", variant),
                                           spectral_path=path, seconds=args.cpu_seconds, max_spp=4 if matpreview else 8)
    if not args.no_parity:
        # same kernels, a 96x64 @ 64 spp render of the same scene against the oracle with the same per-sample seeds
        ob = _oracle()
        p = scenes.bumpy_sphere_sensor(96, 64, 64)
        s2 = render.make_sensor(p)
        assert render.PathIntegrator().render(scene, s2)
        got = ob.film_develop(s2.film().bitmap(raw=True).cpu().numpy())[..., :3]
        S = ob.OracleScene(sd, spectral_path=render.srgb_coeff_path() if variant == "spectral" else None)
        ref = ob.film_develop(S.render(ob.make_desc(p), mode=1, n_threads=host_cores())[0])[..., :3]
        out["parity"] = {"render": "same scene 96x64@64spp", "relmse_vs_oracle_same_seeds": float(np.mean((got - ref) ** 2 / (ref ** 2 + 1e-2)))}
    return out


def run_autodiff(args, R):
    """configs[3]: the loop of docs/examples/10_inverse_rendering/invert_cbox.py (diff_render.rst:11-28: path max_depth=3, box
    filter; spp=1, unbiased=True, write_bitmap commented out) in two variants: recovering the red wall's constant reflectance
    ('red.reflectance.value', the reference's example) and -- BASELINE config 4 as worded, "optimise diffuse-albedo texture" -- the
    256 x 256 texels of a bitmap albedo on the back wall and the floor ('tex.reflectance.data', src/textures/bitmap.cpp:250-299: the
    adjoint scatters 12 float atomics per textured path vertex).  The reference quotes ~50 ms (unbiased) / ~27 ms (biased) per
    iteration on a Titan RTX (diff_render.rst:311-314), film size not stated."""
    import numpy as np
    import torch
    from mitsuba2_amd import render, scenes, autodiff
    res, iters = args.ad_res, args.ad_iters
    p = scenes.cornell_box_sensor(res, res, 1, max_depth=3, rfilter="box")

    def build(texture):
        sd = scenes.cornell_box(texture=texture)
        for b, nm in zip(sd["bsdfs"], ["white", "red", "green", "light", "tex"]):
            b["id"] = nm
        return sd, render.Scene(sd, device=R.device, sensor=render.make_sensor(p), integrator=render.PathIntegrator(max_depth=3))

    def optimise(scene, key, start, lr):
        """times `iters` iterations (after 5 warm-up ones) in the unbiased and the biased form; then the stream time of the primal render
        and of the backward pass of one biased iteration (HIP events on the current stream)"""
        params = autodiff.traverse(scene)
        params.keep([key])
        ref = params[key].clone()
        image_ref = autodiff.render(scene, spp=8).detach()
        result = {}
        for unbiased in (True, False):
            params[key] = start(ref)
            params.update()
            opt = autodiff.Adam(params, lr=lr)

            def iteration():
                img = autodiff.render(scene, optimizer=opt, unbiased=unbiased, spp=1)
                (((img - image_ref) ** 2).sum() / img.numel()).backward()
                opt.step()

            before = float(((ref - params[key].detach()) ** 2).mean().item())
            for _ in range(5):
                iteration()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                iteration()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / iters
            result["unbiased" if unbiased else "biased"] = {"ms_per_iteration": ms, "param_mse_at_start": before,
                                                            "param_mse_after_run": float(((ref - params[key].detach()) ** 2).mean().item())}
        params[key] = start(ref)
        params.update()
        opt = autodiff.Adam(params, lr=lr)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        fwd = adj = 0.0
        for _ in range(20):
            e[0].record()
            img = autodiff.render(scene, optimizer=opt, unbiased=False, spp=1)
            e[1].record()
            (((img - image_ref) ** 2).sum() / img.numel()).backward()
            e[2].record()
            torch.cuda.synchronize()
            fwd += e[0].elapsed_time(e[1]); adj += e[1].elapsed_time(e[2])
            opt.step()
        params[key] = ref
        params.update()
        return result, fwd / 20, adj / 20, image_ref

    sd, scene = build(None)
    result, fwd, adj, image_ref = optimise(scene, "red.reflectance.value", lambda ref: torch.full_like(ref, 0.9), 0.2)
    # texture variant: a smooth 256 x 256 albedo pattern is the target, a uniform grey the starting point
    yy, xx = np.meshgrid(np.linspace(0, 1, 256, dtype=np.float32), np.linspace(0, 1, 256, dtype=np.float32), indexing="ij")
    tex = np.stack([0.5 + 0.35 * np.sin(9 * xx) * np.cos(7 * yy), 0.45 + 0.3 * np.cos(5 * xx + 3 * yy), 0.4 + 0.3 * np.sin(11 * yy)], -1).astype(np.float32)
    sd_t, scene_t = build(tex)
    res_t, fwd_t, adj_t, _ = optimise(scene_t, "tex.reflectance.data", lambda ref: torch.full_like(ref, 0.5), 0.05)
    # textured path vertices of one render (upper bound: every vertex of a path lies on a textured surface): segments per sample
    st_integ = render.PathIntegrator(max_depth=3)
    st_integ.render(scene_t, render.make_sensor(p))
    seg_per_sample = st_integ.stats["segments"] / max(st_integ.stats["samples"], 1)
    n_samples = res * res
    # k_adjoint: per camera sample 12 B dLoss/dImage + 4 B film weight of its box-filter pixel; the gradient of a constant reflectance
    # is reduced in LDS (one atomic per workgroup); a textured vertex adds 12 float atomics (4 texels x 3 channels) = 48 B
    adj_bytes = 16 * n_samples
    adj_bytes_t = 16 * n_samples + int(48 * seg_per_sample * n_samples)
    ub = result["unbiased"]["ms_per_iteration"]
    out = {
        "metric": "ms per optimisation iteration (primal + derivative render + adjoint + Adam), differentiable cbox %dx%d, spp 1, max_depth 3, box filter, unbiased=True" % (res, res),
        "value": ub, "unit": "ms", "n_gpus": 1, "steps": iters, "warmup": 5, "ms_per_step": ub, "higher_is_better": False, "scaling": "weak",
        # ~50 ms / iteration (unbiased) on an NVIDIA Titan RTX, film size not stated in the reference tree
        "vs_baseline": ub / 50.0, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "synthetic Cornell box, optimise 'red.reflectance.value' from 0.9 with Adam(lr=0.2) against an 8-spp reference image, "
                               "%dx%d film (the reference does not state its film size), spp 1, path max_depth 3, box rfilter" % (res, res)},
        "biased_ms_per_iteration": result["biased"]["ms_per_iteration"], "vs_baseline_biased": result["biased"]["ms_per_iteration"] / 27.0,
        "param_mse_after_run": result["unbiased"]["param_mse_after_run"],
        "kernel_ms": {"forward_render_stream_ms": fwd, "backward_stream_ms": adj},
        "roofline": {"bound": "hbm", "kernel": "k_adjoint<true>", "achieved": adj_bytes / max(adj * 1e-3, 1e-12) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": adj_bytes / max(adj * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, "traffic": None, "alg_bytes_per_launch": adj_bytes,
                     "note": "%d camera samples per launch: the iteration is launch-latency bound (a handful of ~10 us kernels), not bandwidth bound" % n_samples},
        # BASELINE config 4 as worded: the albedo TEXTURE (256 x 256 x 3 texels on the back wall and the floor) is the parameter
        "texture": {
            "workload": "same setup, optimise the 256x256x3 texels of 'tex.reflectance.data' (back wall + floor) from uniform 0.5 with Adam(lr=0.05)",
            "ms_per_iteration": res_t["unbiased"]["ms_per_iteration"], "biased_ms_per_iteration": res_t["biased"]["ms_per_iteration"],
            "vs_baseline": res_t["unbiased"]["ms_per_iteration"] / 50.0,
            "texel_mse_at_start": res_t["unbiased"]["param_mse_at_start"], "texel_mse_after_run": res_t["unbiased"]["param_mse_after_run"],
            "kernel_ms": {"forward_render_stream_ms": fwd_t, "backward_stream_ms": adj_t},
            "roofline": {"bound": "hbm", "kernel": "k_adjoint<true> (float atomics into the texel gradient)", "achieved": adj_bytes_t / max(adj_t * 1e-3, 1e-12) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": adj_bytes_t / max(adj_t * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "alg_bytes_per_launch": adj_bytes_t, "atomic_bytes_per_launch_upper_bound": int(48 * seg_per_sample * n_samples),
                         "note": "48 B of float atomics per textured path vertex, counted for every vertex (%.2f per sample): an upper bound" % seg_per_sample}},
    }
    if not args.no_cpu_baseline:
        ob = _oracle()
        S = ob.OracleScene(sd)
        d = ob.make_desc(p, analytic=True, film_rgb=True)
        target = image_ref.cpu().numpy().reshape(res, res, 3)
        t0 = time.perf_counter()
        k = 0
        while k < 3 or time.perf_counter() - t0 < min(args.cpu_seconds, 5.0):
            img, film = S.render_image(d)
            S.render_adjoint(d, 2.0 * (img - target) / img.size, film, len(sd["meshes"]), 0)
            k += 1
        out["cpu_baseline"] = {"value": (time.perf_counter() - t0) * 1e3 / k, "unit": "ms", "cores": host_cores(), "kind": "port",
                               "sample": "%d iterations of oracle primal render + adjoint (biased form, no optimiser step) on the same %dx%d setup" % (k, res, res)}
    return out


# ------------------------------------------------------------------------------------------------------------------
def launch_ranks(args):
    """`python bench.py --gpus N` on its own: start the N ranks as a torch.distributed.run child before this process touches the GPU
    (never exec from a process that has initialised HIP); refuse if the node has fewer GPUs than ranks (nccl backend)."""
    import torch
    n_dev = torch.cuda.device_count()           # does not initialise the GPU on this image
    if args.config == "launcher-selftest":
        n_dev = max(n_dev, 1)
    if args.backend == "nccl" and n_dev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible; refusing to report a %d-GPU number "
                         "(use --backend gloo to rehearse the ranks on fewer cards)\n" % (args.gpus, n_dev, args.gpus))
        return 2
    if n_dev < 1:
        sys.stderr.write("bench.py: no GPU visible\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cbox", choices=["cbox", "mesh", "mesh_matpreview", "autodiff", "cbox4k", "launcher-selftest"])
    ap.add_argument("--only", action="store_true", help="--config cbox without the other configurations")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--width4k", type=int, default=4096)
    ap.add_argument("--spp4k", type=int, default=4096)
    ap.add_argument("--mesh-width", type=int, default=1920)
    ap.add_argument("--mesh-height", type=int, default=1080)
    ap.add_argument("--mesh-spp", type=int, default=1024)
    ap.add_argument("--variant", default="spectral", choices=["rgb", "spectral"])
    ap.add_argument("--ad-res", type=int, default=256)
    ap.add_argument("--ad-iters", type=int, default=100)
    ap.add_argument("--paths-per-wave", type=int, default=0)
    ap.add_argument("--pipeline", type=int, default=0, help="0 automatic, 1 fused, 2 split, 3 fused closest + queued shadow rays")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the real path) or gloo (rehearsal of N ranks on fewer GPUs)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be positive")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.config in ("mesh", "mesh_matpreview", "autodiff") and args.gpus != 1:
        ap.error("--config %s is a single-GPU configuration" % args.config)

    R = Ranks(args)
    if args.config == "launcher-selftest":
        R.barrier()
        out = {"n_gpus": R.world, "backend": args.backend, "ranks": R.gather(R.rank), "sum": R.sum([R.rank + 1])[0], "max": R.max(R.rank)}
    elif args.config in ("mesh", "mesh_matpreview"):
        out = run_mesh(args, R, matpreview=args.config == "mesh_matpreview")
    elif args.config == "autodiff":
        out = run_autodiff(args, R)
    elif args.config == "cbox4k":
        out = run_cbox(args, R, strong=True)
    else:
        out = run_cbox(args, R)
        if not args.only:
            others = {}
            if R.world == 1:
                others["mesh"] = run_mesh(args, R)
                others["mesh_matpreview"] = run_mesh(args, R, matpreview=True)
                others["autodiff"] = run_autodiff(args, R)
            strong = run_cbox(args, R, strong=True)
            if R.rank == 0:
                others["cbox4k"] = strong
                out["other_configs"] = others
    if R.rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    R.close()


if __name__ == "__main__":
    main()
