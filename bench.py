#!/usr/bin/env python3
"""Headline benchmark: Msample/s (and Mray/s) of the `path` integrator on the synthetic Cornell box,
1024x1024 @ 256 spp per GPU (BASELINE.json configs[1]), with the dominant kernel's achieved fraction of
the HBM roofline and the CPU port (the oracle, scalar_rgb block mode) timed on the same box.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one complete render (every sample traced to termination, film splatted).  With N > 1 the film is
partitioned into interleaved 32-row tiles (mitsuba2_amd/dist.py), one rank per GPU renders its tiles into a
full-size XYZAW film and the films are summed with one RCCL reduce to rank 0.  Weak scaling: the sample count
grows to 256*N spp so that every GPU keeps tracing 1024*1024*256 = 2^28 camera samples per step.
"""
import argparse
import json
import os
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
    return n


def cpu_baseline(width, height, seconds=12.0):
    """The CPU port (oracle, scalar_rgb block mode, BVH, all host cores) on a bounded sample of the same
    workload: same film, reduced spp.  Checker code used as the *baseline*, never as the thing measured."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from mitsuba2_amd import scenes
    sd = scenes.cornell_box()
    S = ob.OracleScene(sd)
    cores = host_cores()
    # calibrate on 1 spp, then pick an spp that gives roughly `seconds` of CPU work
    p = scenes.cornell_box_sensor(width, height, 1)
    t0 = time.perf_counter()
    S.render(ob.make_desc(p), mode=0, n_threads=cores)
    dt1 = max(time.perf_counter() - t0, 1e-3)
    spp = int(max(1, min(64, round(seconds / dt1))))
    p = scenes.cornell_box_sensor(width, height, spp)
    t0 = time.perf_counter()
    _, stats = S.render(ob.make_desc(p), mode=0, n_threads=cores)
    dt = time.perf_counter() - t0
    samples = width * height * spp
    return {"value": samples / dt / 1e6, "unit": "Msample/s", "cores": cores, "kind": "port",
            "sample": "cbox %dx%d@%dspp, oracle scalar_rgb block mode (spiral 32x32 blocks, BVH), %.1f s" % (width, height, spp, dt),
            "mray_per_s": float(stats[0] + stats[1]) / dt / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--paths-per-wave", type=int, default=0)
    ap.add_argument("--pipeline", type=int, default=0, help="0 automatic, 1 fused, 2 split, 3 fused closest + queued shadow rays")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the real path) or gloo (rehearsal of N ranks on fewer GPUs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mitsuba2_amd import render, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    n = world
    local_rank = local_rank % torch.cuda.device_count()      # rehearsal: several ranks may share a card
    torch.cuda.set_device(local_rank)

    from mitsuba2_amd import dist as mdist
    width, height = args.width, args.height
    spp_total = args.spp * n                # weak scaling: 2^28 camera samples per GPU and step
    sd = scenes.cornell_box()
    p = scenes.cornell_box_sensor(width, height, spp_total)
    scene = render.Scene(sd, device=local_rank)
    sensor = render.make_sensor(p)
    integ = render.PathIntegrator(paths_per_wave=args.paths_per_wave, pipeline=args.pipeline)
    partition = mdist.film_partition(rank, n)

    def step():
        ok = integ.render(scene, sensor, partition=partition)
        assert ok
        return mdist.reduce_film(sensor.film().bitmap(raw=True))   # RCCL over xGMI: per-rank ImageBlocks -> rank 0

    def barrier():
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    acc = dict(closest_hit_rays=0, any_hit_rays=0, samples=0, iterations=0, segments=0, bounce_ns=0, film_ns=0, tri_tests=0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k in acc:
            acc[k] += integ.stats[k]
    barrier()
    dt = time.perf_counter() - t0
    if n > 1:
        cdev = "cuda" if args.backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.tensor([acc["closest_hit_rays"], acc["any_hit_rays"], acc["samples"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_closest, tot_any, tot_samples = [float(x) for x in cnt.tolist()]
    else:
        tot_closest, tot_any, tot_samples = float(acc["closest_hit_rays"]), float(acc["any_hit_rays"]), float(acc["samples"])

    if rank == 0:
        # roofline of the dominant kernel on rank 0 -- the one that advances every in-flight path by one segment: k_shade<PathState,
        # false, true, true> (closest hit + shading + in-kernel shadow ring, the default schedule for LDS-resident scenes) or k_bounce
        # with --pipeline 1: algorithmic bytes per launch / average launch time.
        # Per segment the kernel reads one 88-B path record and writes one (a survivor or a regenerated camera path:
        # survivors + generated == segments over a whole render); per finished sample it writes 16 B radiance + 8 B
        # film position.  Launch durations come from HIP events recorded on the render stream inside mtsamd_render.
        launches = max(acc["iterations"], 1)
        alg_bytes = 2 * STATE_BYTES * acc["segments"] + 24 * acc["samples"]
        bounce_s = acc["bounce_ns"] * 1e-9
        achieved = alg_bytes / max(bounce_s, 1e-12) / 1e9
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3 passes), measured
        # offline with scripts/collect_profiles.sh on the same kernel at 32 spp and committed under profiles/ as the ratio of
        # counted to algorithmic bytes; scaled here to this run's bytes per launch (launch sizes differ with the sample count).
        traffic, traffic_src = None, None
        import glob
        kernel = {0: "k_shade<mtsamd::PathState, false, true, true>", 4: "k_shade<mtsamd::PathState, false, true, true>",
                  1: "k_bounce<true, false>", 2: "k_trace<false, false> + k_shade<mtsamd::PathState, false, false, false> + k_trace<true, false>",
                  3: "k_shade<mtsamd::PathState, false, true, false> + k_trace<true, true>"}[args.pipeline]
        concurrent = min(4, max(1, int(os.environ.get("MTSAMD_STREAMS", "2")))) if args.pipeline in (0, 4) else 1
        pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_bounce_kernel.json")))
        if pmc:
            try:
                traffic = json.load(open(pmc[-1]))["traffic_over_algorithmic"] * (alg_bytes / launches)
                traffic_src = os.path.relpath(pmc[-1], ROOT)
            except (OSError, ValueError, KeyError):
                pass
        out = {
            "metric": "Msample/s, cbox %dx%d@%dspp per GPU, path integrator (max_depth=-1, rr_depth=5)" % (width, height, args.spp),
            "value": tot_samples / dt / 1e6,
            "unit": "Msample/s",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "synthetic Cornell box (36 triangles, diffuse, 1 area light), %dx%d film, %d spp total (%d per GPU), "
                                   "gaussian rfilter, independent sampler" % (width, height, spp_total, args.spp),
                       "partition": "interleaved 32-row film tiles + RCCL reduce" if n > 1 else "single GPU"},
            "mray_per_s": (tot_closest + tot_any) / dt / 1e6,
            "segments_per_sample": acc["segments"] / max(acc["samples"], 1),
            # a "launch" is one iteration of the scheduler: with the default schedule it is issued as `concurrent` part-size kernels on
            # their own streams, which run side by side (rocprofv3 lists them separately; each lasts about one iteration)
            "kernel_ms": {"bounce_kernel_per_step": acc["bounce_ns"] / args.steps * 1e-6, "k_film_tiles_per_step": acc["film_ns"] / args.steps * 1e-6,
                          "bounce_kernel_launches_per_step": launches / args.steps, "bounce_kernel_avg_launch_us": bounce_s / launches * 1e6,
                          "bounce_kernel_concurrent": concurrent},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "alg_bytes_per_launch": alg_bytes / launches},
        }
        if not args.no_cpu_baseline and n == 1:
            out["cpu_baseline"] = cpu_baseline(width, height)
        print(json.dumps(out))
    if n > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
