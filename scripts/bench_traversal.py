#!/usr/bin/env python3
"""Micro-benchmark of the standalone BVH traversal kernels (Scene::ray_intersect / ray_test streams).
Ray distributions: primary camera rays, cosine-hemisphere secondary rays spawned from the
primary hits, uniform random segments inside the bbox.  Prints Mray/s and algorithmic GB/s (48 B / closest-hit
ray, 36 B / any-hit ray)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba2_amd import render, scenes, _lib as L


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in evs) * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cbox")
    ap.add_argument("--n", type=int, default=1 << 24)
    args = ap.parse_args()
    if args.scene == "cbox":
        sd, sp = scenes.cornell_box(), scenes.cornell_box_sensor(1024, 1024, 1)
    else:
        sd, sp = scenes.bumpy_sphere(256, 512), scenes.bumpy_sphere_sensor(1024, 1024, 1)
    scene, sensor = render.Scene(sd), render.make_sensor(sp)
    print("scene:", scene.info())
    n = args.n
    g = torch.Generator(device="cuda").manual_seed(0)
    prim = sensor.sample_ray(torch.rand((n, 2), device="cuda", generator=g))
    si = scene.ray_intersect(prim)
    valid = si.is_valid()
    # secondary rays: cosine hemisphere around the shading normal at the primary hits
    u = torch.rand((n, 2), device="cuda", generator=g)
    r, phi = torch.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
    loc = torch.stack([r * torch.cos(phi), r * torch.sin(phi), torch.sqrt(torch.clamp(1 - u[:, 0], min=0))], 1)
    d = si.sh_frame_s * loc[:, 0:1] + si.sh_frame_t * loc[:, 1:2] + si.sh_frame_n * loc[:, 2:3]
    o = torch.where(valid[:, None], si.p, prim.o)
    d = torch.where(valid[:, None], d, prim.d)
    sec = render.Ray3f(o=o.contiguous(), d=d.contiguous(), mint=torch.full((n,), 1e-2, device="cuda"))
    lo, hi = scene.bbox()
    lo, hi = torch.from_numpy(lo).cuda(), torch.from_numpy(hi).cuda()
    a = lo + (hi - lo) * torch.rand((n, 3), device="cuda", generator=g)
    b = lo + (hi - lo) * torch.rand((n, 3), device="cuda", generator=g)
    dist = torch.linalg.norm(b - a, dim=1)
    rnd = render.Ray3f(o=a, d=((b - a) / dist[:, None]).contiguous(), mint=torch.full((n,), 1e-4, device="cuda"), maxt=dist.contiguous())
    for name, ray in (("primary", prim), ("secondary", sec), ("random-segment", rnd)):
        rs, keep, _, dev = scene._soa(ray, True)
        t = torch.empty(n, device="cuda"); pr = torch.empty(n, dtype=torch.int32, device="cuda"); sh = torch.empty_like(pr)
        uu = torch.empty(n, device="cuda"); vv = torch.empty(n, device="cuda"); hit = torch.empty(n, dtype=torch.uint8, device="cuda")
        lib = L.lib()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda x: C.c_void_p(x.data_ptr())
        dt_c = timed(lambda: lib.mtsamd_ray_intersect(scene._handle, n, C.byref(rs), p(t), p(pr), p(sh), p(uu), p(vv), st))
        dt_a = timed(lambda: lib.mtsamd_ray_test(scene._handle, n, C.byref(rs), p(hit), st))
        if scene.info()['primitives'] <= 64:
            dt_n = timed(lambda: lib.mtsamd_ray_intersect_naive(scene._handle, n, C.byref(rs), p(t), p(pr), p(sh), p(uu), p(vv), st))
            print('%-15s naive closest: %8.1f Mray/s' % (name, n / dt_n / 1e6))
        print("%-15s closest: %8.1f Mray/s %7.1f GB/s (%.3f of 8 TB/s) hit rate %.2f | any: %8.1f Mray/s %7.1f GB/s (%.3f)" % (
            name, n / dt_c / 1e6, 48 * n / dt_c / 1e9, 48 * n / dt_c / 8e12, torch.isfinite(t).float().mean().item(),
            n / dt_a / 1e6, 36 * n / dt_a / 1e9, 36 * n / dt_a / 8e12))


if __name__ == "__main__":
    main()
