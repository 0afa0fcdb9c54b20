#!/usr/bin/env python3
"""Render throughput on a large procedural mesh (BASELINE config 3 geometry class: ~250 k triangles with vertex normals,
one area light) -- exercises the hierarchy (BVH) traversal path.  RGB or spectral variant; prints Msample/s and Mray/s."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitsuba2_amd import render, scenes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--theta", type=int, default=256)
    ap.add_argument("--phi", type=int, default=512)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--variant", default="rgb", choices=["rgb", "spectral"])
    ap.add_argument("--scene", default="diffuse", choices=["diffuse", "matpreview"], help="matpreview: roughplastic object, checkerboard ground, envmap")
    args = ap.parse_args()
    t0 = time.perf_counter()
    sd = scenes.matpreview(args.theta, args.phi) if args.scene == "matpreview" else scenes.bumpy_sphere(args.theta, args.phi)
    t1 = time.perf_counter()
    scene = render.Scene(sd, variant=args.variant)
    t2 = time.perf_counter()
    print("scene:", scene.info(), "mesh gen %.1f s, upload+BVH %.1f s" % (t1 - t0, t2 - t1))
    w, h = (args.width or args.res), (args.height or args.res)
    sensor = render.make_sensor(scenes.bumpy_sphere_sensor(w, h, args.spp))
    integ = render.PathIntegrator()
    integ.render(scene, sensor)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    integ.render(scene, sensor)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = integ.stats
    print("%dx%d@%dspp: %.1f ms, %.1f Msample/s, %.1f Mray/s, %.2f segments/sample, %.1f tri tests/ray, k_bounce %.1f ms, film %.1f ms"
          % (w, h, args.spp, dt * 1e3, st["samples"] / dt / 1e6, (st["closest_hit_rays"] + st["any_hit_rays"]) / dt / 1e6,
             st["segments"] / st["samples"], st["tri_tests"] / max(st["closest_hit_rays"] + st["any_hit_rays"], 1),
             st["bounce_ns"] * 1e-6, st["film_ns"] * 1e-6))


if __name__ == "__main__":
    main()
