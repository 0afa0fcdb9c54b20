#!/usr/bin/env python3
"""Where the lanes of the k_trace waves go (experiment build: scripts/ab_build.sh prof -DMTS_TRACE_PROF=1, MTSAMD_LIB set to it).
Renders the 261 k-triangle mesh once and prints, per phase of the walk, wave-level events, lane events and mean active lanes."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mitsuba2_amd import render, scenes, _lib

NAMES = ["rounds (walk_round calls)", "node steps", "leaf phases", "triangle-test iterations", "stack pops", "fetches",
         "work-loop iterations", "retires", "node steps: lanes at a leaf", "node steps: lanes w/o ray",
         "rounds, list not exhausted", "rounds, list exhausted"]


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "rgb"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    lib = _lib.lib()
    fn = lib.mtsamd_debug_trace_prof
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    scene = render.Scene(scenes.bumpy_sphere(256, 512), variant=variant)
    sensor = render.make_sensor(scenes.bumpy_sphere_sensor(1920, 1080, spp))
    integ = render.PathIntegrator()
    integ.render(scene, sensor)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 64)()
    assert fn(out, 0) == 0
    st = integ.stats
    print("closest-hit rays %d, any-hit rays %d, tri tests %d" % (st["closest_hit_rays"], st["any_hit_rays"], st["tri_tests"]))
    for kind, base, rays in (("closest", 0, st["closest_hit_rays"]), ("any", 32, st["any_hit_rays"])):
        print("== k_trace<%s>" % kind)
        for i, name in enumerate(NAMES):
            ev, lanes = out[base + 2 * i], out[base + 2 * i + 1]
            print("  %-28s wave events %12d  lane events %13d  mean active lanes %5.1f  per ray %6.2f"
                  % (name, ev, lanes, lanes / max(ev, 1), lanes / max(rays, 1)))


if __name__ == "__main__":
    main()
